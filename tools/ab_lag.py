#!/usr/bin/env python3
"""A/B of history_lag = 0 and 1 on ONE box in ONE process (round-4 verdict item 8: the driver's line had 673.6 M/s at lag 1 and 685.0 at lag 0,
round 3's 690 at lag 0 -- is the lag the cost, or the order in which bench.py measures its engines?).  Two engines of the headline
workload, both warmed up, then blocks of K generations ALTERNATING between them (ABAB...), medians per engine; then the same with the
order of creation swapped.
    python tools/ab_lag.py [K] [blocks per engine]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pydream_amd import _capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 200
_capi.load_library()
n = 4096


def make(lag):
    args = argparse.Namespace(dim=100, multitry=5, thin=10, seed=20260929, target="mvn", mvn_kind="tri", snooker=0.1, steps=K, warmup=100, history_lag=lag)
    e = bench.setup_engine(_capi.Engine, args, n, n, 0, 6000 + K * (NB + 4) * 2, trace_capacity=max(K, 500))
    for _ in range(8):                       # 4000 generations: converged, archive of 1.6 M rows, clocks up
        e.trace_reset(); e.step(500)
    e.sync()
    g = e.generation() if callable(e.generation) else e.generation
    al = (-(g - 1)) % 10
    if al:
        e.trace_reset(); e.step(al)
    e.sync()
    return e


for order in ((0, 1), (1, 0)):
    eng = {lag: make(lag) for lag in order}
    t = {0: [], 1: []}
    for b in range(NB):
        for lag in order:
            e = eng[lag]
            e.trace_reset(); e.sync()
            t0 = time.perf_counter(); e.step(K); e.sync(); t[lag].append(time.perf_counter() - t0)
    for lag in order:
        a = np.array(t[lag])
        print("created %s, K = %d, %d alternating blocks: lag %d  median %.1f M proposals/s  (best %.1f, worst %.1f)  %s"
              % ("lag %d first" % order[0], K, NB, lag, n * 5 * K / np.median(a) / 1e6, n * 5 * K / a.min() / 1e6, n * 5 * K / a.max() / 1e6, eng[lag].last_kernel_variant()), flush=True)
    for e in eng.values():
        e.close()
