#!/usr/bin/env python3
"""Repeated timed passes of the bench workload in ONE process (GPU-side variance / clock behaviour study)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pydream_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=10)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--sleep", type=float, default=0.0)
a = ap.parse_args()
args = argparse.Namespace(dim=100, multitry=5, thin=10, seed=20260929, target="mvn", mvn_kind="tri", snooker=0.1,
                          steps=a.steps, warmup=50)
_capi.load_library()
n = 4096
e = bench.setup_engine(_capi.Engine, args, n, n, 0, a.steps * a.passes + 100)
e.step(50); e.sync()
for i in range(a.passes):
    e.trace_reset()
    t0 = time.perf_counter()
    e.step(a.steps)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print("pass %2d: %.1f us/step (enqueue %.1f us/step)  %.1f M/s" % (i, 1e6 * (t2 - t0) / a.steps, 1e6 * (t1 - t0) / a.steps, n * 5 * a.steps / (t2 - t0) / 1e6), flush=True)
    if a.sleep: time.sleep(a.sleep)
