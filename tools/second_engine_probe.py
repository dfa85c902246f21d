#!/usr/bin/env python3
"""Does a second engine in the same process run as fast as the first?  (bench.py's dense leg is one.)  usage: second_engine_probe.py [chains]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G

N, d, gens = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 100, 2000
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
Z0 = np.random.default_rng(3).uniform(-5, 15, (max(10 * d, 2 * N), d))
for name, M, kind, tcap in (("tri", U, 1, 0), ("dense", P, 0, 0), ("dense + trace", P, 0, 1000), ("tri + trace", U, 1, 1000)):
    e = G.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (3 * gens // 10 + 60), trace_capacity=tcap, seed=5)
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), M, kind, 0.0)
    e.step(200); e.sync()
    e.profile_enable(True); e.profile_reset()
    if tcap: e.trace_reset()
    e.step(100); e.sync()
    launches = {k: e.profile_get(k)[1] for k in ("generations", "propose", "logp", "accept")}
    e.profile_enable(False)
    rates = []
    for rep in range(2):
        if tcap: e.trace_reset()
        n = min(gens, tcap) if tcap else gens
        t0 = time.perf_counter(); e.step(n); e.sync(); dt = time.perf_counter() - t0
        rates.append(N * 5 * n / dt / 1e6)
    print("%-14s %s M proposals/s   launches per 100 generations: %s" % (name, ["%.1f" % r for r in rates], launches))
    e.close()
