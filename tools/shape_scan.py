#!/usr/bin/env python3
"""Which kernel runs, and how fast, across shapes (4096 chains unless given; MVN triangular factor, flat prior, no trace buffer):
    python tools/shape_scan.py [chains] [dims, comma separated] [multitry values, comma separated] [mix]
looks for cliffs -- shapes that fall off the persistent kernel"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
DIMS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 50, 100, 128, 160, 200, 256, 512, 1000]
KS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
MIX = len(sys.argv) > 4 and sys.argv[4] == "mix"          # fourth argument "mix": the 3-component Gaussian mixture (the wave-per-chain kernels) instead of the MVN
for d in DIMS:
    i = np.arange(1, d + 1.0)
    P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = np.random.default_rng(3).uniform(-5, 15, (2 * N, d))
    for k in (KS or ((1, 3, 5, 8, 12, 16) if d <= 128 else (1, 5))):
        gens = 600 if d <= 256 else 200
        e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (gens // 10 + 40), trace_capacity=0, seed=5, history_lag=int(os.environ.get('DZ_SCAN_LAG', '1')))
        e.set_history(Z0); e.set_state(Z0[:N])
        if MIX:
            e.set_likelihood_mixture(np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)]), np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi))
        else:
            e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(100); e.sync()
        reps = []
        for rep in range(3):          # (three passes: the first can be a cold one)
            t0 = time.perf_counter(); e.step(gens // 3); e.sync(); reps.append((time.perf_counter() - t0) / (gens // 3))
        dt = min(reps) * gens
        print("d=%4d k=%2d  %7.1f M proposals/s  %6.1f us/gen  %s   (passes: %s us/gen)" % (d, k, N * k * gens / dt / 1e6, 1e6 * dt / gens, e.last_kernel_variant(), " ".join("%.1f" % (1e6 * r) for r in reps)), flush=True)
        e.close()
