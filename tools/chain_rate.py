#!/usr/bin/env python3
"""Rate of the headline workload's kernels at a given chain count (100-D MVN, triangular factor, multitry 5, no trace buffer), steady state:
    python tools/chain_rate.py <chains> [generations] [snooker probability]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
N = int(sys.argv[1]); gens = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
snooker = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
d, k = 100, 5
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
Z0 = np.random.default_rng(3).uniform(-5, 15, (max(1000, 2 * N), d))
e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * ((3 * gens + 4000) // 10 + 30), trace_capacity=0, seed=5, history_lag=int(os.environ.get('DZ_SCAN_LAG', '1')), snooker=snooker)
e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
e.step(4000); e.sync()
best = 0
for rep in range(3):
    t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
    best = max(best, N * k * gens / dt / 1e6)
print("N=%d snooker %g  %.1f M proposals/s  %.2f us/gen  %s" % (N, snooker, best, N * k / best, e.last_kernel_variant()))
