#!/usr/bin/env python3
"""Rate of parallel tempering (core.py:131-236: every chain at its own temperature, one swap attempt per generation) at 4096 chains x 100-D.
DZ_MEGA=0: the multi-kernel path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
N, d, gens, k = 4096, 100, 2000, 5
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
Z0 = np.random.default_rng(3).uniform(-5, 15, (2 * N, d))
e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (gens // 10 + 30), trace_capacity=0, seed=5)
e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
e.set_temperatures(np.power(.001, np.arange(N) / float(N)), swaps=True)
e.step(200); e.sync()
t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
print("parallel tempering, %d chains x %d-D, multitry %d: %.1f M proposals/s (%.1f us per generation), %s" % (N, d, k, N * k * gens / dt / 1e6, 1e6 * dt / gens, e.last_kernel_variant()))
