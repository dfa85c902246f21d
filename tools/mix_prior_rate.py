#!/usr/bin/env python3
"""4096 chains x 100-D three-component mixture with uniform priors + hard boundaries / normal priors / DEpairs = 3: the mixture kernel's
full-code instantiation (DZ_MEGA_MIX_PB=0: the multi-kernel path these configurations took before round 4)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
N, d, gens, k = 4096, 100, 2000, 5
mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
logF = np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi)
rng = np.random.default_rng(3)
Z0 = mu[rng.integers(0, 3, 2 * 3 * N)] + 2.0 * rng.standard_normal((6 * N, d))
for name, depairs, prior in (("flat", 1, None), ("uniform priors + hard boundaries", 1, "uniform"), ("normal priors", 1, "normal"), ("DEpairs=3", 3, None)):
    e = G.Engine(nchains=N, ndim=d, multitry=k, depairs=depairs, history_capacity=len(Z0) + N * (gens // 10 + 30), trace_capacity=0, seed=5)
    if depairs > 1:
        e.set_gamma_table(np.array([[2.38 / np.sqrt(2.0 * (dl + 1) * np.arange(1, d + 1)) for dl in range(depairs)]]))
    if prior == "uniform":
        e.set_prior(np.full(d, 2, np.int32), np.full(d, -20.0), np.full(d, 40.0)); e.set_bounds(np.full(d, -20.0), np.full(d, 20.0))
    elif prior == "normal":
        e.set_prior(np.full(d, 1, np.int32), np.zeros(d), np.full(d, 30.0))
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mixture(mu, logF)
    e.step(200); e.sync()
    t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
    print("mixture, %-34s %7.1f M proposals/s  (%.1f us per generation)  %s" % (name, N * k * gens / dt / 1e6, 1e6 * dt / gens, e.last_kernel_variant()))
    e.close()
