#!/usr/bin/env python3
"""Throughput of the persistent kernel's less common instantiations at 4096 chains x 100-D (triangular factor): priors + hard
boundaries, several DE pairs, multitry off.  usage: python tools/variant_rates.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G

N, d, gens = 4096, 100, 2000
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
Z0 = np.random.default_rng(3).uniform(-5, 15, (3 * 2 * N, d))
for name, k, depairs, prior in (("flat, k=5", 5, 1, None), ("uniform priors + hard boundaries, k=5", 5, 1, "uniform"), ("normal priors, k=5", 5, 1, "normal"),
                                ("uniform priors, no hard boundaries (redraw check), k=5", 5, 1, "uniform-open"),
                                ("DEpairs=3, k=5", 5, 3, None), ("flat, multitry off", 1, 1, None), ("uniform priors + hard boundaries, multitry off", 1, 1, "uniform"),
                                ("DEpairs=3, multitry off", 1, 3, None)):
    if len(sys.argv) > 1 and sys.argv[1] not in name:          # (python tools/variant_rates.py "redraw": that case only, e.g. under rocprofv3)
        continue
    e = G.Engine(nchains=N, ndim=d, multitry=k, depairs=depairs, hardboundaries=0 if prior == "uniform-open" else 1, history_capacity=len(Z0) + N * (gens // 10 + 30), trace_capacity=0, seed=5, snooker=float(os.environ.get("DZ_VR_SNOOKER", "0.1")))
    if depairs > 1:
        e.set_gamma_table(np.array([[2.38 / np.sqrt(2.0 * (dl + 1) * np.arange(1, d + 1)) for dl in range(depairs)]]))
    if prior == "uniform":
        e.set_prior(np.full(d, 2, np.int32), np.full(d, -10.0), np.full(d, 30.0)); e.set_bounds(np.full(d, -10.0), np.full(d, 20.0))
    elif prior == "uniform-open":      # proposals may leave the support: multi-kernel path, one read-back per generation (Dream.py:281-289)
        e.set_prior(np.full(d, 2, np.int32), np.full(d, -10.0), np.full(d, 30.0))
    elif prior == "normal":
        e.set_prior(np.full(d, 1, np.int32), np.zeros(d), np.full(d, 30.0))
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
    e.step(200); e.sync()
    t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
    print("%-56s %7.1f M proposals/s  (%.1f us per generation)%s" % (name, N * k * gens / dt / 1e6, 1e6 * dt / gens,
                                                                    "  redraw rounds %d" % e.redraw_rounds() if prior == "uniform-open" else ""))
    e.close()
