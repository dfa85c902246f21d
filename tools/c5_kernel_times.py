#!/usr/bin/env python3
"""Per-kernel event times of the configs[4] shard (512 chains x 1000-D, multitry 5) over 100 generations -- with DZ_CUMASK=<n> every kernel of
the engine runs on n compute units: what the likelihood product loses on 224 CUs and what the streamed proposal kernels need on 32 is the
arithmetic behind running them side by side.  usage: [DZ_CUMASK=n] python tools/c5_kernel_times.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
N, d, k, gens = 512, 1000, 5, 100
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
Z0 = np.random.default_rng(3).uniform(-5, 15, (10 * d, d))
e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (3 * gens // 10 + 4), trace_capacity=0, seed=5, history_lag=1)
e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
e.step(gens); e.sync()
t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
e.profile_enable(True, prealloc_pairs=16 * gens); e.profile_reset()
e.step(gens); e.sync()
e.profile_enable(False)
print("DZ_CUMASK=%s  %.1f us per generation (%.2f M proposals/s)" % (os.environ.get("DZ_CUMASK", "-"), 1e6 * dt / gens, N * k * gens / dt / 1e6))
for name in ("propose", "logp", "accept"):
    each = 1e3 * e.profile_get_list(name)
    print("  %-8s %4d launches  median %.1f us  mean %.1f us" % (name, len(each), np.median(each), each.mean()))
