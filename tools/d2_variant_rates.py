#!/usr/bin/env python3
"""k_generations_d2's full-code instantiations (SampledParam priors, hard boundaries, several DE pairs) at 4096 chains x 200-D, multitry 5,
against the multi-kernel path (DZ_MEGA_D2=0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
N, d, k = 4096, 200, 5
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
Z0 = np.random.default_rng(3).uniform(-5, 15, (6 * N, d))
for name, kw, prior in (("flat", {}, None), ("uniform priors + hard boundaries", {}, "uniform"), ("normal priors", {}, "normal"), ("DEpairs = 3", dict(depairs=3), None),
                        ("uniform priors + hard boundaries, multitry off", dict(multitry=1), "uniform")):
    for env in ("1", "0"):
        os.environ["DZ_MEGA_D2"] = env
        a = dict(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * 200, trace_capacity=0, seed=5); a.update(kw)
        e = G.Engine(**a)
        if prior == "uniform":
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -40.0), np.full(d, 80.0)); e.set_bounds(np.full(d, -40.0), np.full(d, 40.0))
        elif prior == "normal":
            e.set_prior(np.full(d, 1, np.int32), np.zeros(d), np.full(d, 30.0))
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(300); e.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); e.step(200); e.sync(); best = min(best, (time.perf_counter() - t0) / 200)
        kk = a["multitry"]
        print("%-52s %7.1f M proposals/s  %6.1f us/gen  %s" % (name, N * kk / best / 1e6, 1e6 * best, e.last_kernel_variant()), flush=True)
        e.close()
