#!/usr/bin/env python3
"""A user-written device likelihood at scale: the 100-D banana (pydream_amd/examples/banana) for 4096 chains x multitry 5 through
  (a) dz_set_likelihood_module -- the user's HIP kernel, one thread per point, launched where the built-in densities' kernels run;
  (b) the same function as a Python callable behind the host callback (numpy, vectorised over the batch);
  (c) for scale: the built-in MVN density on the same multi-kernel path (DZ_MEGA=0) and on the persistent kernel.
Steady-state generations per second of the engine (no trace buffer)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd import _capi as G
from pydream_amd.examples.banana import banana_device as B

N, d, k = 4096, 100, 5
Z0 = np.random.default_rng(3).uniform(-10, 10, (2 * N, d))
like = B.make_likelihood(d)
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T


def rate(name, setup, gens, finite=True):
    e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * ((2 * gens + 200) // 10 + 30), trace_capacity=0, seed=5)
    e.set_history(Z0); e.set_state(Z0[:N]); setup(e)
    e.step(100); e.sync()
    best = 0.0
    for rep in range(2):
        t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
        best = max(best, N * k * gens / dt / 1e6)
    print("%-64s %8.2f M proposals/s  %8.1f us/gen  %s" % (name, best, N * k / best, e.last_kernel_variant()), flush=True)
    e.close()


rate("user kernel (dz_set_likelihood_module), always_finite", lambda e: e.set_likelihood_module(like.code_object(), like.name, 1, like.data, always_finite=True), 1000)
rate("user kernel, redraw check on (a read-back per generation)", lambda e: e.set_likelihood_module(like.code_object(), like.name, 1, like.data, always_finite=False), 300)
rate("the same function through the host callback (numpy)", lambda e: e.set_likelihood_host(B.banana_host_batch), 20)
os.environ["DZ_MEGA"] = "0"
rate("built-in MVN (triangular factor), multi-kernel path", lambda e: e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0), 1000)
os.environ["DZ_MEGA"] = "1"
rate("built-in MVN (triangular factor), persistent kernel", lambda e: e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0), 1000)
