#!/usr/bin/env python3
"""What a drop-in user of the reference's examples sees first: 3 chains x 10-D MVN (BASELINE configs[0] shape), multitry 5, 20000 iterations,
through run_dream -- with the device likelihood, and with the same density as a plain Python function (host callback, one batch per
generation and phase)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd.core import run_dream
from pydream_amd.parameters import FlatParam
from pydream_amd.likelihoods import MVNormalLogLike
d, N, G = 10, 3, 20000
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
logF = np.log(((2 * np.pi) ** (-d / 2.0)) * np.linalg.det(np.linalg.inv(P)) ** -0.5)
def pylike(x):
    return logF - 0.5 * np.sum(x * np.dot(P, x))
seed = np.random.default_rng(1).uniform(-5, 15, (10 * d, d))
np.save("/tmp/_seed_small.npy", seed)
kw = dict(nchains=N, start=[seed[c] for c in range(N)], start_random=False, history_file="/tmp/_seed_small.npy", multitry=5, save_history=False, verbose=False)
run_dream([FlatParam(np.zeros(d))], MVNormalLogLike(P), niterations=200, **kw)
for name, like in (("device likelihood", MVNormalLogLike(P)), ("Python likelihood (host callback)", pylike)):
    t0 = time.perf_counter()
    s, lp = run_dream([FlatParam(np.zeros(d))], like, niterations=G, **kw)
    dt = time.perf_counter() - t0
    print("%-36s %d chains x %d iterations x %d-D: %.2f s = %.0f us per iteration (all chains), %.0f k proposals/s; mean of dim 0 %.2f"
          % (name, N, G, d, dt, 1e6 * dt / G, N * 5 * G / dt / 1e3, np.mean([x[G // 2:, 0].mean() for x in s])))
