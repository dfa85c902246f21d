#!/usr/bin/env python3
"""Where run_dream()'s end-to-end time goes on the host (cProfile over one 4096-chain x 2000-iteration restart call on the live engine,
history files written, after a first call)."""
import os, sys, time, cProfile, pstats, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd.core import run_dream, release_engines
from pydream_amd.parameters import FlatParam
from pydream_amd.likelihoods import MVNormalLogLike
N, d, G, K = 4096, 100, 2000, 5
os.chdir(tempfile.mkdtemp(prefix="dz_prof_"))
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
seed = np.random.default_rng(1).uniform(-5, 15, (max(10 * d, 2 * N), d))
np.save("seed.npy", seed)
params, like = [FlatParam(np.zeros(d))], MVNormalLogLike(P)
kw = dict(nchains=N, start_random=False, multitry=K, verbose=False, save_history=True, model_name="prof")
s, _ = run_dream(params, like, niterations=G, start=[seed[c] for c in range(N)], history_file="seed.npy", **kw)
starts = [x[-1] for x in s]
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
sampled, log_ps = run_dream(params, like, niterations=G, start=starts, restart=True, **kw)
dt = time.perf_counter() - t0
pr.disable()
print("total", dt)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
release_engines()
