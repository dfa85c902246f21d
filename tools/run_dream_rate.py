#!/usr/bin/env python3
"""End-to-end rate of run_dream() (host arrays in, every sample returned to the host): PCIe- and host-inclusive."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd.core import run_dream
from pydream_amd.parameters import FlatParam
from pydream_amd.likelihoods import MVNormalLogLike
from pydream_amd.convergence import Gelman_Rubin

N, d, G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 100, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5          # multitry (1: off, the reference's default)
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
seed = np.random.default_rng(1).uniform(-5, 15, (max(10 * d, 2 * N), d))
np.save("/tmp/_seed.npy", seed)
kw = dict(nchains=N, start=[seed[c] for c in range(N)], start_random=False, history_file="/tmp/_seed.npy",
          multitry=(K if K > 1 else False), save_history=False, verbose=False)
run_dream([FlatParam(np.zeros(d))], MVNormalLogLike(P), niterations=200, **kw)          # warm-up (clocks, library load)
t0 = time.perf_counter()
sampled, log_ps = run_dream([FlatParam(np.zeros(d))], MVNormalLogLike(P), niterations=G, **kw)
dt = time.perf_counter() - t0
print("run_dream (multitry %d): %d chains x %d iterations in %.2f s = %.1f M proposals/s end to end (%.0f us per generation); R-hat max %.2f"
      % (K, N, G, dt, N * K * G / dt / 1e6, 1e6 * dt / G, Gelman_Rubin(sampled[:64]).max()))
