#!/usr/bin/env python3
"""End-to-end rate of the examples' convergence loop (dream_ex_ndim_gaussian.py:79-102) through run_dream(): host arrays in, every sample
returned to the host, R-hat after every call, three restart rounds -- PCIe-, file- and host-inclusive.

    python tools/run_dream_rate.py [chains] [iterations] [multitry]      (DREAMZS_KEEP_ENGINE=0: every restart rebuilds from the .npy files)
"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd.core import run_dream, release_engines
from pydream_amd.parameters import FlatParam
from pydream_amd.likelihoods import MVNormalLogLike
from pydream_amd.convergence import Gelman_Rubin, Gelman_Rubin_device

N, d, G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 100, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5          # multitry (1: off, the reference's default)
os.chdir(tempfile.mkdtemp(prefix="dz_rate_"))
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
seed = np.random.default_rng(1).uniform(-5, 15, (max(10 * d, 2 * N), d))
np.save("seed.npy", seed)
params, like = [FlatParam(np.zeros(d))], MVNormalLogLike(P)
kw = dict(nchains=N, start_random=False, multitry=(K if K > 1 else False), verbose=False)
run_dream(params, like, niterations=200, start=[seed[c] for c in range(N)], history_file="seed.npy", save_history=False, **kw)      # warm-up (clocks, library load)
keep = os.environ.get("DREAMZS_KEEP_ENGINE", "1") != "0"
t_all = time.perf_counter()
t0 = time.perf_counter()
sampled, log_ps = run_dream(params, like, niterations=G, start=[seed[c] for c in range(N)], history_file="seed.npy", save_history=True, model_name="rate", **kw)
t_run = time.perf_counter() - t0
t0 = time.perf_counter(); r_host = Gelman_Rubin(sampled); t_rh = time.perf_counter() - t0
t0 = time.perf_counter(); r_dev = Gelman_Rubin_device(sampled); t_rd = time.perf_counter() - t0
print("run_dream (multitry %d): %d chains x %d iterations in %.2f s = %.1f M proposals/s end to end incl. the history files (%.0f us per generation); "
      "R-hat max %.3f: host Gelman_Rubin %.3f s (in place, no copy), device result %.6f s, |difference| %.1e"
      % (K, N, G, t_run, N * K * G / t_run / 1e6, 1e6 * t_run / G, r_host.max(), t_rh, t_rd, np.abs(r_host - r_dev).max()))
for rnd in range(3):
    starts = [s[-1, :] for s in sampled]
    t0 = time.perf_counter()
    sampled, log_ps = run_dream(params, like, niterations=G, start=starts, restart=True, save_history=True, model_name="rate", **kw)
    t_run = time.perf_counter() - t0
    t0 = time.perf_counter(); r = Gelman_Rubin_device(sampled); t_r = time.perf_counter() - t0
    print("  restart %d (%s): %.2f s = %.1f M proposals/s; archive now %d rows; R-hat max %.3f (%.6f s)"
          % (rnd + 1, "live engine" if keep else "rebuilt from the .npy files", t_run, N * K * G / t_run / 1e6,
             os.path.getsize("rate_DREAM_chain_history.npy") // (8 * d), r.max(), t_r))
print("whole loop (4 x %d iterations, 4 R-hat evaluations, history files written 4 times): %.2f s" % (G, time.perf_counter() - t_all))
release_engines()
