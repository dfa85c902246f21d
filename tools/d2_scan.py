import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pydream_amd import _capi as G
N = 4096
for d, tri in ((200, 1), (200, 0), (160, 0), (208, 1), (224, 1)):
    i = np.arange(1, d + 1.0)
    P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = np.random.default_rng(3).uniform(-5, 15, (2 * N, d))
    for env in ("1", "0"):
        os.environ["DZ_MEGA_D2"] = env
        e = G.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * 200, trace_capacity=0, seed=5)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U if tri else P, tri, 0.0)
        e.step(300); e.sync()
        reps = []
        for rep in range(6):
            t0 = time.perf_counter(); e.step(200); e.sync(); reps.append((time.perf_counter() - t0) / 200)
        print("d=%d tri=%d D2=%s  %s   passes %s us/gen" % (d, tri, env, e.last_kernel_variant(), " ".join("%.1f" % (1e6 * r) for r in reps)), flush=True)
        e.close()
