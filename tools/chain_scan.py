"""Rate of the persistent kernel by chain count (100-D MVN, multitry 5, no trace buffer): the block-size choice (16 / 8 / 4 chains per block) and
the quantisation by rounds of blocks (one block per CU is resident: 5000 chains are 313 blocks = two rounds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pydream_amd import _capi as G
d, k = 100, 5
i = np.arange(1, d + 1.0)
P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
U = np.linalg.cholesky((P + P.T) / 2).T
for N in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (256, 512, 1024, 1536, 2048, 2560, 3072, 3584, 4096, 5000, 6144, 8192, 16384, 32768)):
    Z0 = np.random.default_rng(3).uniform(-5, 15, (max(1000, 2 * N), d))
    gens = 400
    e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (gens // 10 + 30), trace_capacity=0, seed=5, history_lag=int(os.environ.get('DZ_SCAN_LAG', '1')))
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
    e.step(100); e.sync()
    t0 = time.perf_counter(); e.step(gens); e.sync(); dt = time.perf_counter() - t0
    print("N=%6d  %7.1f M proposals/s  %6.1f us/gen  %s" % (N, N * k * gens / dt / 1e6, 1e6 * dt / gens, e.last_kernel_variant()), flush=True)
    e.close()
