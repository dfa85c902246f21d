#!/usr/bin/env python3
"""Rate of a USER's device likelihood (pydream_amd.likelihoods.DeviceFunctionLogLike: a wave-level HIP device function) inside the persistent
generation kernel against the same function through the multi-kernel path (DZ_MEGA_USER=0), 4096 chains x 100-D, 5 tries; M proposals/s.

    python tools/user_like_rate.py [chains] [d] [tries]
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SRC = r'''
__device__ double weighted_sq(const double* x, int d, const void* data, int lane)
{
    const double* c = (const double*)data; const double* w = c + d;
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) { const double t = x[j] - c[j]; acc = acc + w[j] * (t * t) + 0.001 * ((t * t) * (t * t)); }
    return -0.5 * dz_wave_sum(acc);
}'''


def one(N, d, k):
    from pydream_amd import _capi as G
    from pydream_amd.likelihoods import DeviceFunctionLogLike
    c = np.linspace(-2.0, 2.0, d); w = 0.5 + np.arange(d) % 7 / 7.0
    like = DeviceFunctionLogLike(SRC, "weighted_sq", d, data=np.concatenate([c, w]), always_finite=True)
    gens = 400
    Z0 = np.random.default_rng(2).uniform(-6, 6, (10 * d + 2 * N, d))
    e = G.Engine(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (gens // 10 + 60), trace_capacity=gens + 100, seed=3, history_lag=1)
    e.set_history(Z0); e.set_state(Z0[:N]); like._dz_apply(e)
    e.step(100); e.sync()
    t = time.time(); e.step(gens); e.sync(); dt = time.time() - t
    rate = N * k * gens / dt / 1e6          # bench.py's convention: multitry proposals per chain and generation
    print("%-28s %7.1f M proposals/s  (%.1f us per generation)" % (e.last_kernel_variant(), rate, dt / gens * 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(*[int(x) for x in sys.argv[2:5]])
    else:
        N, d, k = [int(x) for x in (sys.argv[1:4] + ["4096", "100", "5"][len(sys.argv) - 1:])]
        for mode in ("1", "0"):
            subprocess.run([sys.executable, __file__, "--one", str(N), str(d), str(k)], env=dict(os.environ, DZ_MEGA_USER=mode), check=True)
