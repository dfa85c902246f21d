"""bench.py's load order for N > 1 (engine first, torch second): which HIP runtime and which librccl the engine ends up with,
and an RCCL all-gather with world size 1 through them (used by tests/test_distributed.py)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydream_amd import _capi           # first: binds /opt/rocm's HIP runtime
print("hip:", _capi.hip_library(), "rccl:", _capi.comm_library())
import torch                            # afterwards, as bench.py does for N > 1
from tests import helpers as H
d, N, n = 16, 8, 25
P = H.mvn_precision(d); Z0 = H.seed_history(40, d, 4)
res = []
for use_comm in (False, True):
    e = _capi.Engine(nchains=N, ndim=d, multitry=5, history_capacity=40 + N * 8, trace_capacity=n, seed=5, history_thin=5)
    if use_comm:
        e.comm_init_rccl(0, 1, _capi.comm_unique_id())
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), P, 0, 0.0)
    e.step(n)
    res.append((e.get_trace(0, n)["X"], e.get_history()))
print("equal:", np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]))
