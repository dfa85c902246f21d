#!/usr/bin/env python3
"""One rank of a BASELINE multi-GPU configuration at full size, for tests/test_distributed.py:

    RANK=r WORLD_SIZE=W MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/shard_rank.py <config> <outdir> [transport] [lag] [generations]

config "c3": configs[3], 32768 chains x 100-D MVN (W ranks of 32768 / W);  "c4": configs[4], 4096 chains x 1000-D correlated MVN
(triangular factor);  "c3s" / "c4s": the same shapes at an eighth of the chains (quick local runs).  Rank r owns chains
[r N / W, (r + 1) N / W) on device DZ_SHARD_DEVICE (default 0: the ranks share the test box's one GPU), attaches the transport,
steps the generations and leaves rank<r>.npz: final states, cached log densities, the decision flags of every generation, the archive's
checksum and row count (the archive itself from rank 0 only).  W = 1 is the unsharded comparand (`build` is imported by the test).
"""
import os
import sys

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):       # (eight ranks on one box: no BLAS thread per core each)
    os.environ.setdefault(_v, "2")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {   # (chains, d, mvn kind)
    "c3": (32768, 100, "tri"), "c4": (4096, 1000, "tri"),
    "c3s": (4096, 100, "tri"), "c4s": (512, 1000, "tri"),
    # crossover adaptation on (the reference's default, Dream.py:63-67): a burn-in of BURNIN generations inside the run, so that the ranks
    # exchange adaptation statistics every generation -- group sums where a rank owns whole groups of 256 chains, positions otherwise
    "a512": (512, 100, "tri"), "a1k": (1024, 100, "tri"), "a2k": (2048, 100, "tri"), "a2k_mix": (2048, 100, "mix"), "a512_k1": (512, 100, "tri"),
    "a768": (768, 100, "tri"),        # 2 ranks x 384 chains: NOT whole groups -- the positions travel (the round-4 path)
    "a2k_d200": (2048, 200, "tri"),   # the multi-kernel path's generations (d > 128) with the group exchange
    "a8k": (8192, 100, "tri"),        # 2 ranks x 4096 chains: blocks of 16 chains (with an adapt_lag: the kernels' own unit sums, several generations per launch)
}
ADAPT = {"a512": 5, "a1k": 5, "a2k": 5, "a2k_mix": 5, "a512_k1": 1, "a768": 5, "a2k_d200": 5, "a8k": 5}       # config -> multitry
BURNIN = 24
SEED, K, THIN = 20260930, 5, 10


def matrix(config):
    """the likelihood's matrix, made ONCE per test (by the parent, handed to the ranks as a file): LAPACK's inverse and Cholesky factor
    differ in the last bits between thread counts, and ranks that do not hold the same matrix bit for bit are not one run"""
    from tests import helpers as H
    N, d, kind = CONFIGS[config]
    P = H.mvn_precision(d)
    return np.linalg.cholesky((P + P.T) / 2).T if kind != "dense" else P


def build(config, rank, world, generations, lag, device=0, M=None, engine_cls=None):
    """engine_cls: the test's comparand may be the ORACLE (tests only) instead of the HIP engine"""
    if engine_cls is None:
        from pydream_amd import _capi
        engine_cls = _capi.Engine
    N, d, kind = CONFIGS[config]
    nl = N // world
    m0 = max(10 * d, 2 * N)                                  # Dream.py:168-170, core.py:270-273
    Z0 = np.random.default_rng(SEED).uniform(-5.0, 15.0, (m0, d))
    extra = dict(adapt_crossover=1, crossover_burnin=BURNIN, adapt_lag=int(os.environ.get("DZ_TEST_ADAPT_LAG", "0"))) if config in ADAPT else {}
    e = engine_cls(nchains=N, nchains_local=nl, chain_offset=rank * nl, ndim=d, multitry=ADAPT.get(config, K), history_thin=THIN,
                   history_capacity=m0 + N * (generations // THIN + 2), trace_capacity=generations, seed=SEED, device=device,
                   history_lag=lag, **extra)
    e.set_history(Z0)
    e.set_state(Z0[rank * nl:(rank + 1) * nl])
    if kind == "mix":
        mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
        e.set_likelihood_mixture(mu, np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi))
    else:
        e.set_likelihood_mvn(np.zeros(d), matrix(config) if M is None else M, 1 if kind == "tri" else 0, 0.0)
    return e


def results(e, generations, with_history):
    X, pr, lk = e.get_state()
    tr = e.get_trace(0, generations, with_X=False)
    h, rows = e.history_checksum()
    out = dict(X=X, prior=pr, like=lk, logp=tr["logp"], moved=tr["moved"], try_idx=tr["try_idx"], cr_idx=tr["cr_idx"], snooker=tr["snooker"],
               checksum=np.array([h], dtype=np.uint64), rows=np.array([rows]), variant=np.array(e.last_kernel_variant()),
               cr_probs=e.get_cr_state()[0], cr_delta=e.get_cr_state()[1], cr_n=e.get_cr_state()[2],
               xbytes=np.array(e.exchange_bytes() if hasattr(e, "exchange_bytes") else (0, 0, 0), dtype=np.int64))
    if with_history:
        out["Z"] = e.get_history()
    return out


def main():
    import faulthandler
    faulthandler.dump_traceback_later(280, exit=True)          # a rank that is stuck says where and leaves
    config, outdir = sys.argv[1], sys.argv[2]
    transport = sys.argv[3] if len(sys.argv) > 3 else "peer"
    lag = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    G = int(sys.argv[5]) if len(sys.argv) > 5 else 25
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from pydream_amd.distributed import attach_transport, socket_group_from_env
    group = socket_group_from_env(timeout=240.0)
    e = build(config, rank, world, G, lag, device=int(os.environ.get("DZ_SHARD_DEVICE", "0")), M=np.load(os.path.join(outdir, "matrix.npy")))
    attach_transport(e, rank, world, transport=transport, group=group)
    e.profile_enable(True); e.profile_reset()
    e.step(G)
    launches = e.profile_get("generations")[1]          # launches of the persistent kernels
    e.profile_enable(False)
    out = results(e, G, with_history=(rank == 0))
    out["launches"] = np.array([launches])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    e.sync()
    group.barrier()                  # a rank's buffers stay mapped until no peer can still be writing into them
    e.close()
    group.close()


if __name__ == "__main__":
    main()
