"""The N>1 path: chains sharded over ranks, Z replicated on every rank.

CPU (gloo, world_size 2): the host logic (shard arithmetic, seed agreement, exchange hook, per-rank
assembly) with the ORACLE standing in for the device engine -- sharded == unsharded, bit for bit.
GPU (marked gpu): the same with the HIP engine, the ranks sharing the one GPU of the test box, over the
host-staged transport and over the PEER transport (IPC-mapped archives, copy-stream pushes, gate kernels)
with history_lag 0 and 1; BASELINE configs[3] and configs[4] at full size (8 ranks) against the unsharded
engine; bench.py's N > 1 line incl. its replica check; RCCL with one rank (all a 1-GPU box can run).
"""
import os
import time
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(d, variant="flat"):
    from tests import helpers as H
    from pydream_amd.likelihoods import MVNormalLogLike
    from pydream_amd.parameters import FlatParam, SampledParam
    like = MVNormalLogLike(H.mvn_precision(d), factorize=False)
    if variant == "redraw":      # a uniform prior narrower than the seed archive, no hard boundaries: impossible proposal sets are
        from scipy.stats import uniform                      # drawn again (Dream.py:281-289), on each rank for its own chains
        return [SampledParam(uniform, loc=np.full(d, -3.0), scale=np.full(d, 8.0))], like
    return [FlatParam(np.zeros(d))], like


def _starts(Z0, variant):
    X = Z0[:8]
    return [(-3.0 + 8.0 * (x + 5.0) / 20.0) if variant == "redraw" else x for x in X]      # (inside the prior's support)


def _kw(variant):
    if variant == "redraw":
        return dict(KW, hardboundaries=False)
    if variant in ("lag1", "peer_lag1"):          # appended rows become sampleable one append late (dz_config.history_lag)
        return dict(KW, history_lag=1)
    return dict(KW)


def _transport(variant):
    return "peer" if variant.startswith("peer") else "host"


KW = dict(nchains=8, niterations=45, multitry=5, adapt_crossover=True, crossover_burnin=20, save_history=False, seed=77,
          nseedchains=40, history_thin=5)


def _worker(rank, world, port, backend_engine, outdir, variant="flat", token=None):
    sys.path.insert(0, ROOT)
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)           # a rank that is stuck says where and leaves
    from pydream_amd.distributed import SocketGroup, run_dream_sharded
    if backend_engine == "oracle":                              # CPU: torch.distributed / gloo as the control plane
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        group = None
    else:                                                       # GPU box: the built-in socket group (no torch in the process: its first
        group = SocketGroup(rank, world, "127.0.0.1", port, token=token)     # import on a freshly started box takes minutes)
    from tests import helpers as H
    d = 12
    params, like = _model(d, variant)
    if backend_engine.startswith("oracle"):
        from oracle import oracle as O
        cls = O.Engine
    else:
        cls = None
    Z0 = H.seed_history(40, d, 3)
    hist = os.path.join(outdir, "seed_%d.npy" % rank)
    np.save(hist, Z0)
    sampled, log_ps = run_dream_sharded(params, like, start=_starts(Z0, variant), history_file=hist, transport=_transport(variant),
                                        engine_cls=cls, device=0, group=group, **_kw(variant))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), X=np.array(sampled), lp=np.array(log_ps))
    if group is None:
        dist.destroy_process_group()
    else:
        group.close()


def _single(backend_engine, outdir, variant="flat"):
    from pydream_amd import core
    from pydream_amd.Dream import Dream
    from pydream_amd.model import Model
    from tests import helpers as H
    d = 12
    params, like = _model(d, variant)
    Z0 = H.seed_history(40, d, 3)
    hist = os.path.join(outdir, "seed_single.npy")
    np.save(hist, Z0)
    kw = _kw(variant)
    n, it, seed = kw.pop("nchains"), kw.pop("niterations"), kw.pop("seed")
    lag = kw.pop("history_lag", 0)
    step = Dream(model=Model(like, params), history_file=hist, **kw)
    cls = None
    if backend_engine.startswith("oracle"):
        from oracle import oracle as O
        cls = O.Engine
    pool = core._setup_mp_dream_pool(n, it, step, start_pt=_starts(Z0, variant), seed=seed, engine_cls=cls, history_lag=lag)
    try:
        s, l = core._sample_dream_batched(pool.engine, step, it, False, 10)
        Z = pool.engine.get_history()
        cr = pool.engine.get_cr_state()[0]
        redraws = pool.engine.redraw_rounds() if hasattr(pool.engine, "redraw_rounds") else None
    finally:
        pool.close(); pool.join()
    return np.array(s), np.array(l), Z, cr, redraws


def _run_two_ranks(backend_engine, tmp_path, variant="flat"):
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    from pydream_amd.distributed import new_token
    token = new_token()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend_engine, str(tmp_path), variant, token)) for r in range(2)]
    for pr in procs:
        pr.start()
    deadline = time.time() + 240                     # (a rank that never returns fails the test instead of hanging the suite)
    while any(pr.is_alive() for pr in procs):
        if time.time() > deadline:
            for pr in procs:
                if pr.is_alive():
                    pr.kill()
            raise AssertionError("the two-rank run did not finish in time")
        time.sleep(0.1)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    X = np.concatenate([r0["X"], r1["X"]]); lp = np.concatenate([r0["lp"], r1["lp"]])
    Xs, lps, _, cr, redraws = _single(backend_engine, str(tmp_path), variant)
    np.testing.assert_array_equal(X, Xs)
    np.testing.assert_array_equal(lp, lps)
    assert not np.allclose(cr, 1 / 3.)            # adaptation ran (and was identical on both ranks, or the traces would differ)
    if variant == "redraw":
        assert np.all(np.isfinite(lps)) and np.all(Xs >= -3.0) and np.all(Xs <= 5.0)
        assert redraws is None or redraws > 0     # (the HIP engine counts its redraw launches)


def _restart_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from oracle import oracle as O
    from pydream_amd.distributed import run_dream_sharded
    from tests import helpers as H
    os.chdir(outdir)
    d = 12
    params, like = _model(d)
    Z0 = H.seed_history(40, d, 3)
    np.save("seed_r%d.npy" % rank, Z0)
    kw = dict(nchains=8, multitry=5, adapt_crossover=True, history_thin=5, engine_cls=O.Engine, transport="host", device=0)
    s1, l1 = run_dream_sharded(params, like, niterations=30, start=list(Z0[:8]), history_file="seed_r%d.npy" % rank, nseedchains=40, seed=5,
                               model_name="shard", save_history=True, **kw)
    last = np.zeros((8, d))
    last[rank * 4:(rank + 1) * 4] = np.array(s1)[:, -1]
    import torch
    t = torch.from_numpy(last); dist.all_reduce(t)                      # every rank needs ALL chains' last states as `start`
    s2, l2 = run_dream_sharded(params, like, niterations=25, start=list(t.numpy()), restart=True, seed=6, model_name="shard", save_history=True, **kw)
    np.savez("restart_rank%d.npz" % rank, X1=np.array(s1), X2=np.array(s2), L2=np.array(l2))
    dist.destroy_process_group()


def test_two_ranks_restart_equals_the_unsharded_restart(tmp_path):
    """run_dream_sharded(restart=True) (pydream/core.py:46-62, :255-263; Dream.py:128-147): a first sharded run leaves the three files, a
    second one -- two ranks again -- loads history and adapted probabilities from them and continues; equal, bit for bit, to the same two
    calls of run_dream on one engine (oracle backend, gloo control plane)."""
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_restart_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    r0, r1 = np.load(tmp_path / "restart_rank0.npz"), np.load(tmp_path / "restart_rank1.npz")
    # the unsharded comparand: the same two calls through run_dream's own machinery on one oracle engine
    from oracle import oracle as O
    from pydream_amd import core
    from pydream_amd.Dream import Dream
    from pydream_amd.model import Model
    from tests import helpers as H
    d = 12
    params, like = _model(d)
    Z0 = H.seed_history(40, d, 3)
    cwd = os.getcwd()
    os.makedirs(tmp_path / "single", exist_ok=True)
    os.chdir(tmp_path / "single")
    try:
        np.save("seed.npy", Z0)
        common = dict(multitry=5, adapt_crossover=True, history_thin=5, model_name="one", save_history=True)
        step = Dream(model=Model(like, params), history_file="seed.npy", nseedchains=40, **common)
        pool = core._setup_mp_dream_pool(8, 30, step, start_pt=list(Z0[:8]), seed=5, engine_cls=O.Engine)
        s1, _ = core._sample_dream_batched(pool.engine, step, 30, False, 10)
        pool.close(); pool.join()
        step2 = Dream(model=Model(like, params), history_file="one_DREAM_chain_history.npy", crossover_file="one_DREAM_chain_adapted_crossoverprob.npy",
                      gamma_file="one_DREAM_chain_adapted_gammalevelprob.npy", **common)
        pool = core._setup_mp_dream_pool(8, 25, step2, start_pt=[x[-1] for x in s1], seed=6, engine_cls=O.Engine)
        s2, l2 = core._sample_dream_batched(pool.engine, step2, 25, False, 10)
        pool.close(); pool.join()
    finally:
        os.chdir(cwd)
    np.testing.assert_array_equal(np.concatenate([r0["X1"], r1["X1"]]), np.array(s1))
    np.testing.assert_array_equal(np.concatenate([r0["X2"], r1["X2"]]), np.array(s2))
    np.testing.assert_array_equal(np.concatenate([r0["L2"], r1["L2"]]), np.array(l2))
    np.testing.assert_array_equal(np.load(tmp_path / "shard_DREAM_chain_history.npy"), np.load(tmp_path / "single" / "one_DREAM_chain_history.npy"))


def test_shard_arithmetic():
    from pydream_amd.distributed import shard
    assert shard(32768, 3, 8) == (12288, 4096)
    with pytest.raises(Exception):
        shard(10, 0, 4)


@pytest.mark.parametrize("variant", ["flat", "redraw", "lag1"])
def test_two_ranks_gloo_oracle_backend(tmp_path, variant):
    _run_two_ranks("oracle", tmp_path, variant)


def test_two_ranks_socket_group_oracle_backend(tmp_path):
    """the built-in control plane (pydream_amd.distributed.SocketGroup: what bench.py and the GPU tests rendezvous over) instead of gloo"""
    _run_two_ranks("oracle_socket", tmp_path, "lag1")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["flat", "redraw", "lag1", "peer", "peer_lag1"])
def test_two_ranks_one_gpu_hip_engine(tmp_path, variant):
    _run_two_ranks("hip", tmp_path, variant)


def _launch_ranks(script_args, nranks, env, cwd, timeout):
    """N processes of a script the way torch.distributed.run starts them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), without that launcher; every one must return 0 in time."""
    import subprocess
    port = str(_free_port())
    procs = []
    for r in range(nranks):
        renv = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable] + script_args, cwd=cwd, env=renv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    deadline = time.time() + timeout
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the ranks did not finish in time")
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0, se[-3000:]
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("config,transport,lag", [("c3", "peer", 3), ("c3", "peer", 1), ("c4", "peer", 1), ("c4", "peer", 3), ("c3", "peer", 0), ("c4", "host", 0), ("c3", "host", 1)])
def test_baseline_multi_gpu_configs_at_full_size_equal_the_unsharded_engine(tmp_path, config, transport, lag):
    """BASELINE configs[3] (8 ranks x 4096 chains x 100-D MVN = 32768 chains) and configs[4] (8 x 512 chains x 1000-D correlated MVN)
    AS WRITTEN, the eight ranks time-sharing the test box's one MI355X: every rank's states, cached log densities and decision
    sequences over 25 generations -- three history appends, the rows of two of them sampled by later generations -- equal the
    corresponding slice of ONE engine that holds all chains, bit for bit, and every rank's replica of the archive (device checksum)
    equals the unsharded archive.  (What replaces the shared arrays of core.py:281-297 / Dream.py:919-945.  The per-GPU shards are
    compared with the oracle in test_gpu_parity.py; at these sizes the oracle would take minutes, so the comparand is the unsharded
    engine, which runs different block sizes and, at 1000-D, different tile shapes of the likelihood product than the shards do.)"""
    from pydream_amd import _capi
    from tests import shard_rank as SR
    # (round 6) lag 3 over the peer transport, lag 1 through the host: sharded launches that run on past an append -- two per launch, as on one
    # GPU (dz_engine.hip mega_appends_per_launch) -- over 45 generations, the rows of the launches' first appends among those sampled
    G, W = (45 if (lag == 3 or (transport == "host" and lag == 1)) and config == "c3" else 25), 8
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="200", DZ_SHARD_DEVICE="0")
    M = SR.matrix(config)
    np.save(tmp_path / "matrix.npy", M)
    _launch_ranks([os.path.join(ROOT, "tests", "shard_rank.py"), config, str(tmp_path), transport, str(lag), str(G)], W, env, str(tmp_path), timeout=300)
    e = SR.build(config, 0, 1, G, lag, M=M)
    e.step(G)
    ref = SR.results(e, G, with_history=True)
    e.close()
    N, d, _ = SR.CONFIGS[config]
    assert ref["Z"].shape == (max(10 * d, 2 * N) + ((G - 1) // 10 + 1) * N, d) and ref["moved"].mean() > 0.02
    assert int(ref["checksum"][0]) == _capi.history_checksum_host(ref["Z"])          # the device checksum is the documented sum
    nl = N // W
    for r in range(W):
        got = np.load(tmp_path / ("rank%d.npz" % r))
        sl = slice(r * nl, (r + 1) * nl)
        for key in ("X", "prior", "like"):
            np.testing.assert_array_equal(got[key], ref[key][sl], err_msg="%s of rank %d" % (key, r))
        for key in ("logp", "moved", "try_idx", "cr_idx", "snooker"):
            np.testing.assert_array_equal(got[key], ref[key][:, sl], err_msg="%s of rank %d" % (key, r))
        assert int(got["rows"][0]) == len(ref["Z"]) and int(got["checksum"][0]) == int(ref["checksum"][0]), "archive replica of rank %d" % r
        if config == "c3":       # generations 0 | 1-10 | 11-20 | ... one append per launch; with two per launch once `lag` appends are made: 0 | 1-10 | 11-20 | 21-40 | 41-45 (lag 3)
            assert int(got["launches"][0]) == {(25, 0): 4, (25, 1): 4, (45, 3): 5, (45, 1): 4}[(G, lag)], (int(got["launches"][0]), G, lag)
        if r == 0:
            np.testing.assert_array_equal(got["Z"], ref["Z"])


@pytest.mark.gpu
@pytest.mark.parametrize("config,world,transport,lag,against,env", [
    ("a512", 2, "peer", 1, "oracle", {}),                           # one group of 256 chains per rank, fused unit sums inside k_generations
    ("a1k", 2, "peer", 1, "oracle", {}),                            # >= 1024 chains, 2 ranks, peer, lag 1, against the ORACLE (round-4 verdict 1c)
    ("a1k", 4, "host", 0, "oracle", {"DZ_ADAPT_FUSED": "0"}),       # the units' sums by k_adapt_partials over the rank's OWN units
    ("a512_k1", 2, "peer", 0, "oracle", {}),                        # multitry off: no fused sums
    ("a2k", 8, "peer", 1, "engine", {}),                            # eight ranks
    ("a2k_mix", 8, "peer", 0, "engine", {}),                        # the mixture kernel's 16-wave burn-in blocks
    ("a2k_d200", 8, "peer", 1, "engine", {}),                       # d > 128: the multi-kernel path's generations
    ("a768", 2, "peer", 1, "oracle", {}),                           # 384 chains per rank: not whole groups, the positions travel
    ("a1k", 2, "peer", 1, "engine", {"DZ_ADAPT_GROUPS": "0"})])     # the position exchange forced where groups would do
def test_sharded_crossover_burnin_exchanges_group_sums(tmp_path, config, world, transport, lag, against, env):
    """The crossover burn-in of a sharded run (estimate_crossover_probabilities, Dream.py:451-499, over ALL chains' positions -- the shared
    current_positions array of core.py:296): a rank that owns whole groups of 256 chains sends its groups' column sums (reduction
    contract v3: units of 16 chains -> groups of 16 units -> total, in order) instead of its positions, every rank adds all groups in
    order -- the same totals bit for bit at every world size.  Checked: every rank's states, log densities, decision sequences, adapted
    probabilities AND their accumulators (delta_m, ncr_updates) over a run whose burn-in (24 generations) ends inside it, against the
    ORACLE (<= 1024 chains) or the unsharded engine; all archive replicas identical; the bytes per rank and burn-in generation
    (dz_exchange_bytes): group sums, and the positions only once."""
    from tests import shard_rank as SR
    G = 37
    renv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="200", DZ_SHARD_DEVICE="0", **env)
    M = SR.matrix(config)
    np.save(tmp_path / "matrix.npy", M)
    _launch_ranks([os.path.join(ROOT, "tests", "shard_rank.py"), config, str(tmp_path), transport, str(lag), str(G)], world, renv, str(tmp_path), timeout=300)
    N, d, _ = SR.CONFIGS[config]
    if against == "oracle":
        from oracle import oracle as O
        o = SR.build(config, 0, 1, G, lag, M=M, engine_cls=O.Engine)
        o.step(G)
        X, pr, lk = o.get_state()
        tr = o.get_trace(0, G)
        ref = dict(X=X, prior=pr, like=lk, logp=tr["logp"], moved=tr["moved"], try_idx=tr["try_idx"], cr_idx=tr["cr_idx"], snooker=tr["snooker"], Z=o.get_history())
        ref["cr_probs"], ref["cr_delta"], ref["cr_n"] = o.get_cr_state()
    else:
        e = SR.build(config, 0, 1, G, lag, M=M)
        e.step(G)
        ref = SR.results(e, G, with_history=True)
        e.close()
    assert not np.allclose(ref["cr_probs"], 1 / 3.) and ref["cr_n"].sum() > N          # the adaptation did run
    nl = N // world
    sums0 = None
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npz" % r))
        sl = slice(r * nl, (r + 1) * nl)
        for key in ("X", "prior", "like"):
            np.testing.assert_array_equal(got[key], ref[key][sl], err_msg="%s of rank %d" % (key, r))
        for key in ("logp", "moved", "try_idx", "cr_idx", "snooker"):
            np.testing.assert_array_equal(got[key], ref[key][:, sl], err_msg="%s of rank %d" % (key, r))
        for key in ("cr_probs", "cr_delta", "cr_n"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg="%s of rank %d" % (key, r))
        if r == 0:
            np.testing.assert_array_equal(got["Z"], ref["Z"])
            sums0 = (int(got["checksum"][0]), int(got["rows"][0]))
        assert (int(got["checksum"][0]), int(got["rows"][0])) == sums0, "archive replica of rank %d" % r
        zb, pb, sb = (int(x) for x in got["xbytes"])
        ld = (d + 15) // 16 * 16
        assert zb == 4 * nl * ld * 8                                                        # four appends in 37 generations
        groups = nl % 256 == 0 and env.get("DZ_ADAPT_GROUPS") != "0"
        if groups:          # 25 burn-in generations of (2 + nCR + ngamma) ld + 16 doubles per group, + one row; the positions once (generation 0's start)
            assert pb == nl * ld * 8 and sb == 25 * ((nl // 256) * (6 * ld + 16) + ld) * 8
            assert sb // 25 <= 100 * 1024 * max(1, nl // 4096 + (nl % 4096 > 0))
        else:
            assert sb == 0 and pb == 26 * nl * ld * 8


@pytest.mark.gpu
@pytest.mark.parametrize("config,world,transport,lag,alag,against,kind", [
    ("a512", 2, "peer", 1, 3, "oracle", "ring"),        # 256 chains per rank (blocks of 4 chains x 4 waves): every generation's positions into the ring, the unit sums behind the launch
    ("a2k", 2, "peer", 1, 9, "oracle", "ring"),         # 1024 chains per rank, ten generations per launch
    ("a2k_mix", 2, "host", 0, 5, "engine", "multi"),    # the mixture kernel's blocks of 16 make their unit sums themselves; host transport
    ("a2k", 8, "peer", 3, 19, "engine", "ring"),        # eight ranks, twenty generations per launch where the history appends allow
    ("a8k", 2, "peer", 3, 19, "engine", "multi"),       # 4096 chains per rank: k_generations<..,multi>
    ("a768", 2, "peer", 1, 3, "oracle", "single"),      # not whole groups: the positions travel, one burn-in generation per launch
])
def test_sharded_crossover_burnin_with_an_adapt_lag(tmp_path, monkeypatch, config, world, transport, lag, alag, against, kind):
    """dz_config.adapt_lag on several GPUs (round 6): ranks that own whole groups of 256 chains run up to adapt_lag + 1 burn-in generations per launch
    like one GPU does -- the group sums of ALL the launch's generations travel in one exchange (adapt_finish_groups), every rank forms every
    generation's totals from all ranks' records in order and applies them when they are due.  Every rank's states, log densities, decisions,
    adapted probabilities and accumulators equal the oracle's (or the unsharded engine's) bit for bit, all archive replicas are identical, and
    the burn-in took fewer launches than generations."""
    from tests import shard_rank as SR
    G = 47
    monkeypatch.setenv("DZ_TEST_ADAPT_LAG", str(alag))
    renv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="200", DZ_SHARD_DEVICE="0", DZ_TEST_ADAPT_LAG=str(alag))
    M = SR.matrix(config)
    np.save(tmp_path / "matrix.npy", M)
    _launch_ranks([os.path.join(ROOT, "tests", "shard_rank.py"), config, str(tmp_path), transport, str(lag), str(G)], world, renv, str(tmp_path), timeout=300)
    N, d, _ = SR.CONFIGS[config]
    if against == "oracle":
        from oracle import oracle as O
        o = SR.build(config, 0, 1, G, lag, M=M, engine_cls=O.Engine)
        o.step(G)
        X, pr, lk = o.get_state()
        tr = o.get_trace(0, G)
        ref = dict(X=X, prior=pr, like=lk, logp=tr["logp"], moved=tr["moved"], try_idx=tr["try_idx"], cr_idx=tr["cr_idx"], snooker=tr["snooker"], Z=o.get_history())
        ref["cr_probs"], ref["cr_delta"], ref["cr_n"] = o.get_cr_state()
    else:
        e = SR.build(config, 0, 1, G, lag, M=M)
        e.step(G)
        ref = SR.results(e, G, with_history=True)
        e.close()
    assert not np.allclose(ref["cr_probs"], 1 / 3.) and ref["cr_n"].sum() > N
    nl = N // world
    sums0 = None
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npz" % r))
        sl = slice(r * nl, (r + 1) * nl)
        for key in ("X", "prior", "like"):
            np.testing.assert_array_equal(got[key], ref[key][sl], err_msg="%s of rank %d" % (key, r))
        for key in ("logp", "moved", "try_idx", "cr_idx", "snooker"):
            np.testing.assert_array_equal(got[key], ref[key][:, sl], err_msg="%s of rank %d" % (key, r))
        for key in ("cr_probs", "cr_delta", "cr_n"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg="%s of rank %d" % (key, r))
        if r == 0:
            np.testing.assert_array_equal(got["Z"], ref["Z"])
            sums0 = (int(got["checksum"][0]), int(got["rows"][0]))
        assert (int(got["checksum"][0]), int(got["rows"][0])) == sums0, "archive replica of rank %d" % r
        launches = int(got["launches"][0])
        if kind == "single":
            assert launches >= SR.BURNIN + 1, launches                 # one launch per burn-in generation (0 .. 24) and the thin-cycles behind
        else:
            assert launches <= (SR.BURNIN + 1) // 2 + 6, (launches, kind)      # several burn-in generations per launch
@pytest.mark.gpu
def test_history_checksum_is_the_documented_sum_and_sees_a_single_changed_element():
    from pydream_amd import _capi
    from tests import helpers as H
    d, N = 37, 24
    Z0 = H.seed_history(200, d, 9)
    sums = []
    for flip in (False, True):
        Z = Z0.copy()
        if flip:
            Z[150, 36] = np.nextafter(Z[150, 36], 1.0)
        e = _capi.Engine(nchains=N, ndim=d, multitry=3, history_capacity=400, seed=1)
        e.set_history(Z)
        h, rows = e.history_checksum()
        assert rows == 200 and h == _capi.history_checksum_host(Z)
        sums.append(h)
        e.close()
    assert sums[0] != sums[1]
    Zs = Z0.copy(); Zs[[3, 4]] = Zs[[4, 3]]                   # the same rows in other places: another sum
    assert _capi.history_checksum_host(Zs) != sums[0]


_RCCL_SINGLE_RANK = r"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(380, exit=True)       # a bootstrap that never comes up: say where, then leave
import numpy as np
sys.path.insert(0, sys.argv[1])
from pydream_amd import _capi
from tests import helpers as H
# No torch in this process: the engine's HIP runtime is the system ROCm's, and librccl is opened next to it (dz_comm_library).
lib, hip = _capi.comm_library(), _capi.hip_library()
assert os.path.dirname(os.path.realpath(lib)) == os.path.dirname(os.path.realpath(hip)) and "torch" not in lib, (lib, hip)
d, N, n = 16, 8, 25
P = H.mvn_precision(d); Z0 = H.seed_history(40, d, 4)
res = []
for use_comm in (False, True):
    e = _capi.Engine(nchains=N, ndim=d, multitry=5, history_capacity=40 + N * 8, trace_capacity=n, seed=5, history_thin=5)
    if use_comm:
        e.comm_init_rccl(0, 1, _capi.comm_unique_id())
        e.comm_barrier()
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), P, 0, 0.0)
    e.step(n)
    res.append((e.get_trace(0, n)["X"], e.get_history(), e.history_checksum()))
np.testing.assert_array_equal(res[0][0], res[1][0])
np.testing.assert_array_equal(res[0][1], res[1][1])
assert res[0][2] == res[1][2]
print("rccl single rank: equal")
"""


# The cold load of librccl.so (570 MB, paged in from cold storage fault by fault) took 200 s on a freshly started box in round 2;
# tests/conftest.py reads the file sequentially in the background from the start of a GPU session, so that by the time these tests run
# (last in the suite) it sits in the page cache.  The torch-first load order costs minutes more and stays opt-in.
needs_rccl_optin = pytest.mark.skipif(os.environ.get("DZ_TEST_RCCL", "0") != "1", reason="torch-first RCCL test runs with DZ_TEST_RCCL=1 (cold import of torch takes minutes)")


@pytest.mark.gpu
def test_rccl_single_rank_comm(tmp_path):
    """The transport north_star names, in the DEFAULT suite: RCCL bootstrap (ncclGetUniqueId, ncclCommInitRank) + the in-place
    ncclAllGather of the history appends + the one-element rendezvous all-gather, with world size 1 -- all a 1-GPU box can run -- in a
    process without torch, against the same run without a communicator: equal bit for bit.  In a process of its own under a hard
    400 s cap: a bootstrap that does not come up costs this test, not the suite."""
    import subprocess
    try:
        res = subprocess.run([sys.executable, "-c", _RCCL_SINGLE_RANK, ROOT], capture_output=True, text=True, timeout=400, cwd=str(tmp_path),
                             env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    except subprocess.TimeoutExpired as exc:
        raise AssertionError("the RCCL single-rank run did not finish within 400 s: %r" % (exc.stderr,))
    assert res.returncode == 0 and "rccl single rank: equal" in res.stdout, res.stderr[-3000:]


def _launch_bench(nranks, extra_args, env, cwd, timeout=600):
    """bench.py for N ranks the way torch.distributed.run starts it -- one process per rank with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in its environment -- without that launcher (it imports torch: minutes on a fresh box).  With
    DZ_TEST_TORCHRUN=1 the real launcher is used.  Returns rank 0's JSON line."""
    import json
    import subprocess
    port = str(_free_port())
    args = [os.path.join(ROOT, "bench.py"), "--gpus", str(nranks)] + extra_args
    if os.environ.get("DZ_TEST_TORCHRUN", "0") == "1":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
               "--master-port", port] + args
        res = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
        assert res.returncode == 0, res.stderr[-3000:]
        out = res.stdout
    else:
        procs = []
        for r in range(nranks):
            renv = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
            procs.append(subprocess.Popen([sys.executable] + args, cwd=cwd, env=renv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = []
        deadline = time.time() + timeout
        for pr in procs:
            try:
                outs.append(pr.communicate(timeout=max(1.0, deadline - time.time())))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise AssertionError("bench.py ranks did not finish in time")
        for pr, (so, se) in zip(procs, outs):
            assert pr.returncode == 0, se[-3000:]
        out = outs[0][0]
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["host", "peer"])
def test_bench_eight_rank_control_flow_on_one_gpu(tmp_path, transport):
    """bench.py as the driver launches it for N = 8 (one process per rank, torch.distributed.run's environment), rehearsed on the one
    GPU of the test box: the eight ranks share device 0 and exchange through the host (DZ_BENCH_DEVICE / DZ_BENCH_TRANSPORT).
    Proves the control flow the 8-GPU run takes: rendezvous, sharded engines, the convergence run with the
    sharded R-hat, timed blocks with the rank-maximum, one JSON line from rank 0 with whole-job throughput."""
    env = dict(os.environ, DZ_BENCH_DEVICE="0", DZ_BENCH_TRANSPORT=transport, HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="120")
    d = _launch_bench(8, ["--steps", "20", "--warmup", "5", "--chains-per-gpu", "128", "--rhat-max-generations", "400",
                          "--rhat-min-generations", "200", "--rhat-chunk", "100", "--rhat-window", "200", "--min-timed-ms", "20",
                          "--no-cpu-baseline"], env, str(tmp_path))
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["scaling"] == "weak"
    assert d["config"]["chains_global"] == 8 * 128 and transport in d["config"]["parallelism"]
    assert d["history_lag"] == 3 and d["replicas_identical"] is True and len(d["replica_check"]["archive_rows"]) == 8
    assert d["roofline"]["generations_per_launch"] == 20      # (round 6) two history appends per launch on several GPUs as on one
    if transport == "host":
        assert d["transport"] == "host-fallback"          # (a host-staged number is named as such at the top level)
    else:                                                  # eight ranks, each mapping the seven others' archives: seven copy streams per rank
        assert d["transport"] == "peer" and d["exchange"]["gates"] > 0, (d["transport"], d.get("transport_note"))
    assert d["value"] > 0 and abs(d["value"] - 8 * 128 * 5 * 20 / (d["timing"]["block_ms_median"] * 1e-3)) < 1e-6 * d["value"]
    assert d["convergence"]["generations_run"] >= 200 and np.isfinite(d["rhat_max"])
    assert d["kernel_times"]["exchange"]["launches"] > 0            # the Z appends were all-gathered


@pytest.mark.gpu
def test_bench_line_has_the_contracted_fields(tmp_path):
    """`python bench.py --steps K --warmup W` on one GPU (a small population so that it takes seconds): ONE JSON line with the driver's
    contract fields, `roofline` (achieved = algorithmic bytes / launch duration, frac = achieved / peak) and `cpu_baseline` (the oracle
    on a bounded sample), the kernel instantiation that ran, and a launch duration the timed blocks allow."""
    import json
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--chains-per-gpu", "512",
                          "--rhat-max-generations", "600", "--rhat-min-generations", "300", "--rhat-chunk", "100", "--rhat-window", "200",
                          "--min-timed-ms", "5", "--cpu-chains", "128", "--cpu-seconds", "1", "--rccl-leg"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "kernel_variant", "rhat_max", "convergence"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "proposals/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    # one GPU runs the schedule the scaling runs use (history_lag 3, two appends per launch) and times the lockstep schedule (lag 0) beside it
    assert d["history_lag"] == 3 and d["config"]["history_lag"] == 3 and d["value_history_lag0"] > 0 and d["history_lag0"]["kernel_variant"] == d["kernel_variant"]
    assert "replicas_identical" not in d
    assert abs(d["value"] - 512 * 5 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["launch_us"] * (20 / r["generations_per_launch"]) <= d["timing"]["block_ms_median"] * 1e3 * (1 + 1e-9)      # a kernel cannot outlast the block around it
    assert r["launches_timed"] >= 20 and r["generations_per_launch"] == 20 and r["kernel_variant"] == d["kernel_variant"] and d["kernel_variant"] == "k_generations_w4<7,tri,xlds,4,4,lean,ahead>"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "proposals/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"]
    # --rccl-leg: the transport north_star names, through bench.py itself, with the one rank a one-GPU box can give it -- the same engine
    # re-attached to an RCCL communicator and the same blocks timed again (every N > 1 line carries these keys)
    assert d["rccl_ranks"] == 1 and d["rccl_value"] > 0 and d["rccl"]["timed_blocks"] >= 1 and d["rccl_exchange_exposed_us_per_cycle"] is not None
    assert "torch" not in d["rccl"]["library"] and d["rccl"]["kernel_variant"] == d["kernel_variant"]
    assert "configs" not in d                      # (512 chains per GPU is not the default workload: no configs block)


@pytest.mark.gpu
def test_bench_default_line_carries_every_single_gpu_baseline_config(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command; here with a short convergence run): the line's `configs` block
    holds BASELINE configs[1] (1024 chains x 100-D), configs[2] as written (4096-chain 3-Gaussian mixture with crossover adaptation:
    the rate inside the burn-in and after it) and the configs[4] per-GPU shard (512 chains x 1000-D), each with its rate, the kernel
    instantiation that ran, its roofline fraction and the run-so-far R-hat."""
    import json
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--rhat-max-generations", "600", "--rhat-min-generations", "300", "--rhat-chunk", "100", "--rhat-window", "200",
                          "--min-timed-ms", "5", "--no-cpu-baseline", "--no-dense", "--no-lag0"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert "configs[3] per-GPU shard" in d["config"]["workload"]
    c = d["configs"]
    assert set(c) == {"configs[1]", "configs[2]", "configs[3] @ 1 GPU", "configs[4] shard", "example d=200", "user device likelihood"}
    # (round 6) configs[3] as written is 32768 chains: its one-GPU point, the anchor of the strong-scaling reading (`strong_scaling` in the N > 1 lines)
    x = c["configs[3] @ 1 GPU"]
    assert "error" not in x, x
    assert "32768 chains" in x["workload"] and "strong-scaling" in x["workload"] and x["kernel_variant"] == "k_generations<7,tri,xlds,16,1,lean>"
    assert x["value"] > 0 and 0 < x["roofline"]["frac"] < 1 and np.isfinite(x["rhat_run_so_far"])
    assert c["example d=200"]["kernel_variant"] == "k_generations_d2<13,tri,xhbm,16,1,lean>" and c["example d=200"]["value"] > 0
    # a user's device function compiled at run time into the persistent kernel (hipcc is on the GPU box): at least the built-in mixture's rate
    assert c["user device likelihood"]["kernel_variant"] == "k_generations_user" and c["user device likelihood"]["value"] > 0.9 * c["configs[2]"]["value"]
    for key, variant, label in (("configs[1]", "k_generations_w4<7,tri,xlds,4,4,lean,ahead>", "BASELINE configs[1]"),
                                ("configs[2]", "k_generations_mix", "BASELINE configs[2] as written"),
                                ("configs[4] shard", "multi-kernel path", "BASELINE configs[4] per-GPU shard")):
        x = c[key]
        assert "error" not in x, x
        assert x["kernel_variant"] == variant and label in x["workload"] and x["value"] > 0 and x["ms_per_step"] > 0
        assert 0 < x["roofline"]["frac"] < 1 and np.isfinite(x["rhat_run_so_far"]) and x["rhat_generations"] >= 300
    assert c["configs[1]"]["roofline"]["bound"] == "hbm" and c["configs[4] shard"]["roofline"]["bound"] == "fp64_mfma"
    assert 0 < c["configs[4] shard"]["roofline"]["whole_generation_frac"] < 1
    assert c["configs[2]"]["burnin_value"] > 0 and c["configs[2]"]["value"] > c["configs[2]"]["burnin_value"]
    # (round 6) the burn-in runs with an adapt_lag -- named in the workload string --, whole launches of 20 generations inside it; the lockstep adaptation beside it
    b = c["configs[2]"]["burnin"]
    assert b["adapt_lag"] == 19 and "adapt_lag 19" in c["configs[2]"]["workload"] and b["kernel_variant"] == "k_generations_mix<multi>"
    assert 0 < c["configs[2]"]["burnin_value_adapt_lag0"] < c["configs[2]"]["burnin_value"] and b["adapt_lag0"]["kernel_variant"] == "k_generations_mix"
    assert not np.allclose(c["configs[2]"]["burnin"]["cr_probs_after_burnin"], 1 / 3.)


@pytest.mark.gpu
def test_bench_two_ranks_with_crossover_adaptation_exchange_group_sums(tmp_path):
    """`bench.py --gpus 2 --adapt` (two ranks sharing device 0, 512 chains each = two whole groups per rank): `burnin_value` is reported,
    the ranks' replicas and adapted probabilities agree, and what travels per burn-in generation is the groups' sums -- under 100 KB per
    rank, those of a whole launch of generations in one exchange (bench.py's adapt_lag) -- while the positions travel once."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DZ_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="120")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--chains-per-gpu", "512", "--adapt",
                          "--burnin-generations", "160", "--rhat-max-generations", "400", "--rhat-min-generations", "200", "--rhat-chunk", "100", "--rhat-window", "200",
                          "--min-timed-ms", "20", "--no-cpu-baseline", "--transport", "peer", "--no-rccl-leg"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d["n_gpus"] == 2 and d["transport"] == "peer" and d["replicas_identical"] is True
    # ranks of whole groups of 256 chains: the same adapt_lag as on one GPU (twenty burn-in generations per launch, their group sums in one exchange)
    assert d["burnin_value"] > 0 and d["burnin"]["kernel_variant"] == "k_generations_w4<7,tri,xlds,4,4,lean,ahead> +ring" and d["config"]["adapt_lag"] == 19
    assert not np.allclose(d["burnin"]["cr_probs_after_burnin"], 1 / 3.)
    xb = d["exchange_bytes_to_each_peer"]
    assert xb["positions"] == 512 * 112 * 8 and 0 < xb["per_burnin_generation"] <= 100 * 1024
    assert xb["adaptation_group_sums"] == 161 * (2 * (6 * 112 + 16) + 112) * 8


@pytest.mark.gpu
def test_bench_with_crossover_adaptation_reports_the_burnin_rate(tmp_path):
    """BASELINE configs[2] as written (crossover adaptation ON): `bench.py --target mix3 --adapt` times blocks inside the burn-in
    (`burnin_value`, persistent kernel with one generation per launch + the adaptation launches) and after it (`value`)."""
    import json
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--target", "mix3", "--adapt",
                          "--burnin-generations", "160", "--chains-per-gpu", "512", "--rhat-max-generations", "400", "--rhat-min-generations", "200",
                          "--rhat-chunk", "100", "--rhat-window", "200", "--min-timed-ms", "5", "--no-cpu-baseline"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d["burnin"]["kernel_variant"] == "k_generations_mix<multi>" and d["kernel_variant"] == "k_generations_mix" and d["config"]["adapt_lag"] == 19
    assert d["burnin_value"] > 0 and d["value"] > d["burnin_value"] > d["burnin_value_adapt_lag0"] > 0
    assert not np.allclose(d["burnin"]["cr_probs_after_burnin"], 1 / 3.)          # the probabilities were adapted


@pytest.mark.gpu
def test_bench_gpus_2_as_one_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 --steps 20 --warmup 5` as ONE process without a launcher's environment (how the driver starts
    `--gpus 1`): bench.py starts its own two ranks (here both on device 0: DZ_BENCH_DEVICE) and prints "n_gpus": 2 -- it can no longer
    silently run one GPU.  Rows exchanged by the PEER transport (IPC-mapped archives, copy-stream pushes, gate kernels) with
    history_lag = 1; the line names the transport, reports how long the gates waited per thin-cycle, and carries the replica check:
    both ranks' archives reduce to the same checksum.  Without DZ_BENCH_DEVICE on a box with fewer devices than ranks it refuses."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DZ_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="120")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--chains-per-gpu", "512",
                          "--rhat-max-generations", "400", "--rhat-min-generations", "200", "--rhat-chunk", "100", "--rhat-window", "200",
                          "--min-timed-ms", "20", "--no-cpu-baseline", "--transport", "peer"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["transport"] == "peer" and d["history_lag"] == 3, (d.get("transport"), d.get("transport_note"))
    assert d["roofline"]["generations_per_launch"] == 20      # (round 6) the N = 1 line's launches: two history appends each
    assert d["replicas_identical"] is True and len(set(d["replica_check"]["archive_checksums"])) == 1 and len(d["replica_check"]["archive_rows"]) == 2
    assert d["exchange"]["gates"] > 0 and d["exchange_exposed_us_per_cycle"] is not None and d["exchange_exposed_us_per_cycle"] >= 0.0
    assert d["kernel_variant"] == "k_generations_w4<7,tri,xlds,4,4,lean,ahead>"
    assert np.isfinite(d["rhat_max"]) and d["value"] > 0
    # the RCCL leg of every N > 1 line: the keys are there; on this box both ranks sit on ONE device, which RCCL refuses -- then the line says so
    for key in ("rccl_value", "rccl_ranks", "rccl_exchange_exposed_us_per_cycle"):
        assert key in d, key
    assert (d["rccl_value"] is None and "RCCL leg not run" in d["rccl"]["note"]) or (d["rccl_value"] > 0 and d["rccl_ranks"] == 2 and d["rccl"]["replicas_identical"])
    assert d["exchange_bytes_to_each_peer"]["history_rows"] > 0 and d["exchange_bytes_to_each_peer"]["positions"] == 0
    # (round 6) BASELINE configs[3] as written, the strong-scaling reading: 32768 chains over the run's GPUs
    ss = d["strong_scaling"]
    assert ss["chains_global"] == 32768 and ss["chains_per_gpu"] == 16384 and ss["value"] > 0 and ss["replicas_identical"] is True and ss["transport"] == "peer"
    assert ss["kernel_variant"] == "k_generations<7,tri,xlds,16,1,lean>"
    from pydream_amd import _capi
    if _capi.device_count() < 2:                           # the refusal: never fewer ranks than asked for
        env.pop("DZ_BENCH_DEVICE")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], cwd=str(tmp_path), env=env,
                             capture_output=True, text=True, timeout=120)
        assert res.returncode != 0 and '"metric"' not in res.stdout and "refusing" in res.stderr


@pytest.mark.gpu
@needs_rccl_optin
def test_rccl_next_to_the_engines_hip_runtime_in_bench_load_order():
    """bench.py's load order for N > 1 -- libdreamzs.so first (binds the system ROCm's HIP runtime), torch afterwards --, in a fresh
    process (tools/rccl_rocm_check.py): the engine opens the librccl NEXT TO THAT runtime, not the copy torch bundles, and an
    all-gather through it (world size 1) leaves the run unchanged."""
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_rocm_check.py")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("hip:")][0].split()
    hip, rccl = line[1], line[3]
    assert os.path.dirname(os.path.realpath(hip)) == os.path.dirname(os.path.realpath(rccl)) and "torch" not in rccl, line
    assert "equal: True" in res.stdout


def test_socket_group_wire_format_round_trips_and_names_no_code():
    """The control plane's values (None, ints beyond 64 bits, floats, bytes, text, arrays, nested lists / tuples) survive the tagged framing;
    anything else is refused at the sender, an unknown tag or a bad length at the receiver -- nothing on the wire can name code (the
    advisor's finding against pickle)."""
    from pydream_amd.distributed import _decode, _encode
    vals = [None, True, 7, -2 ** 70, 1.5, b"\x00\xff", "hé", [1, (2.0, None), b"x"], (np.arange(6.0).reshape(2, 3), np.ones(3, np.uint8)), np.zeros((0, 4))]
    for v in vals:
        buf = bytearray(); _encode(v, buf)
        out, pos = _decode(bytes(buf))
        assert pos == len(buf)
        if isinstance(v, np.ndarray):
            assert out.dtype == v.dtype and out.shape == v.shape and np.array_equal(out, v)
        elif isinstance(v, tuple) and isinstance(v[0], np.ndarray):
            assert all(np.array_equal(a, b) for a, b in zip(out, v)) and isinstance(out, tuple)
        else:
            assert out == v and type(out) is type(v)
    with pytest.raises(TypeError):
        _encode(object(), bytearray())
    with pytest.raises(TypeError):
        _encode(np.array(["a"]), bytearray())
    with pytest.raises(ValueError):
        _decode(b"Z")
    with pytest.raises(ValueError):
        _decode(b"B" + (10 ** 6).to_bytes(8, "little") + b"ab")
    import pickle
    with pytest.raises(ValueError):
        _decode(pickle.dumps([1, 2, 3]))


def _sg_rank(rank, world, port, token, q):
    sys.path.insert(0, ROOT)
    from pydream_amd.distributed import SocketGroup
    g = SocketGroup(rank, world, "127.0.0.1", port, timeout=30.0, token=token)
    q.put((rank, g.all_gather_object("r%d" % rank), g.all_reduce_max([float(rank)])))
    g.close()


def test_socket_group_seats_only_ranks_that_know_the_token():
    """Rank 0 listens on loopback; a stranger (wrong token), a rank number out of range and a second claimant of a seated rank are dropped, the
    collective then completes with the real ranks; a non-loopback address is refused outright."""
    import multiprocessing as mp
    import socket as S
    import struct
    from pydream_amd.distributed import SocketGroup, new_token
    with pytest.raises(ValueError):
        SocketGroup(0, 2, "0.0.0.0", 1, token=new_token())
    port, token = _free_port(), new_token()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    hub = ctx.Process(target=_sg_rank, args=(0, 3, port, token, q)); hub.start()
    time.sleep(0.5)

    def knock(rank, tok):
        c = S.create_connection(("127.0.0.1", port), timeout=5.0)
        c.sendall(b"DZRDV1" + struct.pack("<i", rank) + tok)
        c.settimeout(3.0)
        try:
            return c.recv(1)
        except OSError:
            return b""
        finally:
            c.close()
    assert knock(1, bytes(32)) == b""                               # wrong token
    assert knock(7, bytes.fromhex(token)) == b""                    # rank out of range
    others = [ctx.Process(target=_sg_rank, args=(r, 3, port, token, q)) for r in (1, 2)]
    for pr in others:
        pr.start()
    res = sorted(q.get(timeout=60) for _ in range(3))
    for pr in [hub] + others:
        pr.join(30); assert pr.exitcode == 0
    assert [r[1] for r in res] == [["r0", "r1", "r2"]] * 3 and [r[2] for r in res] == [[2.0]] * 3


_GATE_TIMEOUT_RANK = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from pydream_amd import _capi
from pydream_amd.distributed import attach_transport, socket_group_from_env
from tests import helpers as H
rank, world = int(os.environ["RANK"]), 2
group = socket_group_from_env(timeout=60.0)
d, N = 16, 16
Z0 = H.seed_history(64, d, 4)
e = _capi.Engine(nchains=N, nchains_local=N // 2, chain_offset=rank * (N // 2), ndim=d, multitry=5, history_capacity=64 + N * 8, trace_capacity=40, seed=5, history_thin=5)
e.set_history(Z0); e.set_state(Z0[rank * (N // 2):(rank + 1) * (N // 2)]); e.set_likelihood_mvn(np.zeros(d), H.mvn_precision(d), 0, 0.0)
attach_transport(e, rank, world, transport="peer", group=group)
if rank == 1:                       # this rank never steps: its rows never arrive
    time.sleep(8.0)
    os._exit(0)
t0 = time.time()
try:
    for _ in range(40):
        e.step(1); e.sync()
    print("NO ERROR", flush=True)
except _capi.DreamZSError as exc:
    print("stopped after %.1f s: %s" % (time.time() - t0, exc), flush=True)
os._exit(0)
"""


@pytest.mark.gpu
def test_a_peer_gate_that_times_out_stops_the_run(tmp_path):
    """A rank whose peer never delivers its rows does not sample an archive with holes in it: the gate kernel gives up after
    DZ_PEER_TIMEOUT_S, and from then on dz_step queues nothing and every synchronising call fails, naming the silent rank (advisor,
    round 3: the time-out used to be recorded and found at the end of the run, if at all)."""
    import subprocess
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DZ_PEER_TIMEOUT_S="2", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    procs = [subprocess.Popen([sys.executable, "-c", _GATE_TIMEOUT_RANK, ROOT], cwd=str(tmp_path), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [pr.communicate(timeout=120) for pr in procs]
    assert "stopped after" in outs[0][0] and "rank 1" in outs[0][0] and "NO ERROR" not in outs[0][0], outs[0]
