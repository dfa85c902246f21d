"""Parity tests proper: the HIP engine (through the C ABI) against (a) the vectors produced by the
REFERENCE (tests/golden) and (b) the CPU oracle on the same seeded inputs -- bit-exact."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

S2_TRACES = ["trace_s2_adapt", "trace_s2_k1_bounds", "trace_s2_k3_bounds", "trace_s2_k3_redraw", "trace_s2_k5_redraw_mvn", "trace_s2_depairs_gamma",
             "trace_s2_mvn100", "trace_s2_mix3", "trace_s2_restart", "trace_s2_lag1", "trace_s2_lag2_k1", "trace_s2_lag3",
             "trace_s2_adaptlag1", "trace_s2_adaptlag9_mix", "trace_s2_adaptlag3_gamma"]


@pytest.fixture(scope="module")
def G():
    from pydream_amd import _capi
    return _capi


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def assert_traces_identical(a, b):
    for key in ("snooker", "cr_idx", "try_idx", "moved", "X", "logp"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)


@pytest.mark.parametrize("name", S2_TRACES)
def test_golden_traces(name, G, O):
    """HIP == reference (decision sequences exact, logp 1e-10) and HIP == oracle bit-for-bit."""
    fx = H.load(name)
    n = int(fx["cfg_G"])
    e = H.engine_from_trace_fixture(G.Engine, fx)
    e.step(n)
    tr = e.get_trace(0, n)
    gp = e.get_gamma_state()[0] if int(fx["cfg_adapt_gamma"]) else None
    H.compare_with_reference(tr, fx, e.get_history(), e.get_cr_state()[0], gp)
    assert (e.redraw_rounds() > 0) == ("redraw" in name)      # Dream.py:281-289 taken exactly where the reference took it
    o = H.engine_from_trace_fixture(O.Engine, fx)
    o.step(n)
    assert_traces_identical(tr, o.get_trace(0, n))
    np.testing.assert_array_equal(e.get_history(), o.get_history())
    for a, b in zip(e.get_cr_state(), o.get_cr_state()):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(e.get_gamma_state(), o.get_gamma_state()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(e.get_rhat(), o.get_rhat())
    for a, b in zip(e.get_state(), o.get_state()):
        np.testing.assert_array_equal(a, b)


MVN_S2_TRACES = [n for n in S2_TRACES if str(H.load(n)["lk_kind"]) == "mvn"]


@pytest.mark.parametrize("name", MVN_S2_TRACES)
def test_golden_traces_with_the_triangular_factor(name, G, O):
    """The likelihood form the driver's headline times (the triangular factor of the precision matrix, kind 1) on the HIP engine against
    the decision sequences the REFERENCE made with its dense formula (dream_ex_ndim_gaussian.py:49-52; Dream.py:908, :993): snooker flags,
    CR indices, selected tries and accept flags exact, log p to 1e-10 -- and bit for bit against the oracle given the same factor."""
    fx = H.load(name)
    n = int(fx["cfg_G"])
    e = H.engine_from_trace_fixture(G.Engine, fx, mvn_kind="tri")
    e.step(n)
    tr = e.get_trace(0, n)
    gp = e.get_gamma_state()[0] if int(fx["cfg_adapt_gamma"]) else None
    H.compare_with_reference(tr, fx, e.get_history(), e.get_cr_state()[0], gp)
    o = H.engine_from_trace_fixture(O.Engine, fx, mvn_kind="tri")
    o.step(n)
    assert_traces_identical(tr, o.get_trace(0, n))
    np.testing.assert_array_equal(e.get_history(), o.get_history())
    for a, b in zip(e.get_cr_state(), o.get_cr_state()):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("tag", ["d100k5", "d4k5b", "d4k1b", "d10k1"])
def test_generate_proposal_points(tag, G, O):
    fx = H.load("proposals")
    g = lambda a: fx[tag + "__" + a]
    d, k = int(g("d")), int(g("k"))
    Z, q0 = g("Z"), g("q0")
    kw = dict(nchains=4, ndim=d, multitry=k, depairs=int(g("depairs")), ngamma=int(g("ngamma")),
              history_capacity=len(Z) + 8, seed=int(g("seed")), lamb=float(g("lamb")))
    e, o = G.Engine(**kw), O.Engine(**kw)
    for x in (e, o):
        x.set_history(Z)
        if int(g("bounded")):
            x.set_bounds(g("mins"), g("maxs"))
    pos = 0
    for (trial, phase, snk, cr_idx, delta, glev) in g("meta"):
        n = k if phase == 0 else k - 1
        args = (3, int(trial), int(phase), q0, int(snk), int(cr_idx), int(delta), int(glev))
        pts, slogp = e.debug_propose(*args)[:2]
        opts, oslogp = o.debug_propose(*args)[:2]
        np.testing.assert_array_equal(pts, opts)
        np.testing.assert_array_equal(slogp, oslogp)
        ref = g("pts")[pos:pos + n * d].reshape(n, d)
        pos += n * d
        if snk:
            np.testing.assert_allclose(pts, ref, rtol=1e-12, atol=1e-14)
        else:
            np.testing.assert_array_equal(pts, ref)       # DE proposals: bit-exact against the reference


def test_logpdf_kernels(G, O):
    """MVN d=10/100/200 (dense precision and triangular factor) and mixtures: 1e-10 vs the reference,
    bit-exact vs the oracle."""
    fx = H.load("densities")
    for d in (10, 100, 200):
        P, X, ref = fx["mvn%d_invC" % d], fx["mvn%d_X" % d], fx["mvn%d_logp" % d]
        logF = float(fx["mvn%d_logF" % d])
        e, o = G.Engine(nchains=3, ndim=d, history_capacity=8), O.Engine(nchains=3, ndim=d, history_capacity=8)
        U = np.linalg.cholesky((P + P.T) / 2).T
        for M, kind in ((P, 0), (U, 1)):
            e.set_likelihood_mvn(np.zeros(d), M, kind, logF)
            o.set_likelihood_mvn(np.zeros(d), M, kind, logF)
            got = e.eval_logp(X)[1]
            np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-10)
            np.testing.assert_array_equal(got, np.array([o.loglike(x) for x in X]))
    for tag, d in (("mix2", 10), ("mix3", 100)):
        e, o = G.Engine(nchains=3, ndim=d, history_capacity=8), O.Engine(nchains=3, ndim=d, history_capacity=8)
        e.set_likelihood_mixture(fx[tag + "_mu"], fx[tag + "_logF"])
        o.set_likelihood_mixture(fx[tag + "_mu"], fx[tag + "_logF"])
        got = e.eval_logp(fx[tag + "_X"])[1]
        np.testing.assert_allclose(got, fx[tag + "_logp"], rtol=0, atol=1e-10)
        np.testing.assert_array_equal(got, np.array([o.loglike(x) for x in fx[tag + "_X"]]))


def _c2_like(Cls, N, d, G_, seed, k=5, **kw):
    P = H.mvn_precision(d)
    Z0 = H.seed_history(max(10 * d, 2 * N), d, seed)
    e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (G_ // 10 + 2), trace_capacity=G_, seed=seed, **kw)
    e.set_history(Z0)
    e.set_state(Z0[:N])
    e.set_likelihood_mvn(np.zeros(d), P, 0, 0.0)
    return e


def test_c2_config_against_oracle(G, O):
    """BASELINE configs[1] (1024 chains, 100-D MVN, multitry 5, DE+snooker): 30 generations, everything bit-exact."""
    n = 30
    e, o = _c2_like(G.Engine, 1024, 100, n, 20260929), _c2_like(O.Engine, 1024, 100, n, 20260929)
    e.step(n); o.step(n)
    assert_traces_identical(e.get_trace(0, n), o.get_trace(0, n))
    np.testing.assert_array_equal(e.get_history(), o.get_history())
    acc = e.get_trace(0, n)["moved"].mean()
    assert 0.1 < acc < 0.6


def test_c3_config_against_oracle(G, O):
    """BASELINE configs[2] shape (mixture of 3 Gaussians, CR adaptation on) at 512 chains x 100-D."""
    N, d, n, seed = 512, 100, 40, 5
    mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
    logF = np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi)
    Z0 = H.seed_history(2 * N, d, seed, lo=-8, hi=8)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1, crossover_burnin=25)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mixture(mu, logF)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_cr_state(), e.get_history()))
    assert_traces_identical(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert not np.allclose(out[0][1][0], 1 / 3.)       # crossover probabilities did adapt


def test_d1000_stress_shape_against_oracle(G, O):
    """BASELINE configs[4] shape (1000-D correlated MVN, triangular-factor logpdf) at 16 chains."""
    N, d, n, seed = 16, 1000, 6, 9
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = H.seed_history(64, d, seed)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * 3, trace_capacity=n, seed=seed)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(n)
        out.append(e.get_trace(0, n))
    assert_traces_identical(out[0], out[1])


@pytest.mark.parametrize("N,d,k,tri,burnin,eligible,prior", [(1024, 100, 5, 0, 0, True, None), (1000, 100, 5, 1, 0, True, None), (96, 10, 3, 1, 0, True, None),
                                                             (64, 64, 4, 0, 0, True, None), (256, 100, 5, 1, 12, True, None),
                                                             (64, 128, 4, 0, 0, True, None), (64, 128, 4, 1, 0, True, None),
                                                             (4096, 20, 5, 1, 0, True, "uniform"), (200, 100, 5, 0, 0, True, "uniform"),
                                                             (96, 10, 3, 1, 0, True, "normal"), (64, 128, 4, 0, 0, True, "normal"),
                                                             (4096, 100, 1, 1, 0, True, None), (1000, 100, 1, 0, 0, True, None), (96, 10, 1, 1, 0, True, "uniform"),
                                                             (256, 100, 1, 1, 12, True, None), (64, 128, 1, 0, 0, True, "normal"), (4096, 20, 1, 1, 0, True, "uniform"),
                                                             # 128 < d <= 256 (round 5): k_generations_d2 -- the matrix from L2, two 128-dimension chunks per lane
                                                             (300, 160, 5, 1, 0, True, None), (1100, 200, 5, 1, 0, True, None), (100, 200, 3, 0, 0, False, None),
                                                             (250, 224, 5, 1, 12, True, None), (64, 140, 4, 1, 12, True, None), (48, 129, 15, 1, 0, True, None), (48, 129, 8, 1, 0, True, None),
                                                             (250, 256, 5, 1, 0, True, None), (1100, 240, 5, 1, 0, True, None), (200, 256, 5, 1, 12, True, None), (100, 250, 4, 1, 0, True, "normal"),      # (round 6: 8 chains x 2 waves per block where 16 chains' point tiles do not fit)
                                                             (1100, 256, 8, 1, 0, True, None), (200, 250, 9, 1, 12, True, "normal"), (120, 256, 12, 1, 0, True, None),      # (8 x 2 with the two-pass set; 4 x 4)
                                                             (3072, 100, 5, 1, 0, True, None), (2500, 100, 5, 1, 12, True, "uniform"), (2300, 64, 3, 0, 0, True, None),      # (round 6: 12 chains per block)
                                                             (64, 200, 1, 1, 0, True, None), (300, 160, 1, 1, 12, True, None), (64, 200, 5, 1, 0, True, "normal"), (100, 160, 4, 1, 12, True, "uniform"), (64, 200, 1, 1, 0, True, "uniform"), (64, 200, 1, 0, 0, False, None),
                                                             (80, 128, 5, 1, 12, True, None), (80, 127, 6, 1, 0, True, "normal"), (48, 100, 12, 1, 0, True, None),      # (d <= 128 where 16 chains do not fit next to the matrix: k_generations_d2<8 / 7,..>)
                                                             (32, 300, 5, 1, 0, False, None)])
def test_persistent_kernel_equals_multi_kernel_path_and_oracle(G, O, N, d, k, tri, burnin, eligible, prior, monkeypatch):
    """k_generations (whole thin-cycles in one launch, the default wherever it is eligible) against the
    multi-kernel path (DZ_MEGA=0) and the oracle: 35 generations across three history appends, chain counts that do
    not fill the last block, dense and triangular matrix, and a crossover burn-in in front (multi-kernel during the
    burn-in, persistent afterwards).  Chain counts below 4096 run 8 or 4 chains per block (the 128-D cases only fit that
    way); the multi-kernel run of the dense 128-D case also exercises the likelihood kernel that takes its operands from
    L2 (its matrix does not fit LDS).  prior: SampledParam-style priors -- "uniform" with hard boundaries narrow enough that
    proposals are reflected and some are redrawn (Dream.py:733-791), "normal" without boundaries (with priors the persistent
    kernel keeps the chain states in LDS; at 4 chains per block even the dense 128-D matrix leaves room for that).
    multitry = 1 (the reference's default, Dream.py:271-275 and :326-334): one proposal per generation, no reference set, the
    snooker move's current-point term."""
    n, seed = 35, 77
    P = H.mvn_precision(d)
    M = np.linalg.cholesky((P + P.T) / 2).T if tri else P
    Z0 = H.seed_history(max(10 * d, 2 * N), d, seed)

    def run(Cls, mega):
        monkeypatch.setenv("DZ_MEGA", "1" if mega else "0")
        monkeypatch.setenv("DZ_MEGA_D2", "2" if N < 1100 else "1")           # (k_generations_d2 is the default from 1025 chains on; the small cases force it)
        e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1 if burnin else 0, crossover_burnin=burnin)
        mu = np.linspace(-1.0, 1.0, d) if d in (10, 64, 140) else np.zeros(d)      # (a zero mean takes a shorter code path)
        if prior == "uniform":                                                # scipy uniform(loc=-6, scale=22): support [-6, 16]
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -6.0), np.full(d, 22.0))
            e.set_bounds(np.full(d, -6.0), np.full(d, 16.0))
        elif prior == "normal":
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 2.0, d), np.full(d, 30.0))
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(mu, M, tri, 0.0)
        launches = None
        if Cls is G.Engine:
            e.profile_enable(True); e.profile_reset()
        e.step(n)
        if Cls is G.Engine:
            launches = e.profile_get("generations")[1]
            e.profile_enable(False)
        return e.get_trace(0, n), e.get_history(), e.get_cr_state()[0], launches

    a, b, o = run(G.Engine, True), run(G.Engine, False), run(O.Engine, False)
    assert (a[3] > 0) == eligible and b[3] == 0              # the persistent kernel really ran / really did not
    for other in (b, o):
        assert_traces_identical(a[0], other[0])
        np.testing.assert_array_equal(a[1], other[1])
        np.testing.assert_array_equal(a[2], other[2])


@pytest.mark.parametrize("N,tri,target,lag,variant", [(4096, 1, "mvn", 3, "k_generations<7,tri,xlds,16,1,lean>"),    # bench.py's headline `value` (history_lag = 3, two appends per launch at every N: round 6)
                                                      (4096, 1, "mvn", 0, "k_generations<7,tri,xlds,16,1,lean>"),    # ... `value_history_lag0`
                                                      (4096, 1, "mvn", 1, "k_generations<7,tri,xlds,16,1,lean>"),    # (rounds 3-5's headline schedule)
                                                      (4096, 0, "mvn", 3, "k_generations<7,dense,xhbm,16,1,lean>"),  # ... `dense_value`
                                                      (4096, 0, "mvn", 0, "k_generations<7,dense,xhbm,16,1,lean>"),
                                                      (1024, 1, "mvn", 3, "k_generations_w4<7,tri,xlds,4,4,lean,ahead>"),     # BASELINE configs[1] (the `configs` block of the line)
                                                      (1024, 1, "mvn", 0, "k_generations_w4<7,tri,xlds,4,4,lean,ahead>"),
                                                      (1024, 0, "mvn", 3, "k_generations_w4<7,dense,xlds,4,4,lean,ahead>"),
                                                      (2048, 1, "mvn", 3, "k_generations<7,tri,xlds,8,1,lean>"),
                                                      (3072, 1, "mvn", 3, "k_generations<7,tri,xlds,12,1,lean>"),    # (round 6: 12 chains per block)
                                                      (4096, 1, "mix3", 3, "k_generations_mix"),                     # BASELINE configs[2] after the burn-in
                                                      (4096, 1, "mix3", 0, "k_generations_mix")])
def test_the_instantiations_the_bench_times_equal_the_oracle(G, O, N, tri, target, lag, variant, monkeypatch):
    """Exactly what bench.py times -- 100-D, multitry 5, flat prior, bench.py's own engine set-up INCLUDING its history_lag (3 for the
    driver's `value`, with bench.py's two appends per launch; 0 for `value_history_lag0`) -- against the oracle, bit for bit, over 65
    generations (seven history appends, so that with lag 3 the rows of three of them have been sampled), with the engine reporting which
    instantiation ran (dz_last_kernel_variant): 16 chains per block with one wave per chain needs >= ~2100 chains, which no other oracle
    comparison reaches."""
    monkeypatch.setenv("DZ_MEGA_SEGS", "2")
    import argparse
    import bench
    n = 65 if lag == 3 else 45
    args = argparse.Namespace(dim=100, multitry=5, seed=20260929, thin=10, snooker=0.1, target=target, mvn_kind="tri" if tri else "dense",
                              steps=n, warmup=0, history_lag=lag)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = bench.setup_engine(Cls, args, N, N, 0, n, trace_capacity=n, **({"schedule": 2} if Cls is O.Engine else {}))
        assert int(e.cfg.history_lag) == lag
        e.step(n)
        if Cls is G.Engine:
            assert e.last_kernel_variant() == variant
        out.append((e.get_trace(0, n), e.get_history()))
    assert_traces_identical(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert 0.05 < out[0][0]["moved"].mean() < 0.9


@pytest.mark.parametrize("N,d,k,target,lag,thin", [(4096, 100, 5, "mvn", 1, 10), (1024, 100, 5, "mvn", 1, 10), (1024, 100, 5, "mvn", 3, 2), (2048, 48, 3, "mvn", 2, 5),
                                                   (1040, 200, 5, "mvn", 1, 10), (1040, 160, 1, "mvn", 2, 3), (4096, 100, 5, "mix3", 1, 10), (512, 10, 1, "mix3", 3, 1)])
def test_launches_that_hold_several_history_appends(G, O, N, d, k, target, lag, thin, monkeypatch):
    """With history_lag L >= 1 on one GPU the rows an append writes are not sampled before L more appends, so a launch of the persistent
    kernels runs on past its appends (up to L + 1 of them, dz_engine.hip mega_segment): every generation must still sample exactly the rows
    the lag schedule shows it (record_history Dream.py:919-938, sample_from_history :646-668).  All four kernels (k_generations,
    k_generations_w4, k_generations_d2, k_generations_mix), stepped in uneven pieces so that launches start and end inside a thinning cycle,
    against the same engine with one append per launch (DZ_MEGA_SEGS=1) and against the oracle, bit for bit -- and with fewer launches."""
    import argparse
    import bench
    pieces = (13, 2 * thin * (lag + 1) + 1, 3 * thin + 2, 7)
    n = sum(pieces)
    args = argparse.Namespace(dim=d, multitry=k, seed=424242, thin=thin, snooker=0.1, target=target, mvn_kind="tri", steps=n, warmup=0, history_lag=lag)

    def run(Cls, segs):
        if segs is None:
            monkeypatch.delenv("DZ_MEGA_SEGS", raising=False)
        else:
            monkeypatch.setenv("DZ_MEGA_SEGS", str(segs))
        e = bench.setup_engine(Cls, args, N, N, 0, n, trace_capacity=n, **({"schedule": 2} if Cls is O.Engine else {}))
        assert int(e.cfg.history_lag) == lag
        launches = None
        if Cls is G.Engine:
            e.profile_enable(True); e.profile_reset()
        for m in pieces:
            e.step(m)
        if Cls is G.Engine:
            launches = e.profile_get("generations")[1]
            e.profile_enable(False)
            assert e.last_kernel_variant().startswith("k_generations")
        return e.get_trace(0, n), e.get_history(), launches

    a, b, o = run(G.Engine, None), run(G.Engine, 1), run(O.Engine, None)
    assert 0 < a[2] < b[2], (a[2], b[2])                                    # the same generations in fewer launches
    for other in (b, o):
        assert_traces_identical(a[0], other[0])
        np.testing.assert_array_equal(a[1], other[1])


def test_an_archive_sized_to_the_last_append_is_enough_for_launches_with_several_appends(G, O):
    """history_capacity = seed rows + exactly the appends the run makes (core.py:260-268 sizes the reference's archive the same way): a launch
    that would make two appends must not ask for rows a one-append launch would not have needed -- the run goes through, equals the oracle,
    and the append after the last one that fits fails loudly instead of writing past the archive."""
    N, d, k, thin, lag, n = 256, 16, 3, 5, 1, 40                            # appends after generations 0, 5, .., 40 would be nine; the archive holds eight
    Z0 = H.seed_history(2 * N, d, 5)
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=k, history_thin=thin, history_lag=lag, history_capacity=len(Z0) + N * (n // thin), trace_capacity=n + 1, seed=77)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_history()))
        if Cls is G.Engine:
            assert e.last_kernel_variant().startswith("k_generations")
            with pytest.raises(RuntimeError, match="history capacity exceeded"):
                e.step(1); e.sync()
    assert_traces_identical(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert len(out[0][1]) == len(Z0) + N * (n // thin)


@pytest.mark.parametrize("N,d,k,depairs,ngamma,prior", [(1000, 100, 5, 3, 2, None), (96, 10, 3, 2, 1, "uniform"), (256, 100, 1, 2, 3, None),
                                                        (64, 128, 4, 3, 1, None), (80, 200, 5, 3, 2, None), (64, 160, 1, 2, 3, "uniform")])       # (the last two: k_generations_d2's full-code instantiations)
def test_persistent_kernel_with_several_pairs_and_gamma_levels(G, O, N, d, k, depairs, ngamma, prior, monkeypatch):
    """DEpairs > 1 and several gamma levels (set_DEpair Dream.py:571-583, set_gamma_level :585-599, the gamma table :692) inside
    k_generations -- the instantiations with the full proposal code -- against the multi-kernel path and the oracle, bit for bit."""
    n, seed = 35, 19
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = H.seed_history(max(10 * d, 2 * N * depairs), d, seed)
    table = np.array([[2.38 / np.sqrt(2.0 * (dl + 1) * np.arange(1, d + 1)) / (lv + 1) for dl in range(depairs)] for lv in range(ngamma)])
    gp = np.arange(1, ngamma + 1, dtype=float); gp /= gp.sum()

    def run(Cls, mega):
        monkeypatch.setenv("DZ_MEGA", "1" if mega else "0")
        monkeypatch.setenv("DZ_MEGA_D2", "2")
        e = Cls(nchains=N, ndim=d, multitry=k, depairs=depairs, ngamma=ngamma, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed)
        e.set_gamma_table(table); e.set_gamma_probs(gp)
        if prior == "uniform":
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -6.0), np.full(d, 22.0))
            e.set_bounds(np.full(d, -6.0), np.full(d, 16.0))
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        launches = None
        if Cls is G.Engine:
            e.profile_enable(True); e.profile_reset()
        e.step(n)
        if Cls is G.Engine:
            launches = e.profile_get("generations")[1]
            e.profile_enable(False)
        return e.get_trace(0, n), e.get_history(), launches

    a, b, o = run(G.Engine, True), run(G.Engine, False), run(O.Engine, False)
    assert a[2] > 0 and b[2] == 0
    for other in (b, o):
        assert_traces_identical(a[0], other[0])
        np.testing.assert_array_equal(a[1], other[1])


@pytest.mark.parametrize("N,d,k,snooker,split", [(256, 10, 3, 0.1, None), (200, 6, 5, 0.3, "5"), (32, 140, 4, 0.1, None), (4096, 100, 5, 0.1, None)])
def test_impossible_proposal_sets_are_drawn_again(G, O, N, d, k, snooker, split, monkeypatch):
    """Dream.py:281-289 at sizes the fixtures do not reach: a uniform prior without hard boundaries and an archive wider than its
    support, so that whole proposal sets are impossible and are generated again (round r from the key seed + r*step).  With the device
    MVN likelihood and d <= 128 the redraw rounds run INSIDE the persistent kernel (its `redo` instantiations: a block-level loop around
    the proposal and likelihood steps); DZ_MEGA_REDO=0 and d > 128 take the multi-kernel path with its host-side check.  States,
    decisions and the archive equal the oracle's bit for bit on both paths, one wave per chain or per (chain, try), one or two
    128-dimension chunks per lane, through a crossover burn-in and beyond it."""
    n, seed = (25 if d < 100 else 12), 41
    if N >= 4096:
        n = 24
    rng = np.random.default_rng(seed)
    P = H.mvn_precision(d)
    lo, width = np.full(d, -2.0), np.full(d, 6.0)
    Z0 = lo + (3 * rng.uniform(0, 1, (max(10 * d, 2 * N), d)) - 1) * width
    Z0[:N] = lo + rng.uniform(0, 1, (N, d)) * width
    if split:
        monkeypatch.setenv("DZ_PROPOSE_SPLIT", split)

    def run(Cls, in_kernel=True):
        monkeypatch.setenv("DZ_MEGA_REDO", "1" if in_kernel else "0")
        e = Cls(nchains=N, ndim=d, multitry=k, hardboundaries=0, snooker=snooker, crossover_burnin=12, adapt_crossover=1,
                history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed)
        e.set_prior(np.full(d, 2, np.int32), lo, width)
        e.set_history(Z0); e.set_state(Z0[:N])
        if N >= 4096:      # (at 16 chains per block only the packed triangle leaves room for the chain states the full-code kernels keep in LDS)
            e.set_likelihood_mvn(np.zeros(d), np.linalg.cholesky((P + P.T) / 2).T, 1, 0.0)
        else:
            e.set_likelihood_mvn(np.zeros(d), P, 0, 0.0)
        if Cls is G.Engine:
            e.profile_enable(True); e.profile_reset()
        e.step(n)
        extra = (e.profile_get("generations")[1], e.redraw_rounds(), e.last_kernel_variant()) if Cls is G.Engine else None
        return e.get_trace(0, n), e.get_history(), e.get_cr_state(), extra

    a, o = run(G.Engine), run(O.Engine)
    b = run(G.Engine, in_kernel=False) if N < 4096 else None
    if d <= 128:
        assert a[3][0] > 0 and a[3][1] > 0 and a[3][2].endswith(",full,redo>"), a[3]      # persistent launches, with redraw rounds inside
    else:
        assert a[3][0] == 0 and a[3][1] > n                                              # no persistent launch; more redraw rounds than generations
    if b is not None:
        assert b[3][0] == 0 and b[3][1] > n
    assert np.isfinite(a[0]["logp"]).all()
    for other in (o, b):
        if other is None:
            continue
        assert_traces_identical(a[0], other[0])
        np.testing.assert_array_equal(a[1], other[1])
        for x, y in zip(a[2], other[2]):
            np.testing.assert_array_equal(x, y)


def test_a_host_likelihood_sees_only_the_redrawn_sets(G, monkeypatch):
    """Redraw rounds (Dream.py:281-289) with a host likelihood: the callback is handed the k points of the chains whose sets were drawn
    again and nothing else -- as the reference evaluates only that chain's new proposals -- one batch per round."""
    fx = H.load("trace_s2_k3_redraw")
    N, k, n = int(fx["cfg_N"]), int(fx["cfg_k"]), int(fx["cfg_G"])
    sizes, plain = [], H.simple_logp

    def counted(X):
        sizes.append(len(X))
        return plain(X)

    monkeypatch.setattr(H, "simple_logp", counted)
    e = H.engine_from_trace_fixture(G.Engine, fx)
    e.step(n)
    tr = e.get_trace(0, n)
    np.testing.assert_array_equal(tr["moved"], fx["moved"])
    regular = [s for s in sizes if s in (N * k, N * (k - 1), N)]           # proposal sets, reference sets, the start states
    redrawn = [s for s in sizes if s not in (N * k, N * (k - 1), N)]
    assert len(regular) >= 2 * n and all(s % k == 0 and 0 < s < N * k for s in redrawn)
    assert len(redrawn) > n // 2 and len(redrawn) <= e.redraw_rounds() <= len(redrawn) + sizes.count(N * k) - n


def test_full_size_run_recovers_the_target_moments(G):
    """BASELINE headline size (4096 chains x 100-D MVN, multitry 5) through properties that do not depend on the size:
    after burn-in the pooled samples have the target's analytic moments (mean 0, Var(x_i) = i, all correlations 0.5;
    dream_ex_ndim_gaussian.py:30-36) and the chains agree with each other (R-hat over a 3000-generation window,
    convergence.py:3-20, computed on the device trace)."""
    N, d, burn, win = 4096, 100, 2000, 3000
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = H.seed_history(2 * N, d, 11)
    e = G.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * ((burn + win) // 10 + 2), trace_capacity=win, seed=2026)
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
    e.profile_enable(True); e.profile_reset()
    e.step(burn)
    e.trace_reset(); e.step(win)
    assert e.profile_get("generations")[1] > 0          # this is the persistent-kernel path
    e.profile_enable(False)
    rhat = e.get_rhat()
    S = np.empty((N, 40, d))
    e.get_trace_chains(win - 40, 40, S)
    X = S[:, ::20, :].reshape(-1, d)                     # pooled sample, 4096 chains x 2 points 20 generations apart
    var_true = np.arange(1, d + 1.0)
    assert np.all(np.abs(X.mean(axis=0)) < 0.08 * np.sqrt(var_true))
    np.testing.assert_allclose(X.var(axis=0) / var_true, 1.0, atol=0.08)
    C = np.corrcoef(X.T)
    off = C[~np.eye(d, dtype=bool)]
    assert abs(off.mean() - 0.5) < 0.02 and off.min() > 0.4 and off.max() < 0.6
    assert rhat.max() < 1.2, rhat.max()
    acc = e.get_trace(win - 40, 40, with_X=False)["moved"].mean()
    assert 0.2 < acc < 0.7
    e.close()


@pytest.mark.parametrize("name", ["trace_pt_mvn10", "trace_pt_adapt"])
def test_parallel_tempering_fixtures(G, O, name):
    """parallel tempering (core.py:131-236): HIP == reference fixture (both interleaved sample streams, swap pairs,
    accepted-swap sequence, history; trace_pt_adapt: with crossover adaptation on, run_dream's default, where the jump after
    an accepted swap is measured from the swapped state, Dream.py:371-378) and HIP == oracle bit for bit."""
    fx = H.load(name)
    n = int(fx["cfg_G"])
    e, o = H.pt_engine_from_fixture(G.Engine, fx), H.pt_engine_from_fixture(O.Engine, fx)
    e.step(n); o.step(n)
    H.compare_pt_with_reference(e, fx)
    assert_traces_identical(e.get_trace(0, n), o.get_trace(0, n))
    np.testing.assert_array_equal(e.get_swaps(0, n), o.get_swaps(0, n))
    for a, b in zip(e.get_cr_state(), o.get_cr_state()):
        np.testing.assert_array_equal(a, b)


def test_parallel_tempering_reference_and_oracle(G, O):
    """parallel tempering at a size the fixtures do not cover, with crossover adaptation on: HIP == oracle bit for bit."""
    # 256 chains x 100-D, 60 generations
    N, d, n, seed = 256, 100, 60, 31
    P = H.mvn_precision(d)
    Z0 = H.seed_history(max(10 * d, 2 * N), d, seed)
    T = np.array([np.power(.001, float(i) / N) for i in range(N)])
    out = []
    for Cls in (G.Engine, O.Engine):
        en = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                 adapt_crossover=1, crossover_burnin=40)
        en.set_history(Z0); en.set_state(Z0[:N]); en.set_likelihood_mvn(np.zeros(d), P, 0, 0.0); en.set_temperatures(T)
        en.step(n)
        out.append((en.get_trace(0, n), en.get_swaps(0, n), en.get_history(), en.get_state(), en.get_cr_state()))
    assert_traces_identical(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    for a, b in zip(out[0][3], out[1][3]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(out[0][4], out[1][4]):
        np.testing.assert_array_equal(a, b)
    assert out[0][1][11:40, 2].sum() > 0                # swaps were accepted inside the adaptation window


@pytest.mark.parametrize("N,d,tri,gemm,zero_mean,qfin,prior", [(160, 333, 1, 1, 0, 1, 0), (200, 200, 0, 1, 0, 1, 0), (160, 333, 1, 0, 0, 1, 0), (128, 1000, 1, 1, 0, 1, 0),
                                                               (128, 200, 1, 1, 1, 1, 0), (128, 200, 0, 1, 1, 1, 0), (160, 333, 1, 1, 0, 0, 0),
                                                               (160, 333, 1, 1, 0, 1, 1), (128, 1000, 1, 1, 0, 0, 1), (160, 333, 1, 1, 0, 1, 2)])
def test_large_d_likelihood_kernels_against_oracle(G, O, N, d, tri, gemm, zero_mean, qfin, prior, monkeypatch):
    """ld > 128: the LDS-tiled product (k_logp_mvn_gemm, >= 512 points per launch) and the register-operand tiled
    kernel (DZ_LOGP_GEMM=0) against the oracle -- dimensions that are not multiples of 16 or 64, non-zero mean, dense and
    triangular matrix; everything bit-exact (the row-tile sums are added in the same order whatever kernel made them, and
    whoever adds them: k_q_finish -- DZ_QFIN=0 -- or the kernels that use the log likelihoods, ld > 256); flat, normal and
    uniform priors, the last with hard boundaries."""
    monkeypatch.setenv("DZ_LOGP_GEMM", str(gemm))
    monkeypatch.setenv("DZ_QFIN", str(qfin))
    n, seed = 3, 9
    P = H.mvn_precision(d)
    M = np.linalg.cholesky((P + P.T) / 2).T if tri else P
    Z0 = H.seed_history(max(10 * d, 2 * N), d, seed)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * 3, trace_capacity=n, seed=seed)
        if prior == 1:
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 2.0, d), np.full(d, 30.0))
        elif prior == 2:                                                      # uniform(-6, 22) with hard boundaries: reflections and redraws (Dream.py:733-791)
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -6.0), np.full(d, 22.0))
            e.set_bounds(np.full(d, -6.0), np.full(d, 16.0))
        e.set_likelihood_mvn(np.zeros(d) if zero_mean else np.linspace(-1, 1, d), M, tri, 0.0)      # (a zero mean skips the subtraction)
        e.set_history(Z0); e.set_state(Z0[:N])
        e.step(n)
        out.append(e.get_trace(0, n))
    assert_traces_identical(out[0], out[1])


@pytest.mark.parametrize("N,d,J,burnin,k", [(512, 100, 3, 0, 5), (70, 20, 2, 0, 5), (256, 100, 3, 15, 5), (512, 100, 3, 0, 1), (70, 20, 2, 12, 1), (64, 10, 2, 0, 3),
                                            (300, 200, 3, 0, 5), (130, 129, 2, 12, 4), (256, 256, 3, 0, 1), (200, 160, 2, -15, 5)])      # 128 < d <= 256 (round 6): <wide>; burnin < 0: with normal priors
def test_persistent_mixture_kernel_equals_multi_kernel_path_and_oracle(G, O, N, d, J, burnin, k, monkeypatch):
    """k_generations_mix (mixture likelihood: every wave carries its chain on its own, no barriers) against the
    multi-kernel path and the oracle, bit for bit: across history appends, a ragged last block, and with a crossover
    burn-in in front (configs[2] shape); multitry 5, 3 and off (one proposal per generation); up to 256 dimensions (a lane then owns four
    dimensions of its chain's state: the <wide> instantiations)."""
    priors = burnin < 0
    burnin = abs(burnin)
    n, seed = 36, 41
    mu = np.array([np.full(d, m) for m in np.linspace(-5.0, 5.0, J)])
    logF = np.log(np.arange(1, J + 1) / np.arange(1, J + 1).sum()) - (d / 2.) * np.log(2 * np.pi)
    Z0 = H.seed_history(max(10 * d, 2 * N), d, seed, lo=-8, hi=8)

    def run(Cls, mega):
        monkeypatch.setenv("DZ_MEGA", "1" if mega else "0")
        e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1 if burnin else 0, crossover_burnin=burnin)
        if priors:
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 1.0, d), np.full(d, 15.0))
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mixture(mu, logF)
        launches = None
        if Cls is G.Engine:
            e.profile_enable(True); e.profile_reset()
        e.step(n)
        if Cls is G.Engine:
            launches = e.profile_get("generations")[1]
            e.profile_enable(False)
        return e.get_trace(0, n), e.get_history(), e.get_cr_state()[0], launches, e.last_kernel_variant() if Cls is G.Engine else ""

    a, b, o = run(G.Engine, True), run(G.Engine, False), run(O.Engine, False)
    assert a[3] > 0 and b[3] == 0
    assert a[4] == "k_generations_mix" + ("<full,wide>" if priors and d > 128 else "<wide>" if d > 128 else "<full>" if priors else "")
    for other in (b, o):
        assert_traces_identical(a[0], other[0])
        np.testing.assert_array_equal(a[1], other[1])
        np.testing.assert_array_equal(a[2], other[2])


def test_eval_logp_scratch_buffers_survive_growth(G, O):
    """dz_eval_logp with a growing number of points at d = 200 (the large-d likelihood path with its row-tile scratch array),
    a trace download that uses the same array, then dz_step: the staging buffer and the row-tile array are separate allocations and
    growing one must not release the other (round-1 advisor finding: a dangling d_qpart after need_scratch)."""
    d, N, seed = 200, 64, 3
    P = H.mvn_precision(d)
    Z0 = H.seed_history(10 * d, d, seed)
    e = G.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * 3, trace_capacity=8, seed=seed)
    o = O.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * 3, trace_capacity=8, seed=seed)
    for x in (e, o):
        x.set_history(Z0); x.set_state(Z0[:N]); x.set_likelihood_mvn(np.linspace(-1, 1, d), P, 0, 0.0)
    for n in (40, 700, 90, 1500):
        got = e.eval_logp(Z0[:n])[1]
        np.testing.assert_array_equal(got, np.array([o.loglike(x) for x in Z0[:n]]))
    e.step(4); o.step(4)
    S = np.empty((N, 4, d)); LP = np.empty((N, 4, 1))
    e.get_trace_chains(0, 4, S, row0=0, logp_out=LP)
    got = e.eval_logp(Z0[:1800])[1]
    np.testing.assert_array_equal(got, np.array([o.loglike(x) for x in Z0[:1800]]))
    e.step(4); o.step(4)
    assert_traces_identical(e.get_trace(0, 8), o.get_trace(0, 8))
    np.testing.assert_array_equal(S, e.get_trace(0, 4)["X"].transpose(1, 0, 2))


def test_c3_full_size_mixture_with_adaptation(G, O):
    """BASELINE configs[2] at full size: 4096 chains x 100-D mixture of three Gaussians (examples/mixturemodel/mixturemodel.py:18-48
    generalised: weights 1/6, 1/3, 1/2, means -5, 0, +5 in every dimension), crossover adaptation on.
    (a) the first 30 generations (all inside the crossover burn-in: the persistent mixture kernel, one generation per launch, followed by the adaptation launches) equal the
        oracle bit for bit, adapted probabilities included;
    (b) size-independent properties of the long run (burn-in ends inside it, the persistent mixture kernel takes over), from a
        seed archive and starts spread over all three modes: every chain ends inside a mode's shell (|x - mu_j|^2 / d ~ 1) and
        inside each mode the pooled sample has the component's moments (mean mu_j, unit variance per dimension).
    (The mode OCCUPANCIES are not asserted: at d = 100 the modes are 50 standard deviations apart, a differential-evolution jump
    between them lands with three times the component's variance in every dimension and is rejected with overwhelming
    probability -- in the reference as here --, so the occupancies stay those of the starts.)"""
    N, d, seed = 4096, 100, 5
    means = (-5.0, 0.0, 5.0)
    w = np.array([1 / 6., 1 / 3., 1 / 2.])
    mu = np.array([np.full(d, m) for m in means])
    logF = np.log(w) - (d / 2.) * np.log(2 * np.pi)
    rng = np.random.default_rng(seed)
    comp = rng.integers(0, 3, 2 * N)
    Z0 = mu[comp] + 2.0 * rng.standard_normal((2 * N, d))                  # over-dispersed around every mode
    n = 30
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1, crossover_burnin=300)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mixture(mu, logF)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_cr_state(), e.get_history()))
    assert_traces_identical(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    # (b) the long run
    total, burn = 6000, 300
    e = G.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (total // 10 + 2), trace_capacity=1000, seed=seed,
                 adapt_crossover=1, crossover_burnin=burn)
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mixture(mu, logF)
    e.profile_enable(True); e.profile_reset()
    for _ in range(total // 1000):
        e.trace_reset(); e.step(1000)
    assert e.profile_get("generations")[1] > 0          # the persistent mixture kernel ran after the burn-in
    e.profile_enable(False)
    X = e.get_state()[0]
    d2 = ((X[:, None, :] - mu[None, :, :]) ** 2).sum(axis=2) / d            # [chain, mode]
    mode = d2.argmin(axis=1)
    shell = (d2.min(axis=1) < 1.6) & (d2.min(axis=1) > 0.55)                # chi^2_100 / 100 lies in (0.55, 1.6) with overwhelming probability
    print("chains inside a mode's shell after %d generations: %.4f; largest |x - mu|^2 / d: %.2f" % (total, shell.mean(), d2.min(axis=1).max()))
    assert shell.mean() > 0.97 and d2.min(axis=1).max() < 4.0               # (the starts sit at 4: a few chains are still on their way in)
    frac = np.bincount(mode, minlength=3) / float(N)
    assert frac.min() > 0.25                                                # all three modes stay populated (the starts' thirds)
    for j in range(3):
        sel = X[(mode == j) & shell]
        np.testing.assert_allclose(sel.var(axis=0).mean(), 1.0, atol=0.08)
        assert np.abs(sel.mean(axis=0) - means[j]).max() < 0.25
    assert not np.allclose(e.get_cr_state()[0], 1 / 3.)
    acc = e.get_trace(900, 100, with_X=False)["moved"].mean()
    assert 0.05 < acc < 0.7
    e.close()


def test_c5_shard_512_chains_1000d_against_oracle(G, O, monkeypatch):
    """BASELINE configs[4] per-GPU shard at full size: 512 chains x 1000-D correlated MVN, triangular-factor likelihood,
    12 generations across two history appends: everything bit-exact against the oracle -- on one stream and with the chains
    split into three chain-group streams (DZ_STREAMS; each stream's likelihood launches use their own slice of the row-tile
    scratch array)."""
    N, d, n, seed = 512, 1000, 12, 9
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = H.seed_history(10 * d, d, seed)
    out = []
    for Cls, streams in ((G.Engine, "1"), (G.Engine, "3"), (O.Engine, None)):
        if streams:
            monkeypatch.setenv("DZ_STREAMS", streams)
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * 4, trace_capacity=n, seed=seed)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_history()))
    for other in out[:2]:
        assert_traces_identical(other[0], out[2][0])
        np.testing.assert_array_equal(other[1], out[2][1])


@pytest.mark.parametrize("variant", ["flat", "uniform_bounds", "normal", "unfused", "dense"])
def test_crossover_burnin_at_4096_chains_against_oracle(G, O, variant, monkeypatch):
    """The crossover burn-in at the headline size (4096 chains x 100-D MVN, 16 chains per block): the persistent kernel makes its
    block's unit sums of the adaptation (reduction contract v3: DESIGN.md section 5) and applies the previous generation's totals in its
    prologue (round 4) -- 45 generations with the burn-in ending inside them (the hand-over at generation 30, then whole thin-cycles per
    launch) equal the oracle bit for bit, adapted probabilities and accumulators included; with uniform priors + hard boundaries (the log
    prior is one constant there), with normal priors (reciprocal scale), with the sums made by k_adapt_partials instead
    (DZ_ADAPT_FUSED=0) and with the dense matrix (chain states in HBM: no fused sums)."""
    if variant == "unfused":
        monkeypatch.setenv("DZ_ADAPT_FUSED", "0")
    N, d, n, seed = 4096, 100, 45, 31
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = H.seed_history(2 * N, d, 6)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1, crossover_burnin=30)
        if variant == "uniform_bounds":
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -10.0), np.full(d, 30.0)); e.set_bounds(np.full(d, -10.0), np.full(d, 20.0))
        elif variant == "normal":
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 1.0, d), np.linspace(20.0, 40.0, d))
        e.set_history(Z0); e.set_state(Z0[:N])
        if variant == "dense":
            e.set_likelihood_mvn(np.zeros(d), P, 0, 0.0)
        else:
            e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_cr_state(), e.get_history(), e.last_kernel_variant() if Cls is G.Engine else ""))
    assert_traces_identical(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert not np.allclose(out[0][1][0], 1 / 3.) and out[0][3].startswith("k_generations<7,")
    assert ("full" in out[0][3]) == (variant in ("uniform_bounds", "normal"))


@pytest.mark.parametrize("target,N,lag,hlag,multi,variant", [
    ("mix", 4096, 9, 1, "1", "k_generations_mix<multi>"),      # what bench.py times configs[2] under: whole thin-cycles per launch inside the burn-in
    ("mix", 4096, 9, 0, "1", "k_generations_mix<multi>"),
    ("mix", 4096, 3, 1, "1", "k_generations_mix<multi>"),      # launches of four generations, history appends in the middle of some
    ("mix", 4096, 9, 1, "0", "k_generations_mix"),             # the same schedule one generation per launch (DZ_ADAPT_MULTI=0): unit sums by k_adapt_partials
    ("mix", 1000, 19, 1, "1", "k_generations_mix<multi>"),     # a ragged last block of 8 chains; twenty generations per launch
    ("mix_pb", 2048, 9, 1, "1", "k_generations_mix<full,multi>"),
    ("mvn", 4096, 9, 1, "1", "k_generations<7,tri,xlds,16,1,lean,multi>"),      # the MVN kernel's multi instantiation (16 chains per block)
    ("mvn", 4096, 19, 1, "1", "k_generations<7,tri,xlds,16,1,lean,multi>"),
    ("mvn_pb", 4000, 9, 0, "1", "k_generations<7,tri,xlds,16,1,full,multi>"),   # ... with normal priors (the full proposal code), a ragged last block
    ("mvn", 4096, 9, 1, "0", "k_generations<7,tri,xlds,16,1,lean>"),            # the same schedule one generation per launch
    # any other persistent kernel: every generation's positions into the ring of published positions, the unit sums behind the launch (k_adapt_partials_ring)
    ("mvn", 300, 2, 0, "1", "k_generations_w4<7, +ring"),                       # 4 chains x 4 waves per block, three generations per launch, a ragged last unit
    ("mvn", 300, 2, 0, "0", "k_generations_w4<7,"),                             # ... and one burn-in generation per launch, the updates applied when due
    ("mvn", 1024, 19, 3, "1", "k_generations_w4<7, +ring"),                     # BASELINE configs[1]'s population, twenty generations per launch
    ("mvn", 2048, 9, 1, "1", "k_generations<7,tri,xlds,8,1,lean> +ring"),       # 8 chains per block
    ("mvn", 3072, 9, 0, "1", "k_generations<7,tri,xlds,12,1,lean> +ring"),      # 12 chains per block
    ("mvn", 4196, 9, 1, "1", "k_generations<7,tri,xlds,16,1,lean> + k_generations_w4<7, +ring"),      # a generation in two launches
    ("mvn_pb", 2000, 3, 1, "1", "k_generations<7,tri,xlds,8,1,full> +ring"),    # the full proposal code
    ("mvn_redo", 2048, 3, 1, "1", "k_generations<7,tri,xlds,8,1,full,redo> +ring"),      # uniform priors without boundaries: the instantiation with the redraw rounds
    ("mvn_k1", 4096, 9, 1, "1", "k_generations<7,tri,xlds,16,1,lean,k1> +ring"),     # multitry off (the reference's default)
    ("mvn_d200", 2048, 9, 1, "1", "k_generations_d2<13, +ring"),                # 128 < d <= 256
    ("mix_k1", 4096, 9, 1, "1", "k_generations_mix +ring"),                     # the mixture kernel without multi-try (blocks of 4 waves)
    ("mix_d200", 1024, 9, 1, "1", "k_generations_mix<wide> +ring"),             # ... and at 128 < d <= 256
])
def test_crossover_burnin_with_an_adapt_lag_against_oracle(G, O, target, N, lag, hlag, multi, variant, monkeypatch):
    """dz_config.adapt_lag = L (round 6): generation g of the burn-in decides with the probabilities as they were after the updates of
    generations <= g - 1 - L, so a launch of the mixture kernel holds up to L + 1 burn-in generations (its blocks make their unit sums generation
    by generation, the pending updates are applied in the prologue).  55 generations, the window of the adaptation (generations 11 .. burn-in - 1,
    Dream.py:371) and the hand-over (generation 38: everything still held is applied, Dream.py:385-415) inside them, then whole thin-cycles:
    equal to the oracle bit for bit -- decisions, states, archive, adapted probabilities and accumulators -- and the state handed out
    after EVERY dz_step of a run stepped in pieces is the oracle's (updates still pending are not in it)."""
    monkeypatch.setenv("DZ_ADAPT_MULTI", multi)
    if "d200" in target:
        monkeypatch.setenv("DZ_MEGA_D2", "2")
    d, n, seed, burn = (200 if "d200" in target else 100), 55, 77, 38
    k = 1 if target.endswith("_k1") else 5
    rng = np.random.default_rng(seed)
    mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
    logF = np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi)
    Z0 = mu[rng.integers(0, 3, 2 * N + 100)] + 2.0 * rng.standard_normal((2 * N + 100, d)) if target.startswith("mix") else H.seed_history(2 * N + 100, d, 6)
    pieces = (7, 13, 1, 20, 14)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1, crossover_burnin=burn, adapt_lag=lag, history_lag=hlag)
        if target in ("mix_pb", "mvn_pb"):
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 1.0, d), np.linspace(20.0, 40.0, d))
        if target == "mvn_redo":
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -12.0), np.full(d, 40.0))
        e.set_history(Z0); e.set_state(Z0[:N])
        if target.startswith("mix"):
            e.set_likelihood_mixture(mu, logF)
        else:
            P = H.mvn_precision(d)
            e.set_likelihood_mvn(np.zeros(d), np.linalg.cholesky((P + P.T) / 2).T, 1, 0.0)
        states, variants = [], set()
        for m in pieces:
            e.step(m)
            states.append(e.get_cr_state())
            if Cls is G.Engine:
                variants.add(e.last_kernel_variant())
        out.append((e.get_trace(0, n), states, e.get_history(), variants))
    assert_traces_identical(out[0][0], out[1][0])
    for sa, sb in zip(out[0][1], out[1][1]):
        for a, b in zip(sa, sb):
            np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert not np.allclose(out[0][1][-1][0], 1 / 3.)
    ring = variant.endswith(" +ring")
    prefix = variant[:-len(" +ring")] if ring else variant
    assert any(v.startswith(prefix) and v.endswith(" +ring") == ring for v in out[0][3]), out[0][3]
    assert any("multi>" in v for v in out[0][3]) == (multi == "1" and not ring)
    assert any(v.endswith("+ring") for v in out[0][3]) == ring


@pytest.mark.parametrize("adapt", [0, 1])
def test_a_chain_count_just_above_whole_rounds_of_blocks_against_oracle(G, O, adapt):
    """4196 chains are 263 blocks of 16: one more than the CUs hold at once.  The generation then goes in two launches -- 4096 chains in
    blocks of 16, the remaining 100 in blocks of 4 chains x 4 waves (run_mega_segment; 5000 chains: 460 -> 574 M proposals/s) -- and
    still equals the oracle bit for bit, with and without the crossover burn-in (whose unit sums then come from k_adapt_partials)."""
    N, d, n, seed = 4196, 100, 25, 17
    P = H.mvn_precision(d)
    U = np.linalg.cholesky((P + P.T) / 2).T
    Z0 = H.seed_history(2 * N, d, 8)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=adapt, crossover_burnin=15 if adapt else 0)
        e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), U, 1, 0.0)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_cr_state(), e.get_history(), e.last_kernel_variant() if Cls is G.Engine else ""))
    assert_traces_identical(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert out[0][3] == "k_generations<7,tri,xlds,16,1,lean> + k_generations_w4<7,tri,xlds,4,4,lean,ahead>", out[0][3]


@pytest.mark.parametrize("k,d,N,lk,variant", [(16, 100, 300, "tri", "multi-kernel path"), (17, 10, 64, "dense", "multi-kernel path"), (24, 140, 48, "tri", "multi-kernel path"),
                                              (32, 33, 40, "mix", "multi-kernel path"), (32, 300, 16, "tri", "multi-kernel path"),
                                              # (round 6) k_generations_d2 keeps one phase's draw slots at a time: 16..32 tries inside a persistent kernel (triangular factor, > 1024 chains)
                                              (16, 100, 1100, "tri", "k_generations_d2<7,tri,xhbm,8,2,lean>"), (20, 100, 1040, "tri", "k_generations_d2<7,tri,xhbm,8,2,lean>"),
                                              (32, 10, 1030, "tri", "k_generations_d2<1,tri,xhbm,16,1,lean>"), (17, 48, 1100, "tri", "k_generations_d2<3,tri,xhbm,16,1,lean>"),
                                              (16, 132, 1040, "tri", "k_generations_d2<9,tri,xhbm,8,2,lean>"), (16, 200, 1040, "tri", "k_generations_d2<13,tri,xhbm,4,4,lean>"), (24, 100, 1040, "tri", "k_generations_d2<7,tri,xhbm,4,4,lean>"), (32, 128, 1030, "tri", "k_generations_d2<8,tri,xhbm,4,4,lean>"),      # (4 chains x 4 waves where 8 x 2 do not fit)
                                              (15, 100, 1100, "tri", "k_generations_d2<7,tri,xhbm,8,2,lean>")])
def test_more_than_sixteen_tries_against_oracle(G, O, k, d, N, lk, variant):
    """The reference takes any integer as `multitry` (Dream.py:155-161).  Up to 15 tries a generation's draw slots fit one wave's lanes and the
    persistent kernels run; 16..32 tries run the multi-kernel path or -- round 6: triangular factor, more than 1024 chains, the point tiles of 16 chains or of
    8 chains x 2 waves fitting LDS -- k_generations_d2 (selection and multi-try ratio over 32 + 32 lanes either way): states, decisions,
    archive and adapted probabilities equal the oracle's bit for bit, through a crossover burn-in, snooker sets and history appends, for the
    per-chain (d <= 256) and the streamed (d = 300) proposal kernels."""
    n, seed = 24, 7
    Z0 = H.seed_history(max(10 * d, 2 * N), d, seed, lo=-2.0, hi=2.0)          # (near the targets' mass: moves are accepted from the first generations on)
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed, snooker=0.2,
                adapt_crossover=1, crossover_burnin=14)
        e.set_history(Z0); e.set_state(Z0[:N])
        if lk == "mix":
            mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
            e.set_likelihood_mixture(mu, np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi))
        else:
            P = H.mvn_precision(d)
            e.set_likelihood_mvn(np.zeros(d), H.tri_factor(P) if lk == "tri" else P, 1 if lk == "tri" else 0, 0.0)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_history(), e.get_cr_state(), e.last_kernel_variant() if Cls is G.Engine else ""))
    assert_traces_identical(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        np.testing.assert_array_equal(a, b)
    assert out[0][0]["try_idx"].max() >= min(k - 1, 12) and 0.005 < out[0][0]["moved"].mean() < 0.98
    assert out[0][3] == variant, out[0][3]
