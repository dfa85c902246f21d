"""The oracle (oracle/dreamzs_oracle.c) against the vectors produced by the REFERENCE itself
(tests/golden/make_golden.py): this is what pins the oracle.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

TRACES = ["trace_s1_c1", "trace_s1_adapt", "trace_s1_adapt_gamma", "trace_s2_adapt", "trace_s2_k1_bounds", "trace_s2_k3_bounds",
          "trace_s2_k3_redraw", "trace_s2_k5_redraw_mvn",
          "trace_s2_depairs_gamma", "trace_s2_mvn100", "trace_s2_mix3", "trace_s2_restart", "trace_s2_lag1", "trace_s2_lag2_k1", "trace_s2_lag3",
          "trace_s2_adaptlag1", "trace_s2_adaptlag9_mix", "trace_s2_adaptlag3_gamma"]


@pytest.mark.parametrize("name", TRACES)
def test_astep_traces_match_reference(name):
    """Accept/selection/CR/snooker sequences exact; states to 1e-9 relative (snooker projections cancel; the reference sums them in BLAS order); logp to 1e-10 (north_star)."""
    fx = H.load(name)
    if "redraw" in name:       # the case is only worth its name if the reference did draw whole proposal sets again (Dream.py:281-289)
        assert (fx["redraws"] > 0).mean() > 0.3 and fx["redraws"].max() < 64
    e = H.engine_from_trace_fixture(O.Engine, fx)
    G = int(fx["cfg_G"])
    if int(fx["cfg_schedule"]) == 2 and int(fx["cfg_adapt_crossover"]):
        # generation by generation: the shared crossover probabilities as the reference's chains adopted them after every generation
        # (what pins adapt_lag: with a lag they are those of `adapt_lag` generations earlier)
        for g in range(G):
            e.step(1)
            np.testing.assert_allclose(e.get_cr_state()[0], fx["cross_probs"][g], rtol=1e-11, atol=0, err_msg="generation %d" % g)
            if int(fx["cfg_adapt_gamma"]):
                np.testing.assert_allclose(e.get_gamma_state()[0], fx["gamma_probs"][g], rtol=1e-11, atol=0, err_msg="generation %d" % g)
    else:
        e.step(G)
    tr = e.get_trace(0, G)
    gp = e.get_gamma_state()[0] if int(fx["cfg_adapt_gamma"]) else None
    H.compare_with_reference(tr, fx, e.get_history(), e.get_cr_state()[0], gp)
    if int(fx["cfg_adapt_crossover"]):
        _, dm, nu = e.get_cr_state()
        np.testing.assert_allclose(dm, fx["delta_m"], rtol=1e-11)
        np.testing.assert_array_equal(nu, fx["ncr_updates"])
    if int(fx["cfg_adapt_gamma"]):
        _, dm, nu = e.get_gamma_state()
        np.testing.assert_allclose(dm, fx["delta_m_gamma"], rtol=1e-11)
        np.testing.assert_array_equal(nu, fx["ngamma_updates"])


@pytest.mark.parametrize("name,wrong", [("trace_s2_adaptlag1", 0), ("trace_s2_adaptlag1", 2), ("trace_s2_adaptlag9_mix", 8), ("trace_s2_adaptlag3_gamma", 0),
                                        ("trace_s2_adapt", 1)])
def test_adapt_lag_fixtures_pin_their_own_lag(name, wrong):
    """The reference-made adapt_lag fixtures (tests/golden/make_golden.py adapt_lag_cases: the reference's own estimate_* methods replayed
    L generations late) are reproduced at their own L only: with any other lag the crossover decisions inside the burn-in differ."""
    fx = H.load(name)
    e = H.engine_from_trace_fixture(O.Engine, fx, adapt_lag=wrong)
    G = int(fx["cfg_G"])
    e.step(G)
    tr = e.get_trace(0, G)
    b = int(fx["burnin"])
    assert not np.array_equal(tr["cr_idx"][:b + 1], fx["cr_idx"][:b + 1])
    assert np.array_equal(tr["cr_idx"][:12], fx["cr_idx"][:12])          # (no update before generation 11: Dream.py:371)


MVN_TRACES = [n for n in TRACES if str(H.load(n)["lk_kind"]) == "mvn"]


@pytest.mark.parametrize("name", MVN_TRACES)
def test_astep_traces_match_reference_with_the_triangular_factor(name):
    """The likelihood form bench.py's headline times -- log p = log_F - |U x|^2 / 2 with U the triangular factor of the precision matrix --
    against the decision sequences the REFERENCE made with its own formula x.(invC.x) (dream_ex_ndim_gaussian.py:49-52): the two forms
    differ in the last bits of log p, and a difference of 1e-13 can flip `log(u) < ratio` (Dream.py:993) or a multi-try selection (:908).
    Every reference-made MVN fixture (100-D, 10-D, with lag, adaptation, restart, redraw rounds) is replayed with the factor: snooker
    flags, CR indices, selected tries and accept flags exact, log p to 1e-10, states to 1e-9."""
    fx = H.load(name)
    e = H.engine_from_trace_fixture(O.Engine, fx, mvn_kind="tri")
    G = int(fx["cfg_G"])
    e.step(G)
    gp = e.get_gamma_state()[0] if int(fx["cfg_adapt_gamma"]) else None
    H.compare_with_reference(e.get_trace(0, G), fx, e.get_history(), e.get_cr_state()[0], gp)


@pytest.mark.parametrize("tag", ["d100k5", "d4k5b", "d4k1b", "d10k1"])
def test_generate_proposal_points_match_reference(tag):
    """Dream.generate_proposal_points / snooker_update (Dream.py:670-837) on hand-built history:
    DE proposals bit-exact, snooker proposals to 1e-12 relative (BLAS-ordered dot/norm in the reference)."""
    fx = H.load("proposals")
    g = lambda a: fx[tag + "__" + a]
    d, k = int(g("d")), int(g("k"))
    Z, q0 = g("Z"), g("q0")
    e = O.Engine(nchains=4, ndim=d, multitry=k, depairs=int(g("depairs")), ngamma=int(g("ngamma")),
                 history_capacity=len(Z) + 8, seed=int(g("seed")), lamb=float(g("lamb")))
    e.set_history(Z)
    if int(g("bounded")):
        e.set_bounds(g("mins"), g("maxs"))
    pos = 0
    spos = 0
    n_exact = 0
    for (trial, phase, snk, cr_idx, delta, glev) in g("meta"):
        n = k if phase == 0 else k - 1
        pts, slogp, gam, _ = e.debug_propose(3, int(trial), int(phase), q0, int(snk), int(cr_idx), int(delta), int(glev))
        ref_pts = g("pts")[pos:pos + n * d].reshape(n, d)
        ref_sl = g("slogp")[spos:spos + n]
        ref_gam = g("gam")[spos:spos + n]
        pos += n * d
        spos += n
        np.testing.assert_array_equal(gam, ref_gam)
        if snk:
            np.testing.assert_allclose(pts, ref_pts, rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(slogp, ref_sl, rtol=1e-12, atol=1e-12)
        else:
            np.testing.assert_array_equal(pts, ref_pts)
            n_exact += 1
    assert n_exact > 0 and pos == len(g("pts"))


def test_mvn_logpdf_matches_reference(golden_dir):
    """examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52 at d=10,100 (exact log_F) and the shipped
    d=200 module itself (log_F=0 branch); dense precision and its triangular factor; 1e-10 absolute."""
    fx = H.load("densities")
    for d in (10, 100, 200):
        P, X, ref = fx["mvn%d_invC" % d], fx["mvn%d_X" % d], fx["mvn%d_logp" % d]
        e = O.Engine(nchains=3, ndim=d, history_capacity=8)
        e.set_likelihood_mvn(np.zeros(d), P, 0, float(fx["mvn%d_logF" % d]))
        got = np.array([e.loglike(x) for x in X])
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10)
        U = np.linalg.cholesky((P + P.T) / 2).T
        e.set_likelihood_mvn(np.zeros(d), U, 1, float(fx["mvn%d_logF" % d]))
        got = np.array([e.loglike(x) for x in X])
        np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-10)


def test_mixture_logpdf_matches_reference():
    """examples/mixturemodel/mixturemodel.py:37-48 (the shipped 2-component module) and the 3-component C3 target."""
    fx = H.load("densities")
    for tag, d in (("mix2", 10), ("mix3", 100)):
        e = O.Engine(nchains=3, ndim=d, history_capacity=8)
        e.set_likelihood_mixture(fx[tag + "_mu"], fx[tag + "_logF"])
        got = np.array([e.loglike(x) for x in fx[tag + "_X"]])
        np.testing.assert_allclose(got, fx[tag + "_logp"], rtol=0, atol=1e-10)


def test_gamma_table_matches_reference():
    """Dream.py:172-179; values pinned by pydream/tests/test_dream.py:68-76."""
    fx = H.load("densities")
    np.testing.assert_array_equal(O.gamma_table(4, 5, 7), fx["gamma_arr_7_5_4"])
    true_gamma = np.array([1.683, 1.19, .972, .841, .753])
    np.testing.assert_allclose(O.gamma_table(1, 5, 1)[0, :, 0], true_gamma, atol=5e-4)


def test_gelman_rubin_matches_reference():
    """convergence.py:3-20"""
    fx = H.load("densities")
    np.testing.assert_allclose(O.gelman_rubin(fx["gr_traces"]), fx["gr_rhat"], rtol=1e-12)


def test_restart_fixture_continues_the_earlier_run():
    """trace_s2_restart restarts trace_s2_adapt the way run_dream(restart=True) does (core.py:46-62, 255-263; Dream.py:128-141):
    its seed history is everything the first run left (seed rows + appended rows), its starts are the first run's last states and
    its crossover probabilities the adapted ones -- checked here on the fixtures themselves; the replay is in TRACES above."""
    a, b = H.load("trace_s2_adapt"), H.load("trace_s2_restart")
    np.testing.assert_array_equal(b["Z0"], np.concatenate([a["Z0"], a["Z_tail"]]))
    np.testing.assert_array_equal(b["starts"], a["X"][-1])
    np.testing.assert_array_equal(b["restart_cr_probs"], a["cross_probs"][-1])
    assert not np.allclose(b["restart_cr_probs"], 1. / 3)                      # the loaded probabilities really are adapted ones


@pytest.mark.parametrize("name", ["trace_pt_mvn10", "trace_pt_adapt"])
def test_parallel_tempering_matches_reference(name):
    """core._sample_dream_pt (core.py:131-236) itself, run in one process by tests/golden/make_golden.py: temperature
    ladder, per-chain T inside astep, one swap attempt per iteration -- oracle == reference on both interleaved sample
    streams, the swap pairs, the accepted-swap sequence and the history."""
    fx = H.load(name)
    e = H.pt_engine_from_fixture(O.Engine, fx)
    e.step(int(fx["cfg_G"]))
    tr, sw = H.compare_pt_with_reference(e, fx)
    assert 3 <= sw[:, 2].sum() <= 60
    if name == "trace_pt_adapt":        # accepted swaps inside the adaptation window: the case the jump baseline matters for
        assert sw[11:int(fx["burnin"]), 2].sum() >= 2
