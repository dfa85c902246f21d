"""Shared by the CPU (oracle) and GPU (HIP engine) parity tests: build an engine from a golden
fixture's recorded inputs, run it, and compare with what the REFERENCE produced."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def simple_logp(X):
    """pydream/tests/test_models.py:47-50 (likelihood) -- prior handled by the engine's built-in prior."""
    return np.zeros(len(X)), np.sum(X + 3, axis=1)


def tri_factor(P):
    """The upper-triangular factor U of a precision matrix, P = U^T U -- what pydream_amd.likelihoods.MVNormalLogLike builds and
    bench.py's headline times (`--mvn-kind tri`): log p = log_F - |U x|^2 / 2, half the flops of x.(P x)."""
    return np.linalg.cholesky((P + P.T) / 2).T


def engine_from_trace_fixture(EngineCls, fx, schedule=None, trace=True, mvn_kind="dense", **over):
    """mvn_kind="tri": the fixture's MVN likelihood handed over as the triangular factor of its precision matrix (kind 1) instead of the
    dense matrix the reference itself multiplied with (kind 0)."""
    d, N, G, k = int(fx["cfg_d"]), int(fx["cfg_N"]), int(fx["cfg_G"]), int(fx["cfg_k"])
    thin = int(fx["cfg_history_thin"])
    Z0 = fx["Z0"]
    kw = dict(nchains=N, ndim=d, multitry=k, depairs=int(fx["cfg_DEpairs"]), ncr=int(fx["cfg_nCR"]),
              ngamma=int(fx["cfg_gamma_levels"]), history_thin=thin, crossover_burnin=int(min(fx["burnin"], 2 ** 31 - 1)),
              adapt_crossover=int(fx["cfg_adapt_crossover"]), adapt_gamma=int(fx["cfg_adapt_gamma"]),
              hardboundaries=int(fx["cfg_hardboundaries"]), schedule=int(fx["cfg_schedule"]) if schedule is None else schedule,
              history_capacity=len(Z0) + N * (G // thin + 2), trace_capacity=G if trace else 0, seed=int(fx["cfg_seed"]),
              lamb=float(fx["cfg_lamb"]), zeta=float(fx["cfg_zeta"]), snooker=float(fx["cfg_snooker"]),
              p_gamma_unity=float(fx["cfg_p_gamma_unity"]))
    if "cfg_history_lag" in fx:
        kw["history_lag"] = int(fx["cfg_history_lag"])
    if "cfg_adapt_lag" in fx:
        kw["adapt_lag"] = int(fx["cfg_adapt_lag"])
    kw.update(over)
    e = EngineCls(**kw)
    if "mins" in fx:
        e.set_bounds(fx["mins"], fx["maxs"])
    e.set_gamma_table(fx["gamma_arr"])
    e.set_history(Z0)
    e.set_prior(fx["prior_kind"], fx["prior_a"], fx["prior_b"])
    if "restart_cr_probs" in fx:                       # a restarted run: Dream's `crossover_file` (Dream.py:128-134)
        e.set_cr_probs(fx["restart_cr_probs"])
    lk = str(fx["lk_kind"])
    if lk == "mvn" and mvn_kind == "tri":
        e.set_likelihood_mvn(np.zeros(d), tri_factor(fx["invC"]), 1, float(fx["log_F"]))
    elif lk == "mvn":
        e.set_likelihood_mvn(np.zeros(d), fx["invC"], 0, float(fx["log_F"]))
    elif lk == "mix":
        e.set_likelihood_mixture(fx["mix_mu"], fx["mix_logF"])
    else:
        e.set_likelihood_host(simple_logp)
    nl = kw.get("nchains_local", N)
    off = kw.get("chain_offset", 0)
    e.set_state(fx["starts"][off:off + nl])
    return e


def compare_with_reference(tr, fx, Z, cr_probs, gamma_probs=None, x_rtol=1e-9, logp_atol=1e-10):
    """tr: engine trace dict; fx: fixture (reference outputs)."""
    np.testing.assert_array_equal(tr["snooker"], fx["snooker"])
    np.testing.assert_array_equal(tr["cr_idx"], fx["cr_idx"])
    if int(fx["cfg_k"]) > 1:
        np.testing.assert_array_equal(tr["try_idx"], fx["try_idx"])
    np.testing.assert_array_equal(tr["moved"], fx["moved"])
    np.testing.assert_allclose(tr["X"], fx["X"], rtol=x_rtol, atol=1e-11)
    np.testing.assert_allclose(tr["logp"], fx["logp"], rtol=0, atol=logp_atol)
    np.testing.assert_allclose(Z[len(fx["Z0"]):], fx["Z_tail"], rtol=x_rtol, atol=1e-11)
    np.testing.assert_allclose(cr_probs, fx["cross_probs"][-1], rtol=1e-11, atol=0)
    if gamma_probs is not None:
        np.testing.assert_allclose(gamma_probs, fx["gamma_probs"][-1], rtol=1e-11, atol=0)


def mvn_precision(d):
    """Precision matrix of the reference's correlated MVN target
    (pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:30-38) at dimension d."""
    i = np.arange(1, d + 1, dtype=float)
    C = (.5 * np.identity(d) + .5 * np.ones((d, d))) * np.sqrt(np.outer(i, i))
    return np.linalg.inv(C)


def seed_history(rows, d, seed, lo=-5.0, hi=15.0):
    """iid U(lo,hi) seed archive (the shipped example uses a Latin hypercube on [-5,15],
    dream_ex_ndim_gaussian.py:17-26, 45)."""
    return np.random.default_rng(seed).uniform(lo, hi, (rows, d))


def pt_engine_from_fixture(EngineCls, fx, **over):
    """engine for the parallel-tempering fixture (tests/golden/trace_pt_*.npz; reference defaults otherwise)."""
    d, N, G, k = int(fx["cfg_d"]), int(fx["cfg_N"]), int(fx["cfg_G"]), int(fx["cfg_k"])
    Z0 = fx["Z0"]
    kw = dict(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (G // 10 + 2), trace_capacity=G, seed=int(fx["cfg_seed"]),
              adapt_crossover=int(fx["cfg_adapt_crossover"]) if "cfg_adapt_crossover" in fx else 0,
              crossover_burnin=int(fx["burnin"]) if "burnin" in fx else G // 10)
    kw.update(over)
    e = EngineCls(**kw)
    e.set_history(Z0)
    e.set_likelihood_mvn(np.zeros(d), fx["invC"], 0, float(fx["log_F"]))
    e.set_state(fx["starts"][:N])
    e.set_temperatures(fx["T"], swaps=True)
    return e


def pt_arrays(tr, swaps):
    """sampled_params / log_ps of core._sample_dream_pt (core.py:144-145, :179-181, :223-225) from an engine's
    pre-swap trace and its swap log: rows 2i = after the steps of iteration i, rows 2i+1 = after its swap attempt."""
    X, lp = tr["X"], tr["logp"]                        # [G, N, d], [G, N]
    G, N, d = X.shape
    S = np.zeros((N, 2 * G, d)); L = np.zeros((N, 2 * G, 1))
    for g in range(G):
        xa, la = X[g].copy(), lp[g].copy()
        S[:, 2 * g], L[:, 2 * g, 0] = xa, la
        a, b, acc = swaps[g]
        if acc:
            xa[[a, b]] = xa[[b, a]]; la[[a, b]] = la[[b, a]]
        S[:, 2 * g + 1], L[:, 2 * g + 1, 0] = xa, la
    return S, L


def compare_pt_with_reference(e, fx, x_rtol=1e-9, logp_atol=1e-10):
    G = int(fx["cfg_G"])
    tr = e.get_trace(0, G)
    sw = e.get_swaps(0, G)
    np.testing.assert_array_equal(sw[:, :2], fx["pt_swaps"])
    np.testing.assert_array_equal(tr["snooker"], fx["snooker"])
    np.testing.assert_array_equal(tr["cr_idx"], fx["cr_idx"])
    np.testing.assert_array_equal(tr["try_idx"], fx["try_idx"])
    S, L = pt_arrays(tr, sw)
    ref_acc = np.array([not np.array_equal(fx["pt_sampled"][fx["pt_swaps"][g, 0], 2 * g], fx["pt_sampled"][fx["pt_swaps"][g, 0], 2 * g + 1]) for g in range(G)])
    np.testing.assert_array_equal(sw[:, 2].astype(bool), ref_acc)          # the accepted-swap sequence
    np.testing.assert_allclose(S, fx["pt_sampled"], rtol=x_rtol, atol=1e-11)
    np.testing.assert_allclose(L, fx["pt_log_ps"], rtol=0, atol=logp_atol)
    np.testing.assert_allclose(e.get_history()[len(fx["Z0"]):], fx["Z_tail"], rtol=x_rtol, atol=1e-11)
    if "cfg_adapt_crossover" in fx and int(fx["cfg_adapt_crossover"]):
        pr, dm, nu = e.get_cr_state()
        np.testing.assert_allclose(pr, fx["cross_probs"][-1], rtol=1e-11)
        np.testing.assert_allclose(dm, fx["delta_m"], rtol=1e-11)
        np.testing.assert_array_equal(nu, fx["ncr_updates"])
    return tr, sw
