"""End-to-end through the reference-shaped Python API on the GPU -- after the reference's
Test_Dream_Algorithm_Components / Test_Dream_Full_Algorithm (pydream/tests/test_dream.py:499-705)."""
import os

import numpy as np
import pytest
from scipy.stats import uniform

from pydream_amd import Dream_shared_vars
from pydream_amd.Dream import Dream
from pydream_amd.convergence import Gelman_Rubin
from pydream_amd.core import _setup_mp_dream_pool, run_dream
from pydream_amd.likelihoods import MVNormalLogLike
from pydream_amd.model import Model
from pydream_amd.parameters import FlatParam, SampledParam
from tests import helpers as H
from tests.test_api_cpu import multidmodel, multidmodel_uniform, onedmodel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["trace_s1_c1", "trace_s1_adapt", "trace_s1_adapt_gamma"])
def test_astep_round_robin_equals_reference_s1(tmp_path, name):
    """Dream.astep driven round-robin in one process, the reference's own idiom (test_dream.py:507-518):
    identical to the UNMODIFIED reference's trace and bit-identical to the oracle's schedule S1.
    trace_s1_c1: 3 chains, 10-D MVN, multitry 5, adaptation off.  trace_s1_adapt: crossover adaptation on -- the reference's
    default -- with the burn-in ending inside the run: every Dream instance decides with its OWN copy of the crossover probabilities
    (Dream.py:375, :497, :409-415), the standard deviations are summed in numpy's row order.  trace_s1_adapt_gamma: gamma-level
    adaptation as well (three levels, two DE pairs): the instances' own gamma-level probabilities (Dream.py:383, :538)."""
    import copy
    from oracle import oracle as O
    fx = H.load(name)
    d, N, G = int(fx["cfg_d"]), int(fx["cfg_N"]), int(fx["cfg_G"])
    adapt = bool(int(fx["cfg_adapt_crossover"]))
    adapt_g = bool(int(fx["cfg_adapt_gamma"]))
    hist = tmp_path / "seed.npy"
    np.save(hist, fx["Z0"])
    like = MVNormalLogLike(fx["invC"], log_F=float(fx["log_F"]), factorize=False)
    step = Dream(model=Model(like, [FlatParam(np.zeros(d))]), history_file=str(hist), start_random=False, save_history=False,
                 multitry=5, adapt_crossover=adapt, crossover_burnin=int(min(fx["burnin"], 10 ** 9)), adapt_gamma=adapt_g,
                 gamma_levels=int(fx["cfg_gamma_levels"]), DEpairs=int(fx["cfg_DEpairs"]))
    pool = _setup_mp_dream_pool(N, G, step, start_pt=[fx["starts"][i] for i in range(N)], seed=int(fx["cfg_seed"]))
    pool._initializer(*pool._initargs)
    try:
        # the k-th Dream instance claims chain N-1-k (Dream.py:198-200); drive them so that chain 0 goes first
        chains = [copy.copy(step) for _ in range(N)]
        order = list(range(N - 1, -1, -1))
        for k in range(N):
            chains[k].chain_n = None
        x = {c: fx["starts"][c].copy() for c in range(N)}
        X = np.zeros((G, N, d)); lp = np.zeros((G, N))
        claimed = {}
        for g in range(G):
            for c in range(N):
                inst = chains[order[c]] if g == 0 else claimed[c]
                if g == 0:
                    # make instance -> chain mapping explicit
                    Dream_shared_vars.nchains_counter = c + 1
                    claimed[c] = inst
                q, pr, lk = inst.astep(x[c])
                assert isinstance(q, np.ndarray) and isinstance(pr, float) and isinstance(lk, float)
                x[c] = q; X[g, c] = q; lp[g, c] = pr + lk
        np.testing.assert_allclose(X, fx["X"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(lp, fx["logp"], rtol=0, atol=1e-10)
        np.testing.assert_array_equal(np.any(np.diff(np.concatenate([fx["starts"][None, :N], X]), axis=0) != 0, axis=2), fx["moved"].astype(bool))
        o = H.engine_from_trace_fixture(O.Engine, fx)      # schedule S1
        o.step(G)
        np.testing.assert_array_equal(X, o.get_trace(0, G)["X"])
        np.testing.assert_array_equal(pool.engine.get_history(), o.get_history())
        if adapt:
            cr, dm, nu = pool.engine.get_cr_state()
            np.testing.assert_allclose(cr, fx["cross_probs"][-1], rtol=1e-11)
            np.testing.assert_allclose(dm, fx["delta_m"], rtol=1e-11)
            np.testing.assert_array_equal(nu, fx["ncr_updates"])
            for a, b in zip(pool.engine.get_cr_state(), o.get_cr_state()):
                np.testing.assert_array_equal(a, b)
            assert len(np.unique(fx["cr_idx"])) == 3 and not np.allclose(cr, 1 / 3.)      # the probabilities did move
            # every instance keeps the vector as it stood after ITS OWN last update (Dream.py:409-415 in one process): the chain driven
            # last holds the final shared one, the one driven first an earlier one
            np.testing.assert_array_equal(np.asarray(claimed[N - 1].CR_probabilities), cr)
            assert not np.array_equal(np.asarray(claimed[0].CR_probabilities), cr)
        if adapt_g:
            gp, gdm, gnu = pool.engine.get_gamma_state()
            np.testing.assert_allclose(gp, fx["gamma_probs"][-1], rtol=1e-11)
            np.testing.assert_allclose(gdm, fx["delta_m_gamma"], rtol=1e-11)
            np.testing.assert_array_equal(gnu, fx["ngamma_updates"])
            for a, b in zip(pool.engine.get_gamma_state(), o.get_gamma_state()):
                np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(np.asarray(claimed[N - 1].gamma_probabilities), gp)
    finally:
        pool.close(); pool.join()


def test_the_pydream_alias_package_runs_a_reference_script_unchanged():
    """`import pydream` resolves to this repository's alias package (pydream/__init__.py): a PyDREAM script's own imports
    (examples/ndim_gaussian/dream_ex_ndim_gaussian.py:8-12) and calls run on the GPU engine."""
    from pydream.core import run_dream as rd
    from pydream.parameters import SampledParam as SP
    from pydream.convergence import Gelman_Rubin as GR
    import pydream.Dream as PD
    import pydream_amd.core
    assert rd is pydream_amd.core.run_dream and PD.Dream is Dream
    from scipy.stats import norm
    sampled, log_ps = rd([SP(norm, loc=np.zeros(3), scale=np.ones(3))], lambda x: -0.5 * float(np.sum(x * x)), nchains=5, niterations=60,
                         verbose=False, save_history=False, seed=3, multitry=3)
    assert len(sampled) == 5 and sampled[0].shape == (60, 3) and log_ps[0].shape == (60, 1)
    assert GR(sampled).shape == (3,)


def test_run_dream_shapes_and_types():
    """test_dream.py:564-584, 708-806: list of nchains arrays (niterations, d) and (niterations, 1)."""
    for params, like in (onedmodel(), multidmodel(), multidmodel_uniform()):
        sampled, log_ps = run_dream(params, like, nchains=5, niterations=30, verbose=False, save_history=False, seed=1)
        assert len(sampled) == 5 and len(log_ps) == 5
        d = sum(p.dsize for p in params)
        for s, l in zip(sampled, log_ps):
            assert s.shape == (30, d) and l.shape == (30, 1) and np.all(np.isfinite(l))
    sampled, _ = run_dream(*multidmodel(), nchains=5, niterations=20, verbose=False, save_history=False, multitry=5, seed=2,
                           DEpairs=2, gamma_levels=3, adapt_gamma=True, snooker=.2)
    assert sampled[0].shape == (20, 4)


def test_history_file_matches_samples(tmp_path):
    """test_dream.py:629-668: every appended history row is one of the returned samples (thin=1: all of them)."""
    os.chdir(tmp_path)
    params, like = multidmodel()
    sampled, _ = run_dream(params, like, nchains=5, niterations=20, verbose=False, history_thin=1, model_name="t_hist",
                           save_history=True, adapt_crossover=False, seed=3)
    hist = np.load("t_hist_DREAM_chain_history.npy").reshape(-1, 4)
    nseed = 40
    assert len(hist) == nseed + 5 * 20
    appended = hist[nseed:].reshape(20, 5, 4)
    np.testing.assert_array_equal(appended, np.stack(sampled, axis=1))
    assert os.path.exists("t_hist_DREAM_chain_adapted_crossoverprob.npy") and os.path.exists("t_hist_DREAM_chain_adapted_gammalevelprob.npy")
    # restart from the files (core.py:46-62, 255-263)
    s2, _ = run_dream(params, like, nchains=5, niterations=10, verbose=False, restart=True, model_name="t_hist",
                      start=[s[-1] for s in sampled], save_history=False, seed=4)
    assert s2[0].shape == (10, 4)


def test_hard_boundaries_never_violated():
    """test_dream.py:670-705: uniform prior on [-5,10]x[-9,2]x[5,7]x[3,8], 1000 iterations, 5 chains."""
    params, like = multidmodel_uniform()
    lower = np.array([-5, -9, 5, 3]); upper = np.array([10, 2, 7, 8])
    sampled, _ = run_dream(params, like, nchains=5, niterations=1000, verbose=False, save_history=False, hardboundaries=True,
                           multitry=3, seed=5, lamb=0.5)
    S = np.concatenate(sampled)
    assert np.all(S >= lower) and np.all(S <= upper)
    assert len(np.unique(S[:, 0])) > 50           # the chains do move


def test_open_uniform_prior_without_hard_boundaries_stays_inside_its_support():
    """The same uniform-prior model with hardboundaries=False and multitry: a proposal outside the support has log prior -inf, a set
    of such proposals is generated again (Dream.py:281-289) instead of costing the step, and no sample ever leaves the support."""
    params, like = multidmodel_uniform()
    lower = np.array([-5, -9, 5, 3]); upper = np.array([10, 2, 7, 8])
    sampled, log_ps = run_dream(params, like, nchains=5, niterations=600, verbose=False, save_history=False, hardboundaries=False,
                                multitry=3, seed=6, lamb=0.5)
    S = np.concatenate(sampled)
    assert np.all(S >= lower) and np.all(S <= upper) and np.all(np.isfinite(np.concatenate(log_ps)))
    assert len(np.unique(S[:, 0])) > 50


def test_host_likelihood_equals_device_likelihood():
    """The same model through the host callback (arbitrary Python likelihood) and through the device
    descriptor gives the same chain decisions; logp agree to 1e-10."""
    d, N, n = 10, 6, 40
    P = H.mvn_precision(d)
    dev = MVNormalLogLike(P, factorize=False)
    host = lambda x: -.5 * np.sum(x * np.dot(P, x))          # dream_ex_ndim_gaussian.py:49-52
    hist = "/tmp/_dz_seed_%d.npy" % os.getpid()
    np.save(hist, H.seed_history(100, d, 8))
    kw = dict(nchains=N, niterations=n, verbose=False, save_history=False, history_file=hist, multitry=5, seed=6,
              start=[H.seed_history(100, d, 8)[i] for i in range(N)])
    s_dev, l_dev = run_dream([FlatParam(np.zeros(d))], dev, **kw)
    s_host, l_host = run_dream([FlatParam(np.zeros(d))], host, **kw)
    os.remove(hist)
    np.testing.assert_allclose(np.array(l_dev), np.array(l_host), rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.array(s_dev), np.array(s_host), rtol=1e-12)


def test_converges_on_the_reference_example_target(tmp_path):
    """C1 plumbing config (examples/ndim_gaussian at d=10, 3 chains, multitry 5): R-hat < 1.2 on every
    dimension (the example's own criterion, dream_ex_ndim_gaussian.py:80) and Var(x_i) ~ i."""
    d = 10
    hist = tmp_path / "seed.npy"
    Z0 = H.seed_history(1000, d, 9)
    np.save(hist, Z0)
    sampled, _ = run_dream([FlatParam(np.zeros(d))], MVNormalLogLike(H.mvn_precision(d)), nchains=3, niterations=6000, verbose=False,
                           start=[Z0[i] for i in range(3)], start_random=False, save_history=False, history_file=str(hist),
                           multitry=5, seed=10)
    assert np.all(Gelman_Rubin(sampled) < 1.2)
    S = np.concatenate([s[3000:] for s in sampled])
    np.testing.assert_allclose(S.var(axis=0), np.arange(1, d + 1), rtol=0.35)
    assert np.all(np.abs(S.mean(axis=0)) < 0.6)


def test_samples_leaving_while_the_run_continues_equal_one_download_at_the_end(monkeypatch):
    """Large quiet runs hand their samples to the host segment by segment, on a copy stream, while the next segment's generations
    run (dz_trace_download_begin / _wait): the same arrays as one download after the run, for segment counts that do and do not
    divide the run."""
    N, d, n = 4096, 100, 90                          # 295 MB of samples: above the threshold of the segmented path
    P = H.mvn_precision(d)
    hist = "/tmp/_dz_seed_dl%d.npy" % os.getpid()
    Z0 = H.seed_history(2 * N, d, 8)
    np.save(hist, Z0)
    kw = dict(nchains=N, niterations=n, verbose=False, save_history=False, history_file=hist, multitry=5, seed=31,
              start=[Z0[i] for i in range(N)], adapt_crossover=False)
    out = {}
    for segs in ("0", "4", "7"):
        monkeypatch.setenv("DREAMZS_DOWNLOAD_SEGMENTS", segs)
        out[segs] = run_dream([FlatParam(np.zeros(d))], MVNormalLogLike(P), **kw)
    os.remove(hist)
    for segs in ("4", "7"):
        np.testing.assert_array_equal(np.array(out["0"][0]), np.array(out[segs][0]))
        np.testing.assert_array_equal(np.array(out["0"][1]), np.array(out[segs][1]))
    assert len(np.unique(np.array(out["0"][0])[:, -1, 0])) > N // 2


def test_trace_by_chain_equals_trace_by_generation():
    """dz_get_trace_chains (the layout run_dream returns, core.py:98/:127) against dz_get_trace: whole buffer (single
    strided copy) and a window written into the middle of a larger destination (per-chain copies)."""
    from pydream_amd import _capi
    N, d, G = 48, 10, 30
    e = _capi.Engine(nchains=N, ndim=d, multitry=3, history_capacity=400 + N * 6, trace_capacity=G, seed=3)
    Z = np.random.default_rng(0).normal(size=(400, d))
    e.set_history(Z); e.set_state(Z[:N]); e.set_likelihood_mvn(np.zeros(d), np.eye(d), 0, 0.0)
    e.step(G)
    X = e.get_trace(0, G)["X"]                      # [G, N, d]
    S = np.full((N, G, d), np.nan)
    LP = np.full((N, G, 1), np.nan)
    e.get_trace_chains(0, G, S, logp_out=LP)
    np.testing.assert_array_equal(S, X.transpose(1, 0, 2))
    np.testing.assert_array_equal(LP[:, :, 0], e.get_trace(0, G, with_X=False)["logp"].T)
    S2 = np.full((N, 50, d), np.nan)
    L2 = np.full((N, 50), np.nan)
    e.get_trace_chains(5, 20, S2, row0=7, logp_out=L2)
    np.testing.assert_array_equal(S2[:, 7:27], X[5:25].transpose(1, 0, 2))
    np.testing.assert_array_equal(L2[:, 7:27], e.get_trace(5, 20, with_X=False)["logp"].T)
    assert np.isnan(L2[:, :7]).all() and np.isnan(L2[:, 27:]).all()
    assert np.isnan(S2[:, :7]).all() and np.isnan(S2[:, 27:]).all()
    e.close()


def test_run_dream_parallel_tempering_equals_reference(tmp_path):
    """test_dream.py:586-605 (return shapes) and more: run_dream(tempering=True) reproduces the arrays the reference's
    own _sample_dream_pt produced for the same seed (tests/golden/trace_pt_mvn10.npz, made by make_golden.py)."""
    from tests import helpers as H
    from pydream_amd.likelihoods import MVNormalLogLike
    from pydream_amd.parameters import FlatParam
    fx = H.load("trace_pt_mvn10")
    d, N, n = int(fx["cfg_d"]), int(fx["cfg_N"]), int(fx["cfg_G"])
    os.chdir(tmp_path)
    np.save("seed.npy", fx["Z0"])
    like = MVNormalLogLike(fx["invC"], log_F=float(fx["log_F"]), factorize=False)
    sampled, log_ps = run_dream([FlatParam(test_value=np.zeros(d))], like, nchains=N, niterations=n, tempering=True, verbose=False,
                                start=[fx["starts"][i] for i in range(N)], start_random=False, history_file="seed.npy",
                                multitry=int(fx["cfg_k"]), adapt_crossover=False, save_history=False, seed=int(fx["cfg_seed"]))
    assert sampled.shape == (N, 2 * n, d) and log_ps.shape == (N, 2 * n, 1)
    np.testing.assert_allclose(sampled, fx["pt_sampled"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(log_ps, fx["pt_log_ps"], rtol=0, atol=1e-10)


def test_astep_with_temperature_equals_oracle(tmp_path):
    """Dream.astep(q0, T, last_loglike, last_logprior) the way core._sample_dream_pt_chain calls it (core.py:238-248):
    round-robin over the chains, each at its own temperature, against the oracle's schedule S1 with the same ladder."""
    import copy
    from oracle import oracle as O
    fx = H.load("trace_s1_c1")
    d, N, G = int(fx["cfg_d"]), int(fx["cfg_N"]), 40
    T = np.array([np.power(.001, float(i) / N) for i in range(N)])
    hist = tmp_path / "seed.npy"
    np.save(hist, fx["Z0"])
    like = MVNormalLogLike(fx["invC"], log_F=float(fx["log_F"]), factorize=False)
    step = Dream(model=Model(like, [FlatParam(np.zeros(d))]), history_file=str(hist), start_random=False, save_history=False,
                 multitry=5, adapt_crossover=False, crossover_burnin=10 ** 9)
    pool = _setup_mp_dream_pool(N, G, step, start_pt=[fx["starts"][i] for i in range(N)], seed=int(fx["cfg_seed"]))
    pool._initializer(*pool._initargs)
    try:
        chains = [copy.copy(step) for _ in range(N)]
        for k in range(N):
            chains[k].chain_n = None
        x = {c: fx["starts"][c].copy() for c in range(N)}
        last = {c: (None, None) for c in range(N)}
        X = np.zeros((G, N, d))
        for g in range(G):
            for c in range(N):
                if g == 0:
                    Dream_shared_vars.nchains_counter = c + 1
                q, pr, lk = chains[c].astep(x[c], T[c], last[c][0], last[c][1])
                x[c] = q; X[g, c] = q; last[c] = (lk, pr)
        o = H.engine_from_trace_fixture(O.Engine, fx, trace_capacity=G, history_capacity=len(fx["Z0"]) + N * (G // 10 + 2))      # schedule S1
        o.set_temperatures(T, swaps=False)
        o.step(G)
        np.testing.assert_array_equal(X, o.get_trace(0, G)["X"])
        assert not np.array_equal(X, fx["X"][:G])          # the temperatures did change the walk
    finally:
        pool.close(); pool.join()


def _slow_like(x):
    import time
    time.sleep(0.002)
    return -.5 * float(np.sum(x * x))


def test_python_likelihood_over_worker_processes():
    """An expensive Python likelihood is spread over worker processes (the reference gets that from one process per
    chain, core.py:250-314): same samples as the in-process evaluation, and faster."""
    import time
    d, N, n = 6, 8, 30              # (long enough that starting the worker processes -- up to two seconds on a cold box -- does not decide the comparison)
    hist = "/tmp/_dz_seed_w%d.npy" % os.getpid()
    np.save(hist, H.seed_history(80, d, 5))
    kw = dict(nchains=N, niterations=n, verbose=False, save_history=False, history_file=hist, multitry=5, seed=9,
              start=[H.seed_history(80, d, 5)[i] for i in range(N)])
    out = {}
    for workers in ("1", "8"):
        os.environ["DREAMZS_HOST_WORKERS"] = workers
        t0 = time.perf_counter()
        out[workers] = run_dream([FlatParam(np.zeros(d))], _slow_like, **kw)
        out[workers + "t"] = time.perf_counter() - t0
    del os.environ["DREAMZS_HOST_WORKERS"]
    os.remove(hist)
    np.testing.assert_array_equal(np.array(out["1"][0]), np.array(out["8"][0]))
    np.testing.assert_array_equal(np.array(out["1"][1]), np.array(out["8"][1]))
    assert out["8t"] < 0.75 * out["1t"], (out["1t"], out["8t"])


def _oracle_run_dream(params, like, nchains, niterations, start, seed, **kwargs):
    """run_dream's own sequence (core.py) with the CPU oracle as the engine: the checker for API-level runs."""
    from oracle import oracle as O
    from pydream_amd.core import _sample_dream_batched
    restart = kwargs.pop("restart", False)
    if restart:
        mn = kwargs["model_name"]
        kwargs.update(history_file=mn + '_DREAM_chain_history.npy', crossover_file=mn + '_DREAM_chain_adapted_crossoverprob.npy',
                      gamma_file=mn + '_DREAM_chain_adapted_gammalevelprob.npy')
    step = Dream(model=Model(like, params), variables=params, verbose=False, **kwargs)
    pool = _setup_mp_dream_pool(nchains, niterations, step, start_pt=start, seed=seed, engine_cls=O.Engine)
    try:
        pool._initializer(*pool._initargs)
        step.save_history = False
        return _sample_dream_batched(pool.engine, step, niterations, False, 10)
    finally:
        pool.close(); pool.join()


# ---- the shipped Robertson example without PySB (examples/robertson_nopysb/example_sample_robertson_nopysb_with_dream.py:58-118):
# a stiff three-species ODE solved by scipy inside a Python likelihood, log10 rate constants under a uniform prior six decades wide,
# "-inf when the integrator fails".  The experimental data file is replaced by data simulated from the example's own nominal constants.
_ROB_T = np.linspace(0.0, 40.0, 12)
_ROB_TRUE = np.log10([.04, 3.0e7, 1.0e4])


def _rob_rhs(y, t, p1, p2, p3):
    return [-p1 * y[0] + p3 * y[1] * y[2], p1 * y[0] - p3 * y[1] * y[2] - p2 * y[1] ** 2, p2 * y[1] ** 2]


def _rob_ctot(logk):
    from scipy.integrate import odeint
    return odeint(_rob_rhs, [1.0, 0.0, 0.0], _ROB_T, args=tuple(10 ** np.asarray(logk, dtype=float)))[:, 2]


_ROB_DATA = None


def _rob_like(logk):
    global _ROB_DATA
    if _ROB_DATA is None:
        _ROB_DATA = _rob_ctot(_ROB_TRUE)
    if logk[2] > _ROB_TRUE[2] + 2.0:             # stands in for "simulation failed due to integrator errors" (:90-92)
        return -np.inf
    r = (_rob_ctot(logk) - _ROB_DATA) / (0.05 * _ROB_DATA + 1e-3)
    lp = -0.5 * float(np.sum(r * r))
    return lp if np.isfinite(lp) else -np.inf


@pytest.mark.parametrize("multitry,hard", [(False, True), (3, False)])
def test_robertson_example_with_a_python_ode_likelihood(tmp_path, multitry, hard):
    """The example's call (uniform SampledParam, gamma_levels=4, adapt_gamma=True, history_thin=1; run_dream :105-118) on the GPU engine
    with the likelihood behind the host callback equals the same sequence on the oracle bit for bit; as shipped (multitry off, hard
    boundaries) and with multitry 3 and open boundaries, where tries outside the prior's support or in the region the likelihood
    calls failed are impossible and whole sets are drawn again (Dream.py:281-289)."""
    os.chdir(tmp_path)
    N, G = 5, 50
    lower = _ROB_TRUE - 3
    params = [SampledParam(uniform, loc=lower, scale=6)]
    rng = np.random.default_rng(77)
    Z0 = lower + 6 * rng.uniform(0, 1, (40, 3))
    np.save("rob_seed.npy", Z0)
    starts = [_ROB_TRUE + 0.3 * rng.uniform(-1, 1, 3) for _ in range(N)]
    kw = dict(multitry=multitry, gamma_levels=4, adapt_gamma=True, history_thin=1, hardboundaries=hard, history_file="rob_seed.npy")
    os.environ["DREAMZS_HOST_WORKERS"] = "1"
    try:
        sampled, log_ps = run_dream(params, _rob_like, nchains=N, niterations=G, verbose=False, start=starts, save_history=False, seed=55, **kw)
        o_s, o_l = _oracle_run_dream(params, _rob_like, N, G, starts, 55, **kw)
    finally:
        del os.environ["DREAMZS_HOST_WORKERS"]
    S = np.concatenate(sampled)
    assert np.all(S >= lower) and np.all(S <= lower + 6) and np.all(np.isfinite(np.concatenate(log_ps)))
    assert len(np.unique(S[:, 0])) > N
    np.testing.assert_array_equal(np.array(sampled), np.array(o_s))
    np.testing.assert_array_equal(np.array(log_ps), np.array(o_l))


def test_restart_continues_bit_for_bit_like_the_oracle(tmp_path):
    """restart=True (core.py:46-62, 255-263; Dream.py:128-141, 947-969): a first run saves its history and adapted crossover /
    gamma-level probabilities; the restarted run seeds its archive with the WHOLE saved history, loads the probabilities and
    starts where the first run stopped.  The restarted GPU run equals the oracle started from the same three .npy files, bit for
    bit, and really did read them (its archive starts with the saved rows, its first proposals come from the adapted
    probabilities: a run restarted from the files differs from one that ignores them)."""
    os.chdir(tmp_path)
    d, N, G1, G2 = 12, 8, 60, 45
    P = H.mvn_precision(d)
    like = MVNormalLogLike(P, factorize=False)
    params = [FlatParam(np.zeros(d))]
    Z0 = H.seed_history(10 * d, d, 21)
    np.save("seed.npy", Z0)
    kw = dict(multitry=5, adapt_crossover=True, crossover_burnin=30, adapt_gamma=True, gamma_levels=2, history_thin=5)
    s1, _ = run_dream(params, like, nchains=N, niterations=G1, verbose=False, start=[Z0[i] for i in range(N)], history_file="seed.npy",
                      model_name="rs", save_history=True, seed=101, **kw)
    hist = np.load("rs_DREAM_chain_history.npy").reshape(-1, d)
    crp = np.load("rs_DREAM_chain_adapted_crossoverprob.npy")
    assert len(hist) == len(Z0) + N * (G1 // 5) and not np.allclose(crp, 1 / 3.)
    np.testing.assert_array_equal(hist[:len(Z0)], Z0)
    starts = [s[-1] for s in s1]
    # the restarted run: GPU through run_dream, oracle through the same host code
    s2, l2 = run_dream(params, like, nchains=N, niterations=G2, verbose=False, restart=True, start=starts, model_name="rs",
                       save_history=False, seed=202, **kw)
    o2, ol2 = _oracle_run_dream(params, like, N, G2, starts, 202, restart=True, model_name="rs", save_history=False, **kw)
    np.testing.assert_array_equal(np.array(s2), np.array(o2))
    np.testing.assert_array_equal(np.array(l2), np.array(ol2))
    # ... and the files mattered: the same run from the original seed history with uniform probabilities goes elsewhere
    s3, _ = run_dream(params, like, nchains=N, niterations=G2, verbose=False, start=starts, history_file="seed.npy",
                      save_history=False, seed=202, **kw)
    assert not np.array_equal(np.array(s2), np.array(s3))
    # saving again after the restart appends to the loaded history (Dream.py:947-959)
    run_dream(params, like, nchains=N, niterations=10, verbose=False, restart=True, start=starts, model_name="rs", save_history=True,
              seed=303, **kw)
    hist2 = np.load("rs_DREAM_chain_history.npy").reshape(-1, d)
    assert len(hist2) == len(hist) + N * 2
    np.testing.assert_array_equal(hist2[:len(hist)], hist)


def test_the_convergence_loop_of_the_examples_continues_on_the_live_engine(tmp_path, monkeypatch):
    """dream_ex_ndim_gaussian.py:79-102 as a user runs it: run_dream, Gelman_Rubin, then run_dream(restart=True, model_name=...) again
    and again.  The restart continues on the engine of the run before (history and adapted probabilities still in HBM: no engine is
    built, nothing is read from the .npy files or uploaded) and gives, bit for bit, what the same calls give when every restart
    rebuilds the engine from the files (DREAMZS_KEEP_ENGINE=0); the files are written either way; files touched in between, or another
    sampler, and the restart goes back to the files.  The result carries the device-made diagnostic of the run."""
    from pydream_amd import _capi, core
    from pydream_amd.convergence import Gelman_Rubin_device
    os.chdir(tmp_path)
    d, N, G = 12, 16, 60
    like = MVNormalLogLike(H.mvn_precision(d), factorize=False)
    params = [FlatParam(np.zeros(d))]
    Z0 = H.seed_history(10 * d, d, 5)
    np.save("seed.npy", Z0)
    kw = dict(multitry=5, adapt_crossover=True, history_thin=5, verbose=False, save_history=True)
    built = []
    real = _capi.Engine.__init__
    monkeypatch.setattr(_capi.Engine, "__init__", lambda self, **k: (built.append(1), real(self, **k))[1])

    def loop(name, keep):
        monkeypatch.setenv("DREAMZS_KEEP_ENGINE", "1" if keep else "0")
        out = []
        s, l = run_dream(params, like, nchains=N, niterations=G, start=[Z0[i] for i in range(N)], history_file="seed.npy", model_name=name, seed=11, **kw)
        out.append((np.array(s), np.array(l), s.gelman_rubin))
        for rnd in range(3):
            starts = [x[-1, :] for x in s]
            s, l = run_dream(params, like, nchains=N, niterations=G + 20 * rnd, start=starts, restart=True, model_name=name, seed=12 + rnd, **kw)
            out.append((np.array(s), np.array(l), s.gelman_rubin))
        core.release_engines()
        return out
    built.clear()
    a = loop("live", True)
    assert len(built) == 1                                     # one engine for the four calls ...
    built.clear()
    b = loop("files", False)
    assert len(built) == 4                                     # ... against one per call
    for (sa, la, ra), (sb, lb, rb) in zip(a, b):
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(la, lb)
        np.testing.assert_array_equal(ra, rb)
        np.testing.assert_allclose(ra, Gelman_Rubin(list(sa)), rtol=1e-10)          # the device diagnostic is the reference's
    for tail in ("DREAM_chain_history.npy", "DREAM_chain_adapted_crossoverprob.npy", "DREAM_chain_adapted_gammalevelprob.npy"):
        np.testing.assert_array_equal(np.load("live_" + tail), np.load("files_" + tail))
    assert len(np.load("live_DREAM_chain_history.npy")) == (len(Z0) + N * (12 + 12 + 16 + 20)) * d
    # a history file the user has replaced is read again; so is everything when the sampler differs
    monkeypatch.setenv("DREAMZS_KEEP_ENGINE", "1")
    s, _ = run_dream(params, like, nchains=N, niterations=G, start=[Z0[i] for i in range(N)], history_file="seed.npy", model_name="t", seed=11, **kw)
    assert "t" in core._parked
    hist = np.load("t_DREAM_chain_history.npy")
    np.save("t_DREAM_chain_history.npy", hist[:len(Z0) * d + N * d * 4])
    built.clear()
    s2, _ = run_dream(params, like, nchains=N, niterations=30, start=[x[-1] for x in s], restart=True, model_name="t", seed=5, **kw)
    assert len(built) == 1 and len(np.load("t_DREAM_chain_history.npy")) == (len(Z0) + N * 4 + N * 6) * d
    built.clear()
    run_dream(params, like, nchains=N, niterations=30, start=[x[-1] for x in s2], restart=True, model_name="t", seed=6, **dict(kw, snooker=0.3))
    assert len(built) == 1
    core.release_engines()
    assert Gelman_Rubin_device(s2) is not s2.gelman_rubin and np.array_equal(Gelman_Rubin_device(s2), s2.gelman_rubin)


@pytest.mark.parametrize("prior,finite", [("flat", True), ("flat", False), ("uniform_open", False), ("normal", True)])
def test_a_user_device_kernel_equals_the_same_function_through_the_host_callback(prior, finite):
    """dz_set_likelihood_module (the batched device callback for ANY model; the reference takes any callable, pydream/model.py:17-32): the
    banana density as a user-built HIP kernel on the HIP engine against the same function, operation for operation, through the host
    callback of the ORACLE -- states, log densities, every decision and the archive equal bit for bit; with a uniform prior narrower than
    the archive and no hard boundaries whole proposal sets are impossible and are drawn again (Dream.py:281-289) through the user kernel."""
    from oracle import oracle as O
    from pydream_amd import _capi as G
    from pydream_amd.examples.banana import banana_device as B
    d, N, k, n, seed = 12, 256, 5, 35, 5
    like = B.make_likelihood(d)
    Z0 = np.random.default_rng(2).uniform(-8, 8, (10 * d + 2 * N, d))
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1, crossover_burnin=15, hardboundaries=0 if prior == "uniform_open" else 1)
        if prior == "uniform_open":
            e.set_prior(np.full(d, 2, np.int32), np.full(d, -4.0), np.full(d, 8.0))
        elif prior == "normal":
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 1.0, d), np.full(d, 20.0))
        e.set_history(Z0)
        e.set_state(np.clip(Z0[:N], -3.9, 3.9) if prior == "uniform_open" else Z0[:N])
        if Cls is G.Engine:
            e.set_likelihood_module(like.code_object(), like.name, 1, like.data, always_finite=finite)
        else:
            e.set_likelihood_host(B.banana_host_batch)
        e.step(n)
        out.append((e.get_trace(0, n), e.get_history(), e.get_cr_state(), e.redraw_rounds() if Cls is G.Engine else None, e.last_kernel_variant() if Cls is G.Engine else None))
    for key in ("snooker", "cr_idx", "try_idx", "moved", "X", "logp"):
        np.testing.assert_array_equal(out[0][0][key], out[1][0][key], err_msg=key)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        np.testing.assert_array_equal(a, b)
    assert out[0][4] == "multi-kernel path" and 0.02 < out[0][0]["moved"].mean() < 0.95
    assert (out[0][3] > 0) == (prior == "uniform_open")


def test_run_dream_with_a_user_device_kernel_and_a_wave_per_point(tmp_path):
    """The drop-in path: run_dream(parameters, DeviceKernelLogLike(...)) -- compiled on first use, SampledParam priors added on the device --
    gives what run_dream with the Python twin of the kernel as a plain host likelihood gives (same seed: same samples, bit for bit); and
    the wave-per-point launch shape (lanes_per_point = 64) with a data block, against its own host twin."""
    from scipy.stats import norm
    from pydream_amd.examples.banana import banana_device as B
    from pydream_amd.likelihoods import DeviceKernelLogLike
    d, N, n = 6, 8, 30
    hist = str(tmp_path / "seed.npy")
    np.save(hist, np.random.default_rng(4).uniform(-6, 6, (80, d)))
    kw = dict(nchains=N, niterations=n, verbose=False, save_history=False, history_file=hist, multitry=3, seed=11, start=[np.full(d, 0.1 * i) for i in range(N)])
    pri = [SampledParam(norm, loc=np.zeros(d), scale=np.full(d, 10.0))]
    s_dev, l_dev = run_dream(pri, B.make_likelihood(d), **kw)
    s_host, l_host = run_dream(pri, lambda x: B.banana_host(x), **kw)
    np.testing.assert_array_equal(np.array(s_dev), np.array(s_host))
    np.testing.assert_allclose(np.array(l_dev), np.array(l_host), rtol=0, atol=1e-10)      # (the host path adds scipy's prior, the device its own: 1e-10)
    # one wave per point: each lane sums its strided share of the squared distances to data[0..d), then a butterfly over the lanes
    src = r'''
    extern "C" __global__ void sq_dist(const double* X, long long n, int d, int ld, double* like, const void* data)
    {
        const long long i = blockIdx.x * 4ll + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (i >= n) return;
        const double* c = (const double*)data;
        double acc = 0.0;
        for (int j = lane; j < d; j += 64) { const double t = X[i * ld + j] - c[j]; acc = acc + t * t; }
        for (int o = 32; o > 0; o >>= 1) acc = acc + __shfl_xor(acc, o, 64);
        if (lane == 0) like[i] = -0.5 * acc;
    }'''
    d2 = 150
    c = np.linspace(-1, 1, d2)
    like = DeviceKernelLogLike("sq_dist", d2, source=src, data=c, lanes_per_point=64)
    X = np.random.default_rng(0).normal(size=(37, d2))

    def twin(x):          # the same order of additions: lane l adds j = l, l + 64, l + 128; then the xor butterfly 32, 16, .. 1
        part = np.zeros(64)
        for j in range(d2):
            t = x[j] - c[j]; part[j % 64] = part[j % 64] + t * t
        for o in (32, 16, 8, 4, 2, 1):
            part = part + part[np.arange(64) ^ o]
        return -0.5 * part[0]
    got = np.array([like(x) for x in X[:5]])
    np.testing.assert_array_equal(got, np.array([twin(x) for x in X[:5]]))
    from pydream_amd import _capi as G
    e = G.Engine(nchains=3, ndim=d2, history_capacity=8)
    like._dz_apply(e)
    np.testing.assert_array_equal(e.eval_logp(X)[1], np.array([twin(x) for x in X]))
    with pytest.raises(G.DreamZSError, match="hipModuleGetFunction"):
        e.set_likelihood_module(like.code_object(), "no_such_kernel")


@pytest.mark.parametrize("multitry", [5, False])
def test_run_dream_at_the_reference_examples_dimension(tmp_path, monkeypatch, multitry):
    """The shipped n-dimensional Gaussian example's own d = 200 (dream_ex_ndim_gaussian.py:29, multitry = 5 at :65; and the reference's
    default, multitry off) through run_dream with the device likelihood: the generations run k_generations_d2 (the matrix from L2, two
    128-dimension chunks per lane; forced here at 48 chains, the default from 1025 on) and the samples equal run_dream's own sequence on
    the ORACLE, bit for bit, with crossover adaptation on (the reference's default)."""
    from scipy.stats import norm
    monkeypatch.setenv("DZ_MEGA_D2", "2")
    d, N, n = 200, 48, 40
    P = H.mvn_precision(d)
    hist = str(tmp_path / "seed.npy")
    Z0 = H.seed_history(2000, d, 21)
    np.save(hist, Z0)
    pri = [SampledParam(norm, loc=np.zeros(d), scale=np.full(d, 40.0))]
    kw = dict(history_file=hist, multitry=multitry, save_history=False, crossover_burnin=15)
    like = MVNormalLogLike(P)
    sampled, log_ps = run_dream(pri, like, nchains=N, niterations=n, start=[Z0[c] for c in range(N)], verbose=False, seed=4, **kw)
    from pydream_amd import core
    assert core.last_kernel_variant == ("k_generations_d2<13,tri,xhbm,16,1,full>" if multitry else "k_generations_d2<13,tri,xhbm,16,1,full,k1>")
    s_o, l_o = _oracle_run_dream(pri, like, N, n, [Z0[c] for c in range(N)], 4, **kw)
    np.testing.assert_array_equal(np.array(sampled), np.array(s_o))
    np.testing.assert_array_equal(np.array(log_ps), np.array(l_o))
    acc = np.mean([np.any(np.diff(np.asarray(s), axis=0) != 0, axis=1).mean() for s in sampled])
    assert 0.01 < acc < 0.95


USER_FN_SRC = r'''
__device__ double weighted_sq(const double* x, int d, const void* data, int lane)
{   // lane l adds its dimensions l, l + 64, ...; then the xor butterfly over the lanes (dz_wave_sum): -1/2 sum_j w_j (x_j - c_j)^2 + a quartic term
    const double* c = (const double*)data; const double* w = c + d;
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) { const double t = x[j] - c[j]; acc = acc + w[j] * (t * t) + 0.001 * ((t * t) * (t * t)); }
    return -0.5 * dz_wave_sum(acc);
}'''


def _user_fn_twin(c, w):
    def batch(X):          # the same operations in the same order, for a batch of points
        X = np.asarray(X, dtype=float).reshape(-1, len(c))
        part = np.zeros((len(X), 64))
        for j0 in range(0, len(c), 64):          # lane l's dimensions in ascending order: strips of 64 dimensions, all lanes of a strip at once
            m = min(64, len(c) - j0)
            t = X[:, j0:j0 + m] - c[j0:j0 + m]
            part[:, :m] = part[:, :m] + w[j0:j0 + m] * (t * t) + 0.001 * ((t * t) * (t * t))
        for o in (32, 16, 8, 4, 2, 1):
            part = part + part[:, np.arange(64) ^ o]
        return np.zeros(len(X)), -0.5 * part[:, 0]
    return batch


@pytest.mark.parametrize("N,k,prior,lag,persistent,variant", [
    (-600, 5, "flat", 0, "1", "k_generations_user<wide>"),                # (N < 0: at 200 dimensions) a lane owns four dimensions of the chain's state
    (-512, 4, "normal", 2, "1", "k_generations_user<full,wide> +ring"),
    (1024, 5, "flat", 0, "1", "k_generations_user"),                      # the lean instantiation; the burn-in's unit sums by k_adapt_partials (blocks of 4 waves) ...
    (4096, 5, "flat", 0, "1", "k_generations_user"),                      # ... or by the kernel's blocks of 16 themselves
    (1000, 4, "normal", 3, "1", "k_generations_user<full> +ring"),        # priors: the full proposal code; adapt_lag: four burn-in generations per launch
    (512, 1, "bounds", 2, "1", "k_generations_user<full> +ring"),         # multitry off, hard boundaries
    (1024, 5, "flat", 0, "0", "multi-kernel path"),                       # DZ_MEGA_USER=0: the same function through the batch kernel
])
def test_a_user_device_function_inside_the_persistent_kernel(N, k, prior, lag, persistent, variant, monkeypatch, tmp_path):
    """DeviceFunctionLogLike: a user's wave-level HIP device function compiled -- at run time, against the engine's own headers -- into the
    persistent generation kernel the built-in mixture runs in (csrc/dz_megakernel.h generations_wave_body) and into a batch kernel for the
    multi-kernel path.  Against the ORACLE with the same function, operation for operation, as a numpy host callback: states, log densities,
    every decision, the archive and the adapted crossover probabilities equal bit for bit, whichever of the two carries the generations."""
    from oracle import oracle as O
    from pydream_amd import _capi as G
    from pydream_amd.likelihoods import DeviceFunctionLogLike
    monkeypatch.setenv("DZ_MEGA_USER", persistent)
    monkeypatch.setenv("DREAMZS_KERNEL_CACHE", str(tmp_path))
    d, n, seed = (200 if N < 0 else 100), 35, 9
    N = abs(N)
    c = np.linspace(-2.0, 2.0, d); w = 0.5 + np.arange(d) % 7 / 7.0
    like = DeviceFunctionLogLike(USER_FN_SRC, "weighted_sq", d, data=np.concatenate([c, w]), always_finite=True, host=None)
    twin = _user_fn_twin(c, w)
    Z0 = np.random.default_rng(2).uniform(-6, 6, (10 * d + 2 * N, d))
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=k, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=seed,
                adapt_crossover=1, crossover_burnin=15, adapt_lag=lag)
        if prior == "normal":
            e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 1.0, d), np.full(d, 20.0))
        elif prior == "bounds":
            e.set_bounds(np.full(d, -7.0), np.full(d, 7.0))
        e.set_history(Z0); e.set_state(Z0[:N])
        if Cls is G.Engine:
            like._dz_apply(e)
        else:
            e.set_likelihood_host(twin)
        variants = set()
        for m in (7, 13, 15):
            e.step(m)
            if Cls is G.Engine:
                variants.add(e.last_kernel_variant())
        out.append((e.get_trace(0, n), e.get_history(), e.get_cr_state(), variants))
    for key in ("snooker", "cr_idx", "try_idx", "moved", "X", "logp"):
        np.testing.assert_array_equal(out[0][0][key], out[1][0][key], err_msg=key)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        np.testing.assert_array_equal(a, b)
    assert variant in out[0][3], out[0][3]
    assert 0.02 < out[0][0]["moved"].mean() < 0.95 and not np.allclose(out[0][2][0], 1 / 3.)


def test_run_dream_with_a_user_device_function_equals_its_python_twin(tmp_path):
    """The drop-in path for a density that is not built in: run_dream(parameters, DeviceFunctionLogLike(...)) -- the function compiled into the
    persistent kernel on first use, the reference's defaults otherwise (crossover adaptation on, burn-in = a tenth of the run) -- returns what
    run_dream returns for the same function as a plain Python likelihood (the reference's own calling convention, model.py:31): the same
    samples bit for bit, log densities to 1e-10 (scipy's prior on the host, the engine's own on the device)."""
    from scipy.stats import norm
    from pydream_amd import core
    from pydream_amd.likelihoods import DeviceFunctionLogLike
    d, N, n = 20, 64, 60
    c = np.linspace(-1.0, 1.0, d); w = 0.5 + np.arange(d) % 7 / 7.0
    twin = _user_fn_twin(c, w)
    hist = str(tmp_path / "seed.npy")
    np.save(hist, np.random.default_rng(4).uniform(-4, 4, (10 * d + 2 * N, d)))
    kw = dict(nchains=N, niterations=n, verbose=False, save_history=False, history_file=hist, multitry=3, seed=5,
              start=[np.full(d, 0.05 * i) for i in range(N)])
    pri = [SampledParam(norm, loc=np.zeros(d), scale=np.full(d, 5.0))]
    like = DeviceFunctionLogLike(USER_FN_SRC, "weighted_sq", d, data=np.concatenate([c, w]), always_finite=True, host=lambda x: float(twin(x)[1][0]))
    s_dev, l_dev = run_dream(pri, like, **kw)
    variant = core.last_kernel_variant
    s_host, l_host = run_dream(pri, lambda x: float(twin(x)[1][0]), **kw)
    np.testing.assert_array_equal(np.array(s_dev), np.array(s_host))
    np.testing.assert_allclose(np.array(l_dev), np.array(l_host), rtol=0, atol=1e-10)
    assert variant.startswith("k_generations_user<full>"), variant          # (SampledParam priors: the full proposal code)
    assert like(np.zeros(d)) == twin(np.zeros(d))[1][0]                       # the host twin answers a host call


def test_the_banana_example_as_a_device_function():
    """pydream_amd/examples/banana: the same density as a wave-level device function (FUNCTION_SOURCE) -- the device evaluates it to the bits of its
    Python twin, and a run with it goes through the persistent kernel and equals the ORACLE's run with the twin as a host callback."""
    from oracle import oracle as O
    from pydream_amd import _capi as G
    from pydream_amd.examples.banana import banana_device as B
    d, N, n = 12, 256, 30
    like = B.make_function_likelihood(d)
    X = np.random.default_rng(1).normal(scale=4.0, size=(50, d))
    e = G.Engine(nchains=3, ndim=d, history_capacity=8)
    like._dz_apply(e)
    np.testing.assert_array_equal(e.eval_logp(X)[1], np.array([B.banana_host_wave(x) for x in X]))
    Z0 = np.random.default_rng(2).uniform(-8, 8, (10 * d + 2 * N, d))
    out = []
    for Cls in (G.Engine, O.Engine):
        e = Cls(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (n // 10 + 2), trace_capacity=n, seed=3, adapt_crossover=1, crossover_burnin=10)
        e.set_history(Z0); e.set_state(Z0[:N])
        if Cls is G.Engine:
            like._dz_apply(e)
        else:
            e.set_likelihood_host(lambda Xb: (np.zeros(len(Xb)), np.array([B.banana_host_wave(x) for x in Xb])))
        e.step(n)
        out.append((e.get_trace(0, n), e.get_history(), e.last_kernel_variant() if Cls is G.Engine else ""))
    for key in ("snooker", "cr_idx", "try_idx", "moved", "X", "logp"):
        np.testing.assert_array_equal(out[0][0][key], out[1][0][key], err_msg=key)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert out[0][2] == "k_generations_user"
