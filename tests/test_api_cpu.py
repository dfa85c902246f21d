"""Host-side mirror of the reference's Python API (no GPU needed): priors, Model, Gelman_Rubin, the
Dream constructor's option handling and run_dream's validation errors -- written after the
reference's own tests (pydream/tests/test_dream.py)."""
import os

import numpy as np
import pytest
from scipy.stats import norm, uniform, gamma as gamma_dist

from pydream_amd.Dream import Dream
from pydream_amd.convergence import Gelman_Rubin
from pydream_amd.core import run_dream
from pydream_amd.likelihoods import GaussianMixtureLogLike, MVNormalLogLike
from pydream_amd.model import Model
from pydream_amd.parameters import FlatParam, SampledParam
from tests import helpers as H


def simple_likelihood(param):      # pydream/tests/test_models.py:47-50
    return np.sum(param + 3)


def onedmodel():                   # test_models.py:14-22
    return [SampledParam(norm, loc=-2, scale=3)], simple_likelihood


def multidmodel():                 # test_models.py:24-33
    return [SampledParam(norm, loc=np.array([-6.6, 3, 1.0, -.12]), scale=np.array([.13, 5, .9, 1.0]))], simple_likelihood


def multidmodel_uniform():         # test_models.py:35-45
    lower = np.array([-5, -9, 5, 3]); upper = np.array([10, 2, 7, 8])
    return [SampledParam(uniform, loc=lower, scale=upper - lower)], simple_likelihood


def test_check_dimension_and_boundaries():
    """test_dream.py:30-50: dimensions 1 and 4; bounds from interval(1)."""
    p, l = onedmodel()
    assert Dream(model=Model(l, p)).total_var_dimension == 1
    p, l = multidmodel_uniform()
    step = Dream(model=Model(l, p))
    assert step.total_var_dimension == 4
    np.testing.assert_array_equal(step.mins, [-5, -9, 5, 3])
    np.testing.assert_array_equal(step.maxs, [10, 2, 7, 8])
    f = Dream(model=Model(l, [FlatParam(np.zeros(3))]))
    assert np.all(np.isinf(f.mins)) and np.all(np.isinf(f.maxs))


def test_gamma_array_known_values():
    """test_dream.py:68-76"""
    p, l = onedmodel()
    dream = Dream(model=Model(l, p), DEpairs=5, p_gamma_unity=0)
    np.testing.assert_allclose(dream.gamma_arr[0, :, 0], [1.683, 1.19, .972, .841, .753], atol=5e-4)
    fx = H.load("densities")
    np.testing.assert_array_equal(Dream(model=Model(l, [FlatParam(np.zeros(7))]), DEpairs=5, gamma_levels=4).gamma_arr, fx["gamma_arr_7_5_4"])


def test_option_handling():
    """multitry False/True/int (Dream.py:155-161), nCR clipped to d (:110-113), adapt off at d=1 (:115-118),
    nseedchains default 10 d (:168-170), unknown kwargs swallowed (:67), multitry=2 rejected."""
    p, l = multidmodel()
    m = Model(l, p)
    assert Dream(model=m).multitry == 1
    assert Dream(model=m, multitry=True).multitry == 5
    assert Dream(model=m, multitry=3).multitry == 3
    assert Dream(model=m, nCR=9).nCR == 4
    assert Dream(model=m).nseedchains == 40
    assert Dream(model=m, nverbose=3).total_var_dimension == 4
    p1, l1 = onedmodel()
    assert Dream(model=Model(l1, p1), adapt_crossover=True).adapt_crossover is False
    np.testing.assert_allclose(Dream(model=m).CR_values, [1 / 3., 2 / 3., 1.0])
    with pytest.raises(Exception, match="multitry=2"):
        Dream(model=m, multitry=2)


def test_limits_of_the_engine_are_said_at_construction():
    """The reference takes any integer for multitry, DEpairs, nCR, gamma_levels and any dimension (Dream.py:81-83, :108-122, :148-161); the
    GPU kernels have fixed lane budgets.  The user hears it from Dream(...), with the limit, not from dz_create in the middle of run_dream."""
    p, l = multidmodel()
    m = Model(l, p)
    for kw, what in ((dict(multitry=33), "multitry = 33"), (dict(DEpairs=9), "DEpairs = 9"), (dict(gamma_levels=40), "gamma_levels = 40")):
        with pytest.raises(Exception, match=what + ": this GPU engine supports at most"):
            Dream(model=m, **kw)
    assert Dream(model=m, multitry=32, DEpairs=8, gamma_levels=32).multitry == 32
    big = Model(simple_likelihood, [FlatParam(np.zeros(1025))])
    with pytest.raises(Exception, match="total dimension of all variables = 1025"):
        Dream(model=big)
    assert Dream(model=Model(simple_likelihood, [FlatParam(np.zeros(1024))]), nCR=32).nCR == 32
    with pytest.raises(Exception, match="nCR = 33"):
        Dream(model=Model(simple_likelihood, [FlatParam(np.zeros(64))]), nCR=33)


def test_adapt_lag_goes_through_run_dreams_own_sequence(tmp_path, monkeypatch):
    """`adapt_lag` (and `history_lag`) as run_dream keywords reach the engine's configuration: core._setup_mp_dream_pool ->
    _sample_dream_batched with the oracle as the engine gives the samples of an oracle engine configured by hand, and other samples
    than adapt_lag = 0."""
    from oracle import oracle as O
    from pydream_amd.core import _sample_dream_batched, _setup_mp_dream_pool
    monkeypatch.chdir(tmp_path)
    d, N, G = 6, 8, 80
    P = H.mvn_precision(d)
    like = MVNormalLogLike(P, factorize=False)
    Z0 = H.seed_history(40, d, 5)
    np.save("seed.npy", Z0)

    def run(lag):
        step = Dream(model=Model(like, [FlatParam(np.zeros(d))]), verbose=False, multitry=5, history_file="seed.npy", save_history=False, crossover_burnin=60)
        pool = _setup_mp_dream_pool(N, G, step, start_pt=list(Z0[:N]), seed=21, engine_cls=O.Engine, history_lag=1, adapt_lag=lag)
        try:
            pool._initializer(*pool._initargs)
            assert int(pool.engine.cfg.adapt_lag) == lag and int(pool.engine.cfg.history_lag) == 1
            s, _ = _sample_dream_batched(pool.engine, step, G, False, 10)
        finally:
            pool.close(); pool.join()
        return np.array(s)
    a, b = run(4), run(0)
    e = O.Engine(nchains=N, ndim=d, multitry=5, history_capacity=len(Z0) + N * (G // 10 + 2), trace_capacity=G, seed=21, adapt_crossover=1, crossover_burnin=60,
                 adapt_lag=4, history_lag=1)
    e.set_history(Z0); e.set_state(Z0[:N]); e.set_likelihood_mvn(np.zeros(d), P, 0, float(like.log_F) if hasattr(like, "log_F") else 0.0)
    e.step(G)
    np.testing.assert_array_equal(a, e.get_trace(0, G)["X"].transpose(1, 0, 2))
    assert not np.array_equal(a, b)


def test_priors_and_model():
    """parameters.py:37-47, 62-63; model.py:17-32"""
    p, l = multidmodel()
    q = np.array([-6.5, 2.0, 1.1, 0.3])
    m = Model(l, p)
    pr, lk = m.total_logp(q)
    assert pr == pytest.approx(np.sum(norm(loc=[-6.6, 3, 1.0, -.12], scale=[.13, 5, .9, 1.0]).logpdf(q)))
    assert lk == pytest.approx(np.sum(q + 3))
    kind, a, b = m.device_prior()
    np.testing.assert_array_equal(kind, [1, 1, 1, 1]); np.testing.assert_allclose(a, [-6.6, 3, 1.0, -.12]); np.testing.assert_allclose(b, [.13, 5, .9, 1.0])
    pu, _ = multidmodel_uniform()
    kind, a, b = Model(l, pu).device_prior()
    np.testing.assert_array_equal(kind, [2, 2, 2, 2]); np.testing.assert_allclose(a, [-5, -9, 5, 3]); np.testing.assert_allclose(b, [15, 11, 2, 5])
    assert Model(l, [SampledParam(gamma_dist, 2.0)]).device_prior() is None          # host prior for anything else
    f = FlatParam(np.zeros(5))
    assert f.prior(np.ones(5)) == 0 and f.dsize == 5
    pr_b, lk_b = Model(l, [f, SampledParam(gamma_dist, 2.0)]).batch_logp(np.ones((3, 6)), with_prior=True)
    assert pr_b[0] == pytest.approx(gamma_dist(2.0).logpdf(1.0)) and lk_b[0] == 24


def test_device_likelihood_objects_are_plain_callables():
    fx = H.load("densities")
    ll = MVNormalLogLike(fx["mvn10_invC"], log_F=float(fx["mvn10_logF"]))
    np.testing.assert_allclose([ll(x) for x in fx["mvn10_X"]], fx["mvn10_logp"], atol=1e-10)
    mix = GaussianMixtureLogLike(fx["mix2_mu"], fx["mix2_logF"])
    np.testing.assert_allclose([mix(x) for x in fx["mix2_X"]], fx["mix2_logp"], atol=1e-10)


def test_gelman_rubin_matches_reference():
    fx = H.load("densities")
    tr = fx["gr_traces"]
    np.testing.assert_allclose(Gelman_Rubin([tr[c] for c in range(len(tr))]), fx["gr_rhat"], rtol=1e-12)


def test_run_dream_validation_errors():
    """core.py:46-50, 252-254, 270-273 -- raised before any device work."""
    p, l = multidmodel()
    with pytest.raises(Exception, match="no start positions"):
        run_dream(p, l, restart=True)
    with pytest.raises(Exception, match="no model name"):
        run_dream(p, l, restart=True, start=[np.zeros(4)] * 5)
    with pytest.raises(Exception, match=r"at least \(2\*DEpairs\)\+1"):
        run_dream(p, l, nchains=2, niterations=10, verbose=False)
    with pytest.raises(Exception, match="seeded starting history is insufficient"):
        run_dream(p, l, nchains=30, niterations=10, verbose=False)          # default nseedchains = 40 < 2*30



def _quad_like(x):
    return -.5 * float(np.sum(x * x))


def test_host_evaluator_pool_equals_in_process_evaluation(monkeypatch):
    """model.HostEvaluator (worker processes behind the batched host callback; the reference's one process per chain,
    core.py:250-314): the split over workers returns the same (prior, likelihood) columns as Model.batch_logp, for a batch
    smaller than, equal to and larger than the number of workers, with and without the host-side prior."""
    from pydream_amd.model import HostEvaluator, Model
    from pydream_amd.parameters import SampledParam
    from scipy.stats import norm
    model = Model(_quad_like, [SampledParam(norm, loc=np.zeros(3), scale=2.0), FlatParam(np.zeros(2))])
    rng = np.random.default_rng(5)
    monkeypatch.setenv("DREAMZS_HOST_WORKERS", "3")
    for with_prior in (True, False):
        ev = HostEvaluator(model, with_prior, nchains=8, force=True)
        try:
            for n in (1, 2, 3, 11):
                X = rng.normal(size=(n, 5))
                p0, l0 = model.batch_logp(X, with_prior)
                p1, l1 = ev(X)
                np.testing.assert_array_equal(p0, p1)
                np.testing.assert_array_equal(l0, l1)
            assert ev.pool is not None                       # the pool really ran
        finally:
            ev.close()
    monkeypatch.setenv("DREAMZS_HOST_WORKERS", "0")
    ev = HostEvaluator(model, True, nchains=8, force=True)
    ev(rng.normal(size=(4, 5)))
    assert ev.pool is None                                   # disabled: in-process


def test_open_uniform_prior_run_on_the_oracle_stays_inside_its_support(tmp_path, monkeypatch):
    """run_dream's own sequence (core._setup_mp_dream_pool -> _sample_dream_batched) with the oracle as the engine: uniform prior,
    hardboundaries=False, multitry -- proposal sets that lie wholly outside the support are drawn again (Dream.py:281-289) and no
    sample leaves it.  (The GPU engine runs the same call in tests/test_api_gpu.py.)"""
    from oracle import oracle as O
    from pydream_amd.core import _sample_dream_batched, _setup_mp_dream_pool
    monkeypatch.chdir(tmp_path)
    params, like = multidmodel_uniform()
    lower = np.array([-5, -9, 5, 3]); upper = np.array([10, 2, 7, 8])
    rng = np.random.default_rng(4)
    np.save("seed.npy", lower + (3 * rng.uniform(0, 1, (40, 4)) - 1) * (upper - lower))        # an archive wider than the support
    starts = [lower + rng.uniform(0, 1, 4) * (upper - lower) for _ in range(5)]
    step = Dream(model=Model(like, params), variables=params, verbose=False, multitry=3, hardboundaries=False, history_file="seed.npy",
                 save_history=False)
    pool = _setup_mp_dream_pool(5, 300, step, start_pt=starts, seed=8, engine_cls=O.Engine)
    try:
        pool._initializer(*pool._initargs)
        sampled, log_ps = _sample_dream_batched(pool.engine, step, 300, False, 10)
    finally:
        pool.close(); pool.join()
    S = np.concatenate(sampled)
    assert np.all(S >= lower) and np.all(S <= upper) and np.all(np.isfinite(np.concatenate(log_ps)))
    assert len(np.unique(S[:, 0])) > 50



def test_gelman_rubin_on_run_dream_shaped_results_copies_nothing():
    """run_dream returns the chains of ONE [chain, iteration, d] array (core.py:98/:127's list of per-chain arrays as views): the
    diagnostic recognises that and works on the array in place, a block of chains at a time -- same numbers as chain by chain (the
    reference's own order, convergence.py:8-14), no stacked copy of the run."""
    import tracemalloc
    from pydream_amd.convergence import _common_base
    rng = np.random.default_rng(3)
    S = rng.normal(size=(256, 3001, 20)).cumsum(axis=1)            # 123 MB
    views = [S[c] for c in range(len(S))]
    assert _common_base(views) is S
    assert _common_base(views[::-1]) is None and _common_base([v.copy() for v in views]) is None and _common_base(views[:-1]) is None
    tracemalloc.start()
    r = Gelman_Rubin(views)
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert peak < 48 << 20                                         # (np.stack alone would be S.nbytes; the blocks' temporaries are ~32 MB)
    np.testing.assert_array_equal(r, Gelman_Rubin([v.copy() for v in views]))
    assert r.shape == (20,) and np.all(r > 1.0)


def test_a_user_kernel_compiles_to_a_gfx950_code_object_that_exports_it(tmp_path, monkeypatch):
    """pydream_amd.likelihoods.compile_device_kernel (hipcc cross-compiles without a GPU): the example's HIP twin of a Python likelihood
    becomes an ELF code object exporting the extern "C" kernel dz_set_likelihood_module looks up; the host twin is the reference-style
    callable f(x[d]) -> float (pydream/model.py:31)."""
    import subprocess
    from pydream_amd.examples.banana import banana_device as B
    from pydream_amd.likelihoods import DeviceKernelLogLike, compile_device_kernel
    out = compile_device_kernel(B.SOURCE, str(tmp_path / "banana.hsaco"))
    assert open(out, "rb").read(4) == b"\x7fELF"
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", out], capture_output=True, text=True).stdout
    assert "banana_logp" in syms and "banana_logp.kd" in syms
    like = B.make_likelihood(6)
    x = np.arange(6.0) - 2.0
    assert like(x) == B.banana_host(x) == B.banana_host_batch(x[None])[1][0]          # the host twin answers a host call
    with pytest.raises(ValueError):
        DeviceKernelLogLike("f", 3)                                                  # neither source nor path
    with pytest.raises(Exception, match="hipcc failed"):
        compile_device_kernel("this is not HIP", str(tmp_path / "bad.hsaco"))
    # without an output path the code object is cached by (source, flags, architecture): the second request compiles nothing
    monkeypatch.setenv("DREAMZS_KERNEL_CACHE", str(tmp_path / "cache"))
    import time
    t0 = time.perf_counter(); a = compile_device_kernel(B.SOURCE); t1 = time.perf_counter(); b = compile_device_kernel(B.SOURCE); t2 = time.perf_counter()
    assert a == b and os.path.dirname(a) == str(tmp_path / "cache") and (t2 - t1) < 0.2 * (t1 - t0)
    assert compile_device_kernel(B.SOURCE, extra_flags=("-ffp-contract=fast",)) != a
    assert sorted(f for f in os.listdir(tmp_path / "cache")) == sorted(os.path.basename(x) for x in (a, compile_device_kernel(B.SOURCE, extra_flags=("-ffp-contract=fast",))))


def test_a_user_device_function_compiles_into_the_persistent_kernels(tmp_path, monkeypatch):
    """pydream_amd.likelihoods.DeviceFunctionLogLike (hipcc cross-compiles without a GPU): a wave-level device function becomes ONE code object
    with the batch kernel of the multi-kernel path and the two persistent kernels (lean / full proposal code) whose names carry the layout
    generation the engine looks them up by (csrc/dz_kernels.h DZ_USER_ABI); a compile error is the user's to read; the host twin answers
    host calls."""
    import re
    import subprocess
    from pydream_amd.likelihoods import DeviceFunctionLogLike
    monkeypatch.setenv("DREAMZS_KERNEL_CACHE", str(tmp_path))
    src = '''
    __device__ double sq(const double* x, int d, const void* data, int lane)
    {
        double acc = 0.0;
        for (int j = lane; j < d; j += 64) acc = acc + x[j] * x[j];
        return -0.5 * dz_wave_sum(acc);
    }'''
    like = DeviceFunctionLogLike(src, "sq", 10, always_finite=True, host=lambda x: -0.5 * float(np.sum(x * x)))
    out = like.code_object()
    assert open(out, "rb").read(4) == b"\x7fELF" and like.code_object() == out
    abi = int(re.search(r"#define DZ_USER_ABI (\d+)", open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pydream_amd", "csrc", "dz_kernels.h")).read()).group(1))
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", out], capture_output=True, text=True).stdout
    for name in ("dz_user_batch", "dz_user_generations_v%d" % abi, "dz_user_generations_full_v%d" % abi):
        assert name + ".kd" in syms, name
    assert like(np.ones(10)) == -5.0
    assert DeviceFunctionLogLike(None, "sq", 10, path=out).code_object() == out          # a code object built beforehand
    with pytest.raises(ValueError):
        DeviceFunctionLogLike(None, "sq", 10)
    with pytest.raises(Exception, match="hipcc failed"):
        DeviceFunctionLogLike("__device__ double bad(const double* x, int d, const void* data, int lane) { return y; }", "bad", 3).code_object()
