"""The reference's own algorithm-component tests (pydream/tests/test_dream.py), restated against an engine
object.  The same functions are run on the CPU oracle (tests/test_oracle_reference_kat.py) and, through the
C ABI, on the HIP engine (tests/test_reference_suite_gpu.py): `make(**cfg)` builds either one.

Each check names the reference test it restates (file:line of /root/reference/pydream/tests/test_dream.py).
The reference draws from the global numpy stream and asserts frequencies to one decimal place; here the draws
come from the counter-based contract, so the same assertions are made on the engine's recorded decisions.
"""
import numpy as np


def _flat_engine(make, n=64, d=4, k=1, gens=0, Z=None, **kw):
    """N chains on a constant density (MVN with a zero precision matrix): every finite proposal is accepted."""
    cfg = dict(nchains=n, ndim=d, multitry=k, history_capacity=4096 + n * (gens + 2), trace_capacity=max(gens, 1), seed=1234,
               hardboundaries=0)
    cfg.update(kw)
    e = make(**cfg)
    if Z is None:
        Z = np.random.default_rng(5).normal(size=(max(10 * d, 2 * n * cfg.get("depairs", 1)), d))
    e.set_history(Z)
    e.set_state(Z[:n] if Z.shape[0] >= n else np.zeros((n, d)))
    e.set_likelihood_mvn(np.zeros(d), np.zeros((d, d)), 0, 0.0)
    return e, Z


def check_gamma_array(gamma_table):
    """test_gamma_array (:68-76): gamma_arr[0][delta-1][d'-1] for d'=1, delta=1..5 with 3 decimals."""
    t = gamma_table(1, 5, 1)
    np.testing.assert_allclose(t[0, :, 0], [1.683, 1.19, .972, .841, .753], atol=5e-4)
    t = gamma_table(2, 2, 3)                     # Dream.py:172-179: level l divides by 2^(l-1)
    np.testing.assert_allclose(t[1], t[0] / 2.0, rtol=0, atol=0)
    np.testing.assert_allclose(t[0, 1, 2], 2.38 / np.sqrt(2 * 2 * 3), rtol=1e-15)


def check_snooker_and_cr_fractions(make):
    """test_snooker_fraction (:86-98), test_CR_fraction (:100-123): 10000 decisions, one decimal place."""
    probs = np.array([.10, .65, .25])
    e, _ = _flat_engine(make, n=100, gens=100, ncr=3, snooker=0.1)
    e.set_cr_probs(probs)
    e.step(100)
    tr = e.get_trace(0, 100)
    assert abs(tr["snooker"].mean() - 0.1) < 0.05
    freq = np.bincount(tr["cr_idx"].ravel(), minlength=3) / tr["cr_idx"].size
    assert np.all(np.abs(freq - probs) < 0.05)
    e.close()
    e, _ = _flat_engine(make, n=100, gens=20, snooker=0.0)       # snooker == 0 never takes the snooker branch (Dream.py:542-554)
    e.step(20)
    assert e.get_trace(0, 20)["snooker"].sum() == 0
    e.close()


def check_gamma_choices(make):
    """test_gamma_unityfraction (:54-66): gamma == 1 with probability p_gamma_unity, else the table value;
    test_gamma_snooker_choice (:78-84): snooker gamma is uniform on [1.2, 2.2]."""
    d = 4
    Z2 = np.array([[0.0] * d, [8.0] * d])                 # two rows: |za - zb| = 8 in every dimension
    e, _ = _flat_engine(make, n=64, d=d, k=1, Z=Z2, p_gamma_unity=0.2, ncr=1)
    q0 = np.array([2., 3., 4., 5.])
    table = 2.38 / np.sqrt(2 * d)                          # d' = d because CR = 1 (Dream.py:172-179)
    n_unity = n = 0
    for c in range(64):
        for g in range(40):
            pts, _, _, _ = e.debug_propose(c, g, 0, q0, False, 0)
            r = np.abs(pts[0] - q0) / 8.0                  # = e * gamma, e in (0.95, 1.05)  (Dream.py:714)
            assert np.all((np.abs(r - 1.0) < 0.0501) | (np.abs(r / table - 1.0) < 0.0501))
            n_unity += bool(np.all(np.abs(r - 1.0) < 0.0501))
            n += 1
    assert abs(n_unity / n - 0.2) < 0.05
    e.close()
    Z1 = np.array([[0.0], [1.0]])                          # one dimension: zP = zR1 - zR2, so |jump| = gamma_s or 0
    e, _ = _flat_engine(make, n=64, d=1, k=1, Z=Z1)
    gs = []
    for c in range(64):
        for g in range(20):
            pts, slogp, _, _ = e.debug_propose(c, g, 0, np.array([5.0]), True, 2)
            j = abs(pts[0, 0] - 5.0)
            if j > 0:
                gs.append(j)
    gs = np.array(gs)
    assert len(gs) > 300 and gs.min() >= 1.2 and gs.max() <= 2.2 and abs(gs.mean() - 1.7) < 0.05
    assert abs(len(gs) / (64 * 20) - 0.5) < 0.06          # zR1 != zR2 half of the time (independent draws, :808-810)
    e.close()


def check_depair_selection(make):
    """test_DEpair_selec (:125-149): delta ~ U{1,2,3}.  Read back from the jump: with gamma = 1 (p_gamma_unity = 1),
    CR = 1 and N(0,1) history rows, |jump|^2 / d concentrates on 2*delta."""
    n, d, G = 64, 200, 40
    Z = np.random.default_rng(8).normal(size=(4000, d))
    e, _ = _flat_engine(make, n=n, d=d, k=1, gens=G, Z=Z, depairs=3, p_gamma_unity=1.0, ncr=1, snooker=0.0, history_thin=1000000)
    e.step(G)
    X = e.get_trace(0, G)["X"]
    prev = np.concatenate([Z[None, :n], X[:-1]], axis=0)
    m = ((X - prev) ** 2).sum(axis=2) / d                  # [G, n]
    assert np.all(e.get_trace(0, G)["moved"] == 1)         # constant density: every proposal is accepted
    cls = np.where(m < 3.0, 1, np.where(m < 5.0, 2, 3)).ravel()
    freq = np.bincount(cls, minlength=4)[1:] / cls.size
    assert np.all(np.abs(freq - 1 / 3.) < 0.05), freq
    e.close()
    e, _ = _flat_engine(make, n=n, d=d, k=1, gens=5, Z=Z, depairs=1, p_gamma_unity=1.0, ncr=1, snooker=0.0, history_thin=1000000)
    e.step(5)                                              # set_DEpair with one choice returns 1 (:135)
    X = e.get_trace(0, 5)["X"]
    prev = np.concatenate([Z[None, :n], X[:-1]], axis=0)
    # (the 64 rows appended at iteration 0 are moved states with three times the variance: a few percent of the pairs)
    assert np.mean(((X - prev) ** 2).sum(axis=2) / d < 3.0) > 0.93
    e.close()


def check_crossover_fraction_of_dims(make):
    """test_proposal_generation_nosnooker_CR1/CR33/CR66 (:202-305): with the history 0..119 (20 rows of 6? no: 30 rows
    of 4), q0 = [2,3,4,5], the fraction of dimensions left unchanged is 1-CR on average, for 1 and for 5 proposals."""
    d = 4
    for k in (1, 5):
        e = make(nchains=64, ndim=d, multitry=k, history_capacity=1000, trace_capacity=1, seed=99, ncr=3, hardboundaries=0)
        Z = np.arange(120, dtype=float).reshape(30, d)
        e.set_history(Z)
        e.set_state(np.tile(np.array([2., 3., 4., 5.]), (64, 1)))
        e.set_likelihood_mvn(np.zeros(d), np.zeros((d, d)), 0, 0.0)
        q0 = np.array([2., 3., 4., 5.])
        for cr_idx, CR in ((2, 1.0), (0, 1 / 3.), (1, 2 / 3.)):
            kept = 0
            total = 0
            for c in range(64):
                for g in range(60 if k == 1 else 12):
                    pts, _, _, _ = e.debug_propose(c, g, 0, q0, False, cr_idx)
                    assert pts.shape == (k, d)                                   # :209-210, :223-224
                    kept += int((pts == q0).sum())
                    total += pts.size
            # d' = 0 redraws nothing here: the reference keeps the point unchanged in that case as well (Dream.py:704-712
            # only changes which gamma is used), so E[kept fraction] = 1 - CR
            assert abs(kept / total - (1 - CR)) < 0.05, (k, CR, kept / total)
        pts, slogp, _, _ = e.debug_propose(0, 0, 0, q0, True, 2)               # test_proposal_generation_snooker (:307-321)
        assert pts.shape == (k, d) and slogp.shape == (k,) and np.all(np.isfinite(slogp))
        e.close()


def check_history_sampling(make):
    """test_chain_sampling_simple_model / _multidim_model (:160-200): with exactly two rows in the history the
    sampled pair is those two rows (random.sample without replacement)."""
    for d in (1, 4):
        e = make(nchains=4, ndim=d, multitry=1, history_capacity=100, trace_capacity=1, seed=3, hardboundaries=0)
        Z = np.array([[1.5] * d, [-2.25] * d])
        e.set_history(Z)
        e.set_state(np.zeros((4, d)))
        e.set_likelihood_mvn(np.zeros(d), np.zeros((d, d)), 0, 0.0)
        for c in range(4):
            for g in range(25):
                pts, _, _, zidx = e.debug_propose(c, g, 0, np.zeros(d), False, 2)
                diff = pts[0]                                   # q0 = 0: prop = e*gamma*(za - zb) + zeta
                assert np.all(np.abs(np.abs(diff) / 3.75) > 0.5)    # |za - zb| = 3.75 in every dimension: never 0 (za != zb)
        e.close()


def check_multitry_selection(make):
    """test_multitry_proposal_selection (:345-355): log-likelihoods (1000, 500) -> the first proposal is always chosen.
    The host-callback likelihood sees the k proposals of a chain in consecutive rows."""
    k, n, d = 3, 8, 4
    e = make(nchains=n, ndim=d, multitry=k, history_capacity=2000, trace_capacity=30, seed=7, hardboundaries=0, snooker=0.0)
    rng = np.random.default_rng(0)
    Z = rng.normal(size=(64, d))
    e.set_history(Z)
    e.set_state(Z[:n])
    calls = {"n": 0}

    def like(X):
        m = X.shape[0]
        calls["n"] += 1
        out = np.full(m, 500.0)
        if m % k == 0:                  # a proposal batch (one chain's k points, or all chains' n*k); n and n*(k-1) are not multiples of k here
            out[0::k] = 1000.0          # proposal 0 of every chain
        return np.zeros(m), out

    e.set_likelihood_host(like)
    e.step(30)
    tr = e.get_trace(0, 30)
    assert np.all(tr["try_idx"] == 0)
    e.close()


def check_history_recording(make):
    """test_history_recording_simple_model / _multidim_model (:397-439): rows are appended in chain order, one per chain
    per recorded generation, whether or not the move was accepted (Dream.py:919-945)."""
    for d in (1, 4):
        n, G, thin = 3, 12, 4
        e = make(nchains=n, ndim=d, multitry=1, history_capacity=200, trace_capacity=G, seed=11, history_thin=thin, hardboundaries=0)
        rng = np.random.default_rng(1)
        Z0 = rng.normal(size=(20, d))
        e.set_history(Z0)
        e.set_state(Z0[:n])
        e.set_likelihood_mvn(np.zeros(d), np.eye(d), 0, 0.0)
        e.step(G)
        tr = e.get_trace(0, G)
        H = e.get_history()
        rec = [g for g in range(G) if g % thin == 0]             # iteration 0 included (:360)
        assert H.shape[0] == 20 + n * len(rec)
        np.testing.assert_array_equal(H[:20], Z0)
        for i, g in enumerate(rec):
            np.testing.assert_array_equal(H[20 + n * i: 20 + n * (i + 1)], tr["X"][g])
        e.close()


def check_boundaries(make):
    """test_boundaries_obeyed_aftersampling (:670-708): with hardboundaries every sample stays inside the support."""
    n, d, G = 16, 4, 200
    e = make(nchains=n, ndim=d, multitry=5, history_capacity=4000, trace_capacity=G, seed=21, hardboundaries=1)
    lo, hi = np.array([-1., 0., -3., 2.]), np.array([1., 0.5, 3., 2.25])
    rng = np.random.default_rng(2)
    Z = lo + rng.uniform(size=(80, d)) * (hi - lo)
    e.set_bounds(lo, hi)
    e.set_history(Z)
    e.set_state(Z[:n])
    e.set_likelihood_mvn(0.5 * (lo + hi), np.eye(d), 0, 0.0)
    e.step(G)
    X = e.get_trace(0, G)["X"]
    assert np.all(X >= lo) and np.all(X <= hi)
    assert e.get_trace(0, G)["moved"].mean() > 0.05
    e.close()
