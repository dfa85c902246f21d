#!/usr/bin/env python3
"""Differential fuzzing of the ORACLE against the REFERENCE itself (runs only where /root/reference exists, like make_golden.py whose
machinery it uses): random small configurations are run through the reference's own Dream.astep with the contract's draws injected
(make_golden.run_reference), then through the oracle from the same inputs, and compared the way the committed fixtures are
(tests/helpers.compare_with_reference: decision sequences exact, log densities 1e-10, states 1e-9 relative, archive, adapted
probabilities).  Nothing is written to tests/golden; the point is that the pinning does not hang on the dozen committed cases.

    PYTHONPATH=/root/reference python tests/golden/fuzz_reference.py --n 200 --seed 1
    PYTHONPATH=/root/reference python tests/golden/fuzz_reference.py --n 300 --seed 1 --tri

--tri (round 5): the triangular arm.  Every case has the MVN target, at dimensions up to 100, and after the comparison with the dense
precision matrix (the reference's own formula, dream_ex_ndim_gaussian.py:49-52) the oracle runs the case AGAIN with the matrix handed over
as its triangular factor (kind 1: log p = log_F - |U x|^2 / 2 -- what bench.py's headline times) and is compared with the same reference
output: the two forms differ in the last bits of log p, so this measures whether such a difference ever flips a selection (Dream.py:908)
or an accept (:993).  A flip is reported as a mismatch, with the case.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

import make_golden as MG                      # noqa: E402  (asserts that pydream is the reference's)
from oracle import oracle as O                # noqa: E402
from tests import helpers as H                # noqa: E402


def draw_case(rng):
    prior = str(rng.choice(["flat", "flat", "normal", "uniform", "uniform_wide_history"]))
    d = int(rng.integers(2, 5)) if prior.startswith("uniform") else int(rng.choice([2, 3, 4, 7, 10, 12]))
    N = int(rng.integers(3, 8))
    k = int(rng.choice([1, 3, 4, 5, 5, 9, 17, 24, 32]))          # (beyond 16 tries: round 5)
    if prior == "uniform_wide_history" and k == 1:
        k = 3
    schedule = int(rng.choice([1, 2, 2]))
    thin = int(rng.choice([1, 3, 10]))
    G = thin * int(np.ceil(rng.integers(12, 32) / thin))          # a multiple of history_thin: the reference sizes its history as floor(N G / thin)
    depairs = int(rng.choice([1, 1, 2, 3]))                      # rows (core.py:260-268) but appends at iterations 0, thin, 2 thin, ...
    N = max(N, 2 * depairs + 1)                                  # core.py:253-254
    tgt = str(rng.choice(["mvn", "mix", "simple"]))
    if tgt == "mvn":
        target = ("mvn",)
    elif tgt == "mix":
        J = int(rng.choice([2, 3]))
        w = rng.dirichlet(np.ones(J))
        target = ("mix", [float(m) for m in np.linspace(-4, 5, J)], [float(x) for x in w])
    else:
        target = ("simple",)
    ngamma = int(rng.choice([1, 1, 2, 3]))
    adapt_cr = bool(rng.random() < 0.5)
    adapt_g = bool(ngamma > 1 and rng.random() < 0.5)
    kw = dict(nCR=int(min(d, rng.choice([1, 2, 3, 3]))), DEpairs=depairs, gamma_levels=ngamma, adapt_crossover=adapt_cr, adapt_gamma=adapt_g,
              snooker=float(rng.choice([0.0, 0.1, 0.3])), p_gamma_unity=float(rng.choice([0.0, 0.2, 0.5])), history_thin=thin,
              lamb=float(rng.choice([0.05, 0.2])), zeta=float(rng.choice([1e-12, 1e-6])))
    if adapt_cr or adapt_g:
        kw["crossover_burnin"] = int(rng.choice([5, 10, G + 5]))         # (at iter == burnin the reference spins on a barrier in one process: make_golden handles S2, S1 needs it beyond the run)
        if schedule == 1:
            kw["crossover_burnin"] = G + 5
    if prior == "uniform_wide_history":
        kw["hardboundaries"] = False
    elif rng.random() < 0.2:
        kw["hardboundaries"] = False if prior in ("flat", "normal") else True
    lag = int(rng.choice([0, 0, 1, 2, 3])) if schedule == 2 else 0
    c = dict(d=d, N=N, G=G, k=k, schedule=schedule, seed=int(rng.integers(1, 2 ** 31 - 1)), target=target, dream_kwargs=kw, prior=prior,
             rng_seed=int(rng.integers(0, 2 ** 31 - 1)), history_lag=lag)
    if schedule == 2 and (adapt_cr or adapt_g) and (adapt_lag_arm or rng.random() < 0.5):
        # adapt_lag (round 6): the updates reach the chains' decisions L generations late; burn-ins whose adaptation window (generations
        # 11 .. burn-in - 1, Dream.py:371) lies inside the run, ends inside it (the hand-over flushes what is held) or beyond it
        c["adapt_lag"] = int(rng.choice([1, 2, 3, 9, 19]))
        kw["crossover_burnin"] = int(rng.choice([10, 14, max(12, G - 6), G + 5]))
    return c


adapt_lag_arm = False


def draw_tri_case(rng):
    """MVN target only, dimensions up to the headline's 100 (the reference's astep takes ~1 ms a call: the large ones are short)"""
    c = draw_case(rng)
    while c["prior"].startswith("uniform"):
        c = draw_case(rng)
    d = int(rng.choice([2, 3, 5, 8, 12, 17, 33, 64, 100]))
    c["d"] = d
    c["dream_kwargs"]["nCR"] = int(min(d, c["dream_kwargs"]["nCR"]))
    c["target"] = ("mvn",)
    c["tri"] = True
    if d > 20:
        thin = c["dream_kwargs"]["history_thin"]
        c["G"] = thin * int(np.ceil(min(c["G"], 16) / thin))
    return c


def draw_pt_case(rng):
    """parallel tempering through the reference's own _sample_dream_pt (make_golden.run_reference_pt): MVN target, reference defaults"""
    G = int(rng.choice([20, 30, 40]))
    kw = {}
    if rng.random() < 0.5:
        kw = dict(adapt_crossover=True, crossover_burnin=int(rng.choice([5, 12, G + 5])))
    return dict(pt=True, d=int(rng.choice([3, 5, 10])), N=int(rng.integers(3, 8)), G=G, k=int(rng.choice([3, 5])), seed=int(rng.integers(1, 2 ** 31 - 1)),
                rng_seed=int(rng.integers(0, 2 ** 31 - 1)), dream_kwargs=kw)


def run_pt_case(c):
    captured = {}
    MG.save = lambda name, **arrs: captured.update({k: np.asarray(v) for k, v in arrs.items()})
    c = dict(c); c.pop("pt")
    MG.pt_case("fuzz_pt", **c)
    e = H.pt_engine_from_fixture(O.Engine, captured)
    e.step(int(captured["cfg_G"]))
    H.compare_pt_with_reference(e, captured)
    return 0


def run_case(c):
    if c.get("pt"):
        return run_pt_case(c)
    captured = {}
    MG.save = lambda name, **arrs: captured.update({k: np.asarray(v) for k, v in arrs.items()})       # (trace_case hands its arrays to save)
    c = dict(c); tri = c.pop("tri", False)
    MG.trace_case("fuzz", **c)
    fx = captured
    if int(fx["redraws"].max()) >= 64:         # beyond DZ_MAX_REDRAWS the engines give the step up as a rejection (DESIGN.md deviation D1): not comparable
        return -1
    e = H.engine_from_trace_fixture(O.Engine, fx)
    G = int(fx["cfg_G"])
    if int(fx["cfg_schedule"]) == 2 and (int(fx["cfg_adapt_crossover"]) or int(fx["cfg_adapt_gamma"])):
        for g in range(G):     # the shared probabilities as the reference's chains adopted them after EVERY generation (what an adapt_lag shifts)
            e.step(1)
            np.testing.assert_allclose(e.get_cr_state()[0], fx["cross_probs"][g], rtol=1e-11, atol=0, err_msg="cross_probs after generation %d" % g)
            np.testing.assert_allclose(e.get_gamma_state()[0], fx["gamma_probs"][g], rtol=1e-11, atol=0, err_msg="gamma_probs after generation %d" % g)
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
    if tri:
        assert str(fx["lk_kind"]) == "mvn"
        e = H.engine_from_trace_fixture(O.Engine, fx, mvn_kind="tri")
        e.step(G)
        try:
            H.compare_with_reference(e.get_trace(0, G), fx, e.get_history(), e.get_cr_state()[0], e.get_gamma_state()[0] if int(fx["cfg_adapt_gamma"]) else None)
        except AssertionError as ex:
            raise AssertionError("TRIANGULAR FACTOR: " + str(ex).strip().split("\n")[0])
    return int((fx["redraws"] > 0).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--adapt-lag", action="store_true", help="the adapt_lag arm: every lockstep case with an adaptation gets a lag (else: half of them)")
    ap.add_argument("--tri", action="store_true", help="the triangular arm: MVN cases up to 100-D, each also run with the triangular factor")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    global adapt_lag_arm
    adapt_lag_arm = args.adapt_lag
    bad = 0; redrawn = 0; skipped = 0; t0 = time.time()
    devnull = open(os.devnull, "w")
    for i in range(args.n):
        c = draw_tri_case(rng) if args.tri else (draw_pt_case(rng) if rng.random() < 0.1 else draw_case(rng))
        out = sys.stdout
        try:
            sys.stdout = devnull
            r = run_case(c)
            redrawn += r > 0
            skipped += r < 0
        except Exception as ex:
            sys.stdout = out
            bad += 1
            print("MISMATCH #%d: %s\n   %s" % (i, str(ex).strip().split("\n")[0][:300], c), flush=True)
        finally:
            sys.stdout = out
    print("fuzz vs reference%s: %d cases, %d mismatches, %d with redraw rounds, %d skipped (a step of the reference took 64 or more redraw rounds: deviation D1), %.0f s (seed %d)"
          % (" (triangular arm: every case also with the factor U of the precision matrix)" if args.tri else "", args.n, bad, redrawn, skipped, time.time() - t0, args.seed))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
