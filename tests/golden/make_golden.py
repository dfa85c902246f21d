#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container: it imports PyDREAM read-only from
/root/reference (nothing from it is copied here), drives its unmodified
``Dream.astep`` / ``generate_proposal_points`` / ``Gelman_Rubin`` / example
likelihoods, and stores INPUTS and the reference's OUTPUTS as small ``.npz``
files.  The GPU box only ever sees those files.

How the reference is made deterministic:
  * single process: ``_setup_mp_dream_pool`` + ``pool._initializer`` in the
    parent, the reference's own test idiom (pydream/tests/test_dream.py:507-508);
  * the module globals ``pydream.Dream.np`` / ``pydream.Dream.random`` are
    rebound to proxies whose draws come from this repo's counter-based random
    contract (DESIGN.md "Random contract"; evaluated through the oracle's
    building blocks), so reference, oracle and HIP engine consume identical
    numbers;
  * schedule S1 = round-robin over chains (the unmodified class); schedule S2 =
    a test-side subclass that defers record_history / set_current_position_arr
    / estimate_*_probs to the end of the generation and replays the base-class
    methods in chain order.

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import copy
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import oracle as O  # noqa: E402  (test infrastructure)

import pydream.Dream as RD  # noqa: E402
from pydream import Dream_shared_vars as SV  # noqa: E402
from pydream.core import _setup_mp_dream_pool  # noqa: E402
import pydream.core as RC  # noqa: E402
from pydream.model import Model  # noqa: E402
from pydream.parameters import FlatParam, SampledParam  # noqa: E402
from pydream.convergence import Gelman_Rubin  # noqa: E402

# the repository has a `pydream` alias package of its own (pydream/__init__.py -> pydream_amd): the vectors must come from the
# reference, never from the implementation under test
for _m in (RD, SV, RC):
    assert os.path.realpath(_m.__file__).startswith(os.path.realpath("/root/reference") + os.sep), \
        "%s was imported from %s, not from /root/reference" % (_m.__name__, _m.__file__)


# --------------------------------------------------------------------------
# the random contract, as seen through numpy.random / random (App. B order)
# --------------------------------------------------------------------------
class ContractRandom:
    """Stands in for ``np.random`` AND for the ``random`` module inside pydream.Dream."""

    def __init__(self, seed, k, ncr_ctrl_has_snooker=True):
        self.seed = seed
        self.k = k
        self.has_snk = ncr_ctrl_has_snooker
        self.log = {}

    def begin_step(self, gc, g, p_gamma_unity):
        self.gc, self.g = gc, g
        self.pgu = p_gamma_unity
        self.phase = 0
        self.round = 0
        self.n_ctrl = 0
        self.cnt = {}
        self.snk = False
        self.log = dict(snooker=0, cr_idx=-1, sel=0, glev=1, delta=1, redraws=0)
        s = O.stream_id(O.K_CTRL)
        w0 = O.philox(self.seed, 0, s, gc, g)
        w1 = O.philox(self.seed, 1, s, gc, g)
        w2 = O.philox(self.seed, 2, s, gc, g)
        self.u_snk, self.u_cr = O.u53(w0[0], w0[1]), O.u53(w0[2], w0[3])
        self.u_de, self.u_glev = O.u53(w1[0], w1[1]), O.u53(w1[2], w1[3])
        self.u_sel, self.u_acc = O.u53(w2[0], w2[1]), O.u53(w2[2], w2[3])

    def _next(self, name):
        key = (name, self.phase)
        v = self.cnt.get(key, 0)
        self.cnt[key] = v + 1
        return v

    def _ntries(self):
        return self.k if self.phase == 0 else self.k - 1

    # Redraw rounds (Dream.py:281-289: the proposal set is generated again while every try is impossible): the calls of round r
    # follow those of round r-1 in the same phase, so a per-phase call count c splits into (round, try) = divmod(c, tries per
    # call); round r >= 1 draws from the Philox key seed + r * REDRAW_KEY_STEP (oracle/dreamzs_oracle.h).
    REDRAW_KEY_STEP = 0x9E3779B97F4A7C15

    def _key(self):
        return (self.seed + self.round * self.REDRAW_KEY_STEP) & 0xFFFFFFFFFFFFFFFF

    def _split(self, c, per_call=None):
        rnd, tr = divmod(c, per_call or self._ntries())
        self.round = rnd
        self.log["redraws"] = max(self.log.get("redraws", 0), rnd if self.phase == 0 else 0)
        return tr

    def _pt(self, tr, idx):
        return O.philox(self._key(), idx, O.stream_id(O.K_PT, tr, self.phase), self.gc, self.g)

    def _dim(self, tr, d):
        """per-dimension draws of one try: arrays (U, u_e, z); one Philox call per PAIR of dimensions."""
        s = O.stream_id(O.K_DIM, tr, self.phase)
        U, ue, z = np.zeros(d), np.zeros(d), np.zeros(d)
        for q in range((d + 1) // 2):
            w = O.philox(self._key(), q, s, self.gc, self.g)
            zz = (O.normal32(w[2], w[3]), O.normal32_sin(w[2], w[3]))
            for h in (0, 1):
                j = 2 * q + h
                if j < d:
                    U[j] = O.u16(int(w[0]) >> (16 * h)); ue[j] = (int(w[1]) >> (16 * h)) & 0xffff; z[j] = zz[h]     # ue: the raw 16-bit draw
        return U, ue, z

    @staticmethod
    def _onehot(n, m):
        out = np.zeros(n, dtype=int)
        out[m] = 1
        return out

    # ---- numpy.random API used by pydream/Dream.py ----
    def multinomial(self, n, pvals):
        pvals = np.asarray(pvals, dtype=float)
        ctrl_order = (["snk"] if self.has_snk else []) + ["cr", "glev"]
        if self.n_ctrl < len(ctrl_order):
            which = ctrl_order[self.n_ctrl]
            self.n_ctrl += 1
            if which == "snk":      # Dream.py:545
                m = O.invcdf(pvals, self.u_snk)
                self.snk = (m == 0)
                self.log["snooker"] = int(self.snk)
            elif which == "cr":     # Dream.py:565
                m = O.invcdf(pvals, self.u_cr)
                self.log["cr_idx"] = m
            else:                   # Dream.py:595
                m = O.invcdf(pvals, self.u_glev)
                self.log["glev"] = m + 1
            return self._onehot(len(pvals), m)
        if len(pvals) == 2 and pvals[0] == self.pgu and self.k != 2:      # set_gamma, Dream.py:615
            tr = self._split(self._next("gu"))
            w = self._pt(tr, 0)
            return self._onehot(2, O.invcdf(pvals, O.u53(w[0], w[1])))
        # mt_choose_proposal_pt, Dream.py:908
        m = O.invcdf(pvals, self.u_sel)
        self.log["sel"] = m
        self.phase = 1
        self.round = 0
        return self._onehot(len(pvals), m)

    def randint(self, lo, hi, size=None):                                    # set_DEpair, Dream.py:580
        v = lo + int(np.floor(self.u_de * (hi - lo)))
        self.log["delta"] = v
        return np.array([v])

    def normal(self, loc, scale, size):                                      # Dream.py:694
        tr = self._split(self._next("normal"))
        return loc + scale * self._dim(tr, size)[2]

    def uniform(self, low=None, high=None, size=None):
        if low is None:                                                      # metrop_select, Dream.py:993
            return self.u_acc
        if size is None:                                                     # snooker gamma, Dream.py:618
            self._split(self._next("sgamma"), 1)
            w = self._pt(0, 0)
            return low + (high - low) * O.u53(w[2], w[3])
        if isinstance(size, tuple):                                          # U, Dream.py:700
            n, d = size
            self._split(self._next("U"), 1)
            return np.array([self._dim(tr, d)[0] for tr in range(n)])
        tr = self._split(self._next("e"))                                           # e, Dream.py:696: uniform(low, high) from the 16-bit draws, one fma each
        return np.array([O.uniform16(int(h), low, high) for h in self._dim(tr, size)[1]])

    def rand(self, n):                                                       # bounds redraw, Dream.py:749-751, 773-775
        fr = sys._getframe(1).f_locals
        tr = int(fr.get("pt_num", 0)) if self._ntries() > 1 else 0
        masks = [m for m in (fr["x_lower"], fr["x_upper"]) if np.any(m)]
        key = ("bnd", self.phase, self.round, tr)
        c = self.cnt.get(key, 0)
        self.cnt[key] = c + 1
        mask = np.atleast_1d(masks[c])
        dims = np.where(mask)[0]
        assert len(dims) == n
        s = O.stream_id(O.K_BND, tr, self.phase)
        return np.array([O.u32(O.philox(self._key(), int(j), s, self.gc, self.g)[0]) for j in dims])

    # ---- random.sample used by sample_from_history, Dream.py:662-664 ----
    def sample(self, rng, n):
        M = len(rng)
        if not self.snk:
            tr = self._split(self._next("sample"))
            words = np.concatenate([self._pt(tr, 1 + q) for q in range((n + 3) // 4)])[:n]
            return [int(x) for x in O.sample_distinct(words, M)]
        nt = self._ntries()
        c = self._split(self._next("sample"), 3 * nt)
        if c < nt:
            tr, word = c, 0
        else:
            tr, word = (c - nt) // 2, 1 + (c - nt) % 2
        w = self._pt(tr, 1)
        return [int((int(w[word]) * M) >> 32)]


class NPProxy:
    """Forwards to numpy, except ``.random`` (the contract) and the two ufunc calls the reference makes with a
    ``where=`` mask and no ``out=`` (Dream.py:329, :824, :831, :835): numpy leaves the masked-out result
    UNINITIALISED there (whenever |x - z| == 0, i.e. a chain draws its own archived state), so the reference's
    value is whatever was in memory.  The contract defines it as 0; the proxy makes
    the reference do exactly that by supplying a zeroed ``out``."""

    def __init__(self, rnd):
        self.random = rnd

    @staticmethod
    def _masked(ufunc, args, where, kw):
        if where is True:
            return ufunc(*args, **kw)
        shape = np.broadcast(*[np.asarray(a) for a in args], np.asarray(where)).shape
        out = np.zeros(shape, dtype=float)
        ufunc(*args, out=out, where=where, **kw)
        return out if shape else float(out)

    def log(self, x, where=True, **kw):
        return self._masked(np.log, (x,), where, kw)

    def divide(self, a, b, where=True, **kw):
        return self._masked(np.divide, (a, b), where, kw)

    def __getattr__(self, n):
        return getattr(np, n)


class NoSleep:
    @staticmethod
    def sleep(_):
        return None


def install(rnd):
    RD.np = NPProxy(rnd)
    RD.random = rnd
    RD.time = NoSleep


def uninstall():
    import random as pyrandom
    import time as pytime
    RD.np = np
    RD.random = pyrandom
    RD.time = pytime


# --------------------------------------------------------------------------
# schedule S2 from the reference
# --------------------------------------------------------------------------
class DeferredDream(RD.Dream):
    queue = None

    def record_history(self, *a, **k):
        self.queue.append(("hist", a, k))

    def set_current_position_arr(self, ndimensions, q_new):
        if self.nchains is None:
            cp = np.frombuffer(SV.current_positions.get_obj())
            self.nchains = len(cp) // ndimensions
        self.queue.append(("pos", (ndimensions, np.array(q_new).copy()), {}))

    def estimate_crossover_probabilities(self, *a, **k):
        self.queue.append(("cr", a, k))
        return self.CR_probabilities

    def estimate_gamma_level_probs(self, *a, **k):
        self.queue.append(("gam", a, k))
        return self.gamma_probabilities


BASE = {"hist": RD.Dream.record_history, "pos": RD.Dream.set_current_position_arr,
        "cr": RD.Dream.estimate_crossover_probabilities, "gam": RD.Dream.estimate_gamma_level_probs}


def run_reference(params, likelihood, Z0, starts, N, G, seed, schedule, dream_kwargs, workdir, history_lag=0, adapt_lag=0):
    """Drive the reference's astep for G generations; return everything it produced.

    adapt_lag (schedule S2 only): the deferred estimate_crossover_probabilities / estimate_gamma_level_probs calls of generation g are
    replayed `adapt_lag` generations late, i.e. after generation g + adapt_lag (all that are still held at the hand-over, generation
    crossover_burnin: the barrier of Dream.py:385-415) -- through the base-class methods, oldest generation first, with the shared
    current_positions array holding, for the duration of the replay, what it held when generation g ended (the positions the
    reference's np.std of Dream.py:476 / :520 is taken over).  The chains adopt the shared probabilities after every generation as
    before, so generation g decides with the probabilities as they were after the updates of generations <= g - 1 - adapt_lag: the
    schedule under which a kernel launch can hold adapt_lag + 1 burn-in generations (include/dreamzs.h dz_config.adapt_lag).  The
    reference's own chains see each other's updates with a scheduler-dependent delay (Dream.py:371-378 under core.py:80).

    history_lag (schedule S2 only): the deferred record_history calls of an appending generation are replayed `history_lag` appends
    late -- the reference's own record_history / sample_from_history then see the archive grow with that delay (its `count` lags),
    which is the schedule the engine runs when the exchange of appended rows between GPUs is hidden behind the next thin-cycle
    (include/dreamzs.h dz_config.history_lag).  Row positions do not change: record_history appends at `count`, and the held-back
    appends are replayed oldest first; whatever is still held at the end of the run is flushed before the archive is read out."""
    d = Z0.shape[1]
    hist_file = os.path.join(workdir, "seed_hist.npy")
    np.save(hist_file, Z0)
    cls = RD.Dream if schedule == 1 else DeferredDream
    model = Model(likelihood=likelihood, sampled_parameters=params)
    step = cls(model=model, variables=None, history_file=hist_file, start_random=False, save_history=False,
               verbose=False, **dream_kwargs)
    k = step.multitry
    rnd = ContractRandom(seed, k, ncr_ctrl_has_snooker=(step.snooker != 0))
    install(rnd)
    try:
        pool = _setup_mp_dream_pool(N, G, step, start_pt=[starts[i] for i in range(N)])
        pool._initializer(*pool._initargs)
        pool.close()
        pool.join()
        burnin = step.crossover_burnin
        chains = [copy.copy(step) for _ in range(N)]
        for c in chains:
            c.queue = []
        x = [np.array(starts[i], dtype=float).copy() for i in range(N)]
        out = dict(X=np.zeros((G, N, d)), logp=np.zeros((G, N)), prior=np.zeros((G, N)), like=np.zeros((G, N)),
                   moved=np.zeros((G, N), np.uint8), try_idx=np.zeros((G, N), np.int32),
                   cr_idx=np.zeros((G, N), np.int32), snooker=np.zeros((G, N), np.uint8),
                   redraws=np.zeros((G, N), np.int32),
                   cross_probs=np.zeros((G, step.nCR)), gamma_probs=np.zeros((G, step.ngamma)),
                   hist_rows=np.zeros(G, np.int64))
        held = []                                         # appends not replayed yet (history_lag), oldest first
        pend = []                                         # adaptation updates not replayed yet (adapt_lag), oldest first: (g, positions, calls)
        for g in range(G):
            for ci, c in enumerate(chains):
                rnd.begin_step(ci, g, step.p_gamma_unity)
                if g == burnin:
                    SV.nchains.value = N - 1          # lets the barrier at Dream.py:403 fall through
                xn, lp, ll = c.astep(x[ci])
                xn = np.array(xn, dtype=float).copy()
                out["X"][g, ci] = xn
                out["prior"][g, ci], out["like"][g, ci] = lp, ll
                out["logp"][g, ci] = ll + lp                       # core.py:115
                out["moved"][g, ci] = int(np.any(xn != x[ci]))     # core.py:120
                out["try_idx"][g, ci] = rnd.log["sel"]
                out["cr_idx"][g, ci] = rnd.log["cr_idx"]
                out["snooker"][g, ci] = rnd.log["snooker"]
                out["redraws"][g, ci] = rnd.log.get("redraws", 0)
                x[ci] = xn
            if schedule == 2:
                for c in chains:
                    for (kk, a, kw) in c.queue:
                        if kk == "pos":
                            c.iter -= 1           # astep already advanced iter (Dream.py:417); the
                            BASE["pos"](c, *a, **kw)   # base methods must see this generation's value (:446)
                            c.iter += 1
                calls = [(kind, c, a, kw) for kind in ("cr", "gam") for c in chains for (kk, a, kw) in c.queue if kk == kind]
                if calls:
                    pend.append((g, np.frombuffer(SV.current_positions.get_obj()).copy(), calls))
                while pend and pend[0][0] <= (g if g == burnin else g - adapt_lag):
                    _, snap, calls = pend.pop(0)
                    cp = np.frombuffer(SV.current_positions.get_obj())
                    now = cp.copy()
                    cp[:] = snap                      # what current_positions held when that generation ended
                    for (kind, c, a, kw) in calls:
                        c.iter -= 1
                        BASE[kind](c, *a, **kw)
                        c.iter += 1
                    cp[:] = now
                this_append = [(c, a, kw) for c in chains for (kk, a, kw) in c.queue if kk == "hist"]       # chain order
                if this_append:
                    held.append(this_append)
                while len(held) > history_lag or (g == G - 1 and held):
                    for (c, a, kw) in held.pop(0):
                        BASE["hist"](c, *a, **kw)
                for c in chains:
                    c.queue = []
                    if g <= burnin:
                        if step.adapt_crossover:
                            c.CR_probabilities = list(SV.cross_probs[0:step.nCR])
                        if step.adapt_gamma:
                            c.gamma_probabilities = list(SV.gamma_level_probs[0:step.ngamma])
            out["cross_probs"][g] = SV.cross_probs[0:step.nCR]
            out["gamma_probs"][g] = SV.gamma_level_probs[0:step.ngamma]
            out["hist_rows"][g] = SV.count.value + int(step.nseedchains)
        M = int(SV.count.value + step.nseedchains)
        out["Z_tail"] = np.array(SV.history[0:M * d]).reshape(M, d)[len(Z0):]      # rows appended during the run
        out["delta_m"] = np.array(SV.delta_m[:])
        out["ncr_updates"] = np.array(SV.ncr_updates[:])
        out["delta_m_gamma"] = np.array(SV.delta_m_gamma[:])
        out["ngamma_updates"] = np.array(SV.ngamma_updates[:])
        out["burnin"] = burnin
        out["gamma_arr"] = step.gamma_arr
        if step.boundaries:
            out["mins"], out["maxs"] = np.asarray(step.mins, float), np.asarray(step.maxs, float)
        return out
    finally:
        uninstall()


# --------------------------------------------------------------------------
# parallel tempering from the reference (core.py:131-248): `_sample_dream_pt` itself runs, with a pool whose map()
# executes the reference's `_sample_dream_pt_chain` for every chain in this process (schedule S2: the deferred
# end-of-generation updates are flushed after the last chain), and with core.py's numpy rebound to a proxy that
# serves the swap draws of the contract: stream SWAP(=4), counter (0, stream, 0, generation):
# (w0, w1) -> np.random.choice(nchains, 2, replace=False) (:185), u53(w2, w3) -> np.random.uniform() (:197).
# --------------------------------------------------------------------------
class CoreRandom:
    def __init__(self, seed, pool):
        self.seed, self.pool = seed, pool

    def _w(self):
        return O.philox(self.seed, 0, O.stream_id(4, 0, 0), 0, self.pool.g - 1)

    def choice(self, n, size, replace=False):
        assert size == 2 and not replace
        w = self._w()
        a = (int(w[0]) * n) >> 32
        b = (int(w[1]) * (n - 1)) >> 32
        if b >= a:
            b += 1
        self.pool.swaps.append([a, b])
        return np.array([a, b])

    def uniform(self):
        w = self._w()
        return O.u53(w[2], w[3])


class CoreNPProxy:
    def __init__(self, seed, pool):
        self.random = CoreRandom(seed, pool)

    def log(self, x):
        return np.float64(O.log(float(x)))           # np.log(np.random.uniform()) :197 -- the contract's log

    def __getattr__(self, n):
        return getattr(np, n)


class InProcessPool:
    """pool.map(_sample_dream_pt_chain, args) without processes; per-chain Dream copies like the pickled ones."""

    def __init__(self, rnd, step, N, burnin):
        self.rnd, self.step, self.N, self.g, self.chains, self.swaps, self.log, self.burnin = rnd, step, N, 0, None, [], [], burnin
        self.probs = []

    def map(self, fn, args):
        args = list(args)
        if self.chains is None:
            self.chains = [copy.copy(self.step) for _ in range(self.N)]
            for c in self.chains:
                c.queue = []
        out = []
        row = []
        for ci, a in enumerate(args):
            self.rnd.begin_step(ci, self.g, self.step.p_gamma_unity)
            if self.g == self.step.crossover_burnin:          # (set by _setup_mp_dream_pool when it was None)
                SV.nchains.value = self.N - 1          # lets the barrier at Dream.py:403 fall through
            res = fn((self.chains[ci],) + tuple(a[1:]))
            out.append((np.array(res[0], dtype=float).copy(), res[1], res[2], self.chains[ci]))
            row.append((self.rnd.log["sel"], self.rnd.log["cr_idx"], self.rnd.log["snooker"]))
        for kind in ("pos", "cr", "gam", "hist"):                     # schedule S2 flush, as in run_reference
            for c in self.chains:
                for (kk, aa, kw) in c.queue:
                    if kk == kind:
                        c.iter -= 1
                        BASE[kind](c, *aa, **kw)
                        c.iter += 1
        for c in self.chains:
            c.queue = []
            if self.g <= self.step.crossover_burnin:                  # as in run_reference: the chains adopt the shared probabilities
                if self.step.adapt_crossover:
                    c.CR_probabilities = list(SV.cross_probs[0:self.step.nCR])
                if self.step.adapt_gamma:
                    c.gamma_probabilities = list(SV.gamma_level_probs[0:self.step.ngamma])
        self.log.append(row)
        self.probs.append(np.array(SV.cross_probs[0:self.step.nCR]))
        self.g += 1
        return out


def run_reference_pt(params, likelihood, Z0, starts, N, G, seed, dream_kwargs, workdir):
    d = Z0.shape[1]
    hist_file = os.path.join(workdir, "seed_hist.npy")
    np.save(hist_file, Z0)
    model = Model(likelihood=likelihood, sampled_parameters=params)
    step = DeferredDream(model=model, variables=None, history_file=hist_file, start_random=False, save_history=False,
                         verbose=False, **dream_kwargs)
    rnd = ContractRandom(seed, step.multitry, ncr_ctrl_has_snooker=(step.snooker != 0))
    install(rnd)
    fake = InProcessPool(rnd, step, N, step.crossover_burnin)
    RC.np = CoreNPProxy(seed, fake)
    try:
        pool = _setup_mp_dream_pool(N, G, step, start_pt=[starts[i] for i in range(N)])
        pool._initializer(*pool._initargs)
        pool.close()
        pool.join()
        sampled, log_ps = RC._sample_dream_pt(N, G, step, [starts[i].copy() for i in range(N)], fake, False)
        M = int(SV.count.value + step.nseedchains)
        dec = np.array(fake.log, dtype=np.int32)                                   # [G, N, 3]
        T = np.array([np.power(.001, (float(i) / N)) for i in range(N)])           # core.py:133-136
        return dict(pt_sampled=np.asarray(sampled), pt_log_ps=np.asarray(log_ps), pt_swaps=np.array(fake.swaps, np.int32),
                    try_idx=dec[:, :, 0], cr_idx=dec[:, :, 1], snooker=dec[:, :, 2].astype(np.uint8), T=T,
                    Z_tail=np.array(SV.history[0:M * d]).reshape(M, d)[len(Z0):],
                    cross_probs=np.array(fake.probs), delta_m=np.array(SV.delta_m[:]), ncr_updates=np.array(SV.ncr_updates[:]),
                    burnin=step.crossover_burnin)
    finally:
        RC.np = np
        uninstall()


def pt_case(name, *, d, N, G, k, seed, rng_seed=0, dream_kwargs=None):
    rng = np.random.default_rng(rng_seed)
    params = [FlatParam(test_value=np.zeros(d))]
    nseed = max(10 * d, 2 * N)
    Z0 = rng.uniform(-5, 15, (nseed, d))
    invC, log_F, like = mvn_target(d)
    starts = Z0[:N].copy()
    kw = dict(multitry=k, adapt_crossover=False)
    kw.update(dream_kwargs or {})
    with tempfile.TemporaryDirectory() as wd:
        cwd = os.getcwd()
        os.chdir(wd)
        try:
            out = run_reference_pt(params, like, Z0, starts, N, G, seed, kw, wd)
        finally:
            os.chdir(cwd)
    save(name, Z0=Z0, starts=starts, invC=invC, log_F=log_F, cfg_d=d, cfg_N=N, cfg_G=G, cfg_k=k, cfg_seed=seed,
         cfg_adapt_crossover=int(bool(kw["adapt_crossover"])), **out)
    sw = out["pt_swaps"]
    acc = [(not np.array_equal(out["pt_sampled"][sw[g, 0], 2 * g], out["pt_sampled"][sw[g, 0], 2 * g + 1])) for g in range(G)]
    print("   ", name, "swap acceptance", np.mean(acc), "move acceptance (coldest chain)",
          np.mean(np.any(np.diff(out["pt_sampled"][0, ::2], axis=0) != 0, axis=1)))


# --------------------------------------------------------------------------
# target densities of the examples (restated from the cited lines; the d=200
# MVN and the 2-component mixture below call the reference modules themselves)
# --------------------------------------------------------------------------
def mvn_target(d):
    """examples/ndim_gaussian/dream_ex_ndim_gaussian.py:29-52 at dimension d."""
    A = .5 * np.identity(d) + .5 * np.ones((d, d))
    C = np.zeros((d, d))
    for i in range(d):
        for j in range(d):
            C[i][j] = A[i][j] * np.sqrt((i + 1) * (j + 1))
    invC = np.linalg.inv(C)
    log_F = 0 if d > 150 else np.log(((2 * np.pi) ** (-d / 2)) * np.linalg.det(C) ** (- 1. / 2))

    def likelihood(param_vec):
        return log_F - .5 * np.sum(param_vec * np.dot(invC, param_vec))
    return invC, float(log_F), likelihood


def mixture_target(d, means, weights):
    """examples/mixturemodel/mixturemodel.py:18-48 generalised to J components."""
    J = len(weights)
    mu = np.array([np.linspace(m, m, num=d) for m in means])
    log_F = np.log(np.array(weights)) - (d / 2.) * np.log(2 * np.pi)

    def likelihood(params):
        log_lh = np.zeros((J))
        for j in range(J):
            log_lh[j] = -.5 * np.sum((params - mu[j, :]) ** 2) + log_F[j]
        maxll = np.max(log_lh)
        post = np.array(np.exp(log_lh - maxll), dtype='float64')
        density = np.sum(post)
        return np.log(density) + maxll
    return mu, log_F, likelihood


def simple_likelihood(param):
    """pydream/tests/test_models.py:47-50"""
    return np.sum(param + 3)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def trace_case(name, *, d, N, G, k, schedule, seed, target, dream_kwargs=None, prior="flat", rng_seed=0, nseed=None, restart_from=None, history_lag=0, adapt_lag=0):
    """restart_from: name of an earlier trace fixture -- this run then restarts it the way run_dream(restart=True) does
    (core.py:46-62, 255-263; Dream.py:128-141): seed history = everything that run left in its history file (seed rows + appended
    rows), crossover probabilities loaded from its crossover file through Dream's `crossover_file`, starts = its last states."""
    from scipy.stats import norm, uniform
    dream_kwargs = dict(dream_kwargs or {})
    rng = np.random.default_rng(rng_seed)
    cfg = dict(d=d, N=N, G=G, k=k, schedule=schedule, seed=seed)
    extra = {}
    if history_lag:
        cfg["history_lag"] = history_lag
    if adapt_lag:
        cfg["adapt_lag"] = adapt_lag
    nseed = nseed or max(10 * d, 2 * N * dream_kwargs.get("DEpairs", 1))
    if prior == "flat":
        params = [FlatParam(test_value=np.zeros(d))]
        Z0 = rng.uniform(-5, 15, (nseed, d))
        pk, pa, pb = np.zeros(d, np.int32), np.zeros(d), np.ones(d)
    elif prior == "uniform":           # pydream/tests/test_models.py:35-45
        lower = np.array([-5., -9., 5., 3.])[:d]
        upper = np.array([10., 2., 7., 8.])[:d]
        params = [SampledParam(uniform, loc=lower, scale=upper - lower)]
        Z0 = lower + rng.uniform(0, 1, (nseed, d)) * (upper - lower)
        pk, pa, pb = np.full(d, 2, np.int32), lower, upper - lower
    elif prior == "uniform_wide_history":      # the same uniform prior, but an archive three times as wide as its support and no
        lower = np.array([-5., -9., 5., 3.])[:d]   # hard boundaries: most jumps leave the support, whole proposal sets are impossible
        upper = np.array([10., 2., 7., 8.])[:d]    # (log prior -inf) and the reference draws them again (Dream.py:281-289)
        params = [SampledParam(uniform, loc=lower, scale=upper - lower)]
        Z0 = lower + (3 * rng.uniform(0, 1, (nseed, d)) - 1) * (upper - lower)
        Z0[:N] = lower + rng.uniform(0, 1, (N, d)) * (upper - lower)            # the starts lie inside
        pk, pa, pb = np.full(d, 2, np.int32), lower, upper - lower
    elif prior == "normal":            # pydream/tests/test_models.py:24-33
        mu = np.resize(np.array([-6.6, 3, 1.0, -.12]), d)
        sd = np.resize(np.array([.13, 5, .9, 1.0]), d)
        params = [SampledParam(norm, loc=mu, scale=sd)]
        Z0 = mu + sd * rng.standard_normal((nseed, d))
        pk, pa, pb = np.full(d, 1, np.int32), mu, sd
    if target[0] == "mvn":
        invC, log_F, like = mvn_target(d)
        extra.update(lk_kind="mvn", invC=invC, log_F=log_F)
    elif target[0] == "mix":
        mu_m, lF, like = mixture_target(d, target[1], target[2])
        extra.update(lk_kind="mix", mix_mu=mu_m, mix_logF=lF)
    else:
        like = simple_likelihood
        extra.update(lk_kind="simple")
    starts = Z0[:N].copy()
    mt = k if k > 1 else False
    with tempfile.TemporaryDirectory() as wd:
        cwd = os.getcwd()
        os.chdir(wd)
        try:
            if restart_from is not None:
                prev = np.load(os.path.join(HERE, restart_from + ".npz"))
                Z0 = np.concatenate([prev["Z0"], prev["Z_tail"]])           # what save_history_to_disc writes (Dream.py:947-959)
                starts = prev["X"][-1].copy()
                np.save(os.path.join(wd, "prev_crossoverprob.npy"), prev["cross_probs"][-1])      # Dream.py:961-964
                dream_kwargs["crossover_file"] = os.path.join(wd, "prev_crossoverprob.npy")
                extra.update(restart_cr_probs=prev["cross_probs"][-1])
            out = run_reference(params, like, Z0, starts, N, G, seed, schedule, dict(multitry=mt, **dream_kwargs), wd, history_lag=history_lag, adapt_lag=adapt_lag)
            dream_kwargs.pop("crossover_file", None)
        finally:
            os.chdir(cwd)
    kw = dict(nCR=3, adapt_crossover=True, adapt_gamma=False, DEpairs=1, lamb=.05, zeta=1e-12, history_thin=10,
              snooker=.10, p_gamma_unity=.20, gamma_levels=1, hardboundaries=True)
    kw.update(dream_kwargs)
    cfgarr = {("cfg_" + a): np.asarray(b) for a, b in {**cfg, **kw}.items() if b is not None}
    save(name, Z0=Z0, starts=starts, prior_kind=pk, prior_a=pa, prior_b=pb, **cfgarr, **extra, **out)
    print("   ", name, "acc", out["moved"].mean(), "snk", out["snooker"].mean(), "cross_probs", out["cross_probs"][-1])


def function_cases():
    """Function-level vectors: generate_proposal_points / snooker_update on hand-built shared
    arrays (the reference tests' own style, pydream/tests/test_dream.py:207-211)."""
    import multiprocessing as mp
    from scipy.stats import uniform
    cases = {}
    rng = np.random.default_rng(7)
    for tag, d, k, bounded in (("d100k5", 100, 5, False), ("d4k5b", 4, 5, True), ("d4k1b", 4, 1, True), ("d10k1", 10, 1, False)):
        if bounded:
            lower = np.array([-5., -9., 5., 3.]); upper = np.array([10., 2., 7., 8.])
            params = [SampledParam(uniform, loc=lower, scale=upper - lower)]
            Z = lower + rng.uniform(0, 1, (40, d)) * (upper - lower)
            q0 = lower + rng.uniform(0, 1, d) * (upper - lower)
        else:
            params = [FlatParam(test_value=np.zeros(d))]
            Z = rng.uniform(-5, 15, (300, d)); q0 = rng.uniform(-5, 15, d)
        model = Model(likelihood=simple_likelihood, sampled_parameters=params)
        dream = RD.Dream(model=model, multitry=(k if k > 1 else False), DEpairs=2 if tag == "d10k1" else 1,
                         gamma_levels=2 if tag == "d10k1" else 1, lamb=(0.6 if bounded else .05))
        dream.nseedchains = len(Z)
        SV.history = mp.Array('d', list(Z.flatten()))
        SV.count = mp.Value('i', 0)
        seed = 1234
        rnd = ContractRandom(seed, k)
        install(rnd)
        try:
            recs = []
            for trial in range(12):
                snk = trial % 3 == 2
                cr_idx = trial % 3
                delta = 2 if (tag == "d10k1" and trial % 2) else 1
                glev = 2 if (tag == "d10k1" and trial % 4 > 1) else 1
                for phase in ((0, 1) if k > 1 else (0,)):
                    n = k if phase == 0 else k - 1
                    rnd.begin_step(3, trial, dream.p_gamma_unity)
                    rnd.n_ctrl = 99; rnd.phase = phase; rnd.snk = snk
                    CR = dream.CR_values[cr_idx]
                    res = dream.generate_proposal_points(n, q0.copy(), CR, delta, glev, snooker=snk)
                    if snk:
                        pts, slogp, z = res
                        pts = np.array(pts, dtype=float).reshape(n, d); slogp = np.atleast_1d(np.array(slogp, float))
                    else:
                        pts = np.array(res, dtype=float).reshape(n, d); slogp = np.zeros(n)
                    gam = np.atleast_1d(np.array(dream.gamma, float))
                    gam = np.resize(gam, n)
                    recs.append((trial, phase, int(snk), cr_idx, delta, glev, pts, slogp, gam))
        finally:
            uninstall()
        cases[tag] = dict(Z=Z, q0=q0, d=d, k=k, seed=seed, bounded=int(bounded), lamb=dream.lamb,
                          depairs=len(dream.DEpairs), ngamma=dream.ngamma,
                          mins=np.asarray(dream.mins, float), maxs=np.asarray(dream.maxs, float),
                          meta=np.array([r[:6] for r in recs]),
                          pts=np.concatenate([r[6].reshape(-1) for r in recs]),
                          slogp=np.concatenate([r[7] for r in recs]),
                          gam=np.concatenate([r[8] for r in recs]))
    flat = {}
    for tag, c in cases.items():
        for a, b in c.items():
            flat[tag + "__" + a] = np.asarray(b)
    save("proposals", **flat)


def density_cases():
    rng = np.random.default_rng(11)
    out = {}
    for d in (10, 100):
        invC, log_F, like = mvn_target(d)
        X = rng.normal(0, 3, (32, d))
        out["mvn%d_invC" % d] = invC; out["mvn%d_logF" % d] = log_F
        out["mvn%d_X" % d] = X; out["mvn%d_logp" % d] = np.array([like(x) for x in X])
    # the shipped examples themselves (d=200 log_F=0 branch; 2-component mixture, d=10)
    with tempfile.TemporaryDirectory() as wd:
        cwd = os.getcwd(); os.chdir(wd)
        try:
            from pydream.examples.ndim_gaussian import dream_ex_ndim_gaussian as EX
            from pydream.examples.mixturemodel import mixturemodel as MX
        finally:
            os.chdir(cwd)
    X = rng.normal(0, 3, (16, EX.d))
    out["mvn200_invC"] = EX.invC; out["mvn200_logF"] = float(EX.log_F); out["mvn200_X"] = X
    out["mvn200_logp"] = np.array([EX.likelihood(x) for x in X])
    X = np.concatenate([rng.normal(-5, 1, (8, MX.d)), rng.normal(5, 1, (8, MX.d)), rng.normal(0, 4, (8, MX.d))])
    out["mix2_mu"] = MX.mu; out["mix2_logF"] = MX.log_F; out["mix2_X"] = X
    out["mix2_logp"] = np.array([MX.likelihood(x) for x in X])
    mu3, lF3, like3 = mixture_target(100, (-5, 0, 5), (1 / 6., 1 / 3., 1 / 2.))
    X = np.concatenate([rng.normal(m, 1, (8, 100)) for m in (-5, 0, 5)])
    out["mix3_mu"] = mu3; out["mix3_logF"] = lF3; out["mix3_X"] = X
    out["mix3_logp"] = np.array([like3(x) for x in X])
    # gamma table (Dream.py:172-179; pinned by pydream/tests/test_dream.py:68-76)
    m = Model(likelihood=simple_likelihood, sampled_parameters=[FlatParam(test_value=np.zeros(7))])
    out["gamma_arr_7_5_4"] = RD.Dream(model=m, DEpairs=5, gamma_levels=4).gamma_arr
    # Gelman-Rubin (convergence.py:3-20)
    tr = rng.normal(0, 1, (5, 101, 6)).cumsum(axis=1) * 0.05 + rng.normal(0, 1, (5, 1, 6))
    out["gr_traces"] = tr; out["gr_rhat"] = Gelman_Rubin([tr[c] for c in range(5)])
    # mt_choose_proposal_pt known answer (pydream/tests/test_dream.py:345-355) + metrop_select
    save("densities", **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "pt":
        pt_case("trace_pt_mvn10", d=10, N=6, G=150, k=5, seed=23)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "lag":
        lag_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "adaptlag":
        adapt_lag_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "s1gamma":
        s1_gamma_case()
        return
    function_cases()
    density_cases()
    # T1: the C1 plumbing config (3 chains, 10-D MVN, multitry 5), unmodified reference, schedule S1
    trace_case("trace_s1_c1", d=10, N=3, G=200, k=5, schedule=1, seed=20260929, target=("mvn",),
               dream_kwargs=dict(adapt_crossover=False, crossover_burnin=10 ** 9))
    # T1b: S1 with crossover adaptation and the burn-in hand-over inside the run
    trace_case("trace_s1_adapt", d=10, N=4, G=120, k=5, schedule=1, seed=7, target=("mvn",),
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=60))
    # T2: lockstep S2 with adaptation, burn-in inside the run
    trace_case("trace_s2_adapt", d=10, N=4, G=120, k=5, schedule=2, seed=11, target=("mvn",),
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=40))
    # T3: single-try, uniform prior + hard boundaries (pydream/tests/test_models.py:35-50)
    # (100 generations: the reference sums the snooker projection in BLAS order, a 1e-15 difference per snooker step that the
    #  coupled chains amplify from generation to generation -- 4e-10 on log p after 125 generations at snooker = 0.3)
    trace_case("trace_s2_k1_bounds", d=4, N=5, G=100, k=1, schedule=2, seed=3, target=("simple",), prior="uniform",
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=50, snooker=.3))
    # T3b: multi-try with uniform prior + boundaries
    trace_case("trace_s2_k3_bounds", d=4, N=5, G=100, k=3, schedule=2, seed=5, target=("simple",), prior="uniform",
               dream_kwargs=dict(adapt_crossover=False, crossover_burnin=10 ** 9, lamb=.3))
    # T3c/T3d: multi-try, uniform prior WITHOUT hard boundaries and an archive wider than the support: proposal sets whose tries are
    # all impossible are drawn again (Dream.py:281-289); host likelihood (T3c) and the device MVN likelihood with snooker moves (T3d)
    trace_case("trace_s2_k3_redraw", d=4, N=5, G=100, k=3, schedule=2, seed=31, target=("simple",), prior="uniform_wide_history",
               dream_kwargs=dict(adapt_crossover=False, crossover_burnin=10 ** 9, hardboundaries=False))
    trace_case("trace_s2_k5_redraw_mvn", d=4, N=6, G=100, k=5, schedule=2, seed=37, target=("mvn",), prior="uniform_wide_history",
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=40, hardboundaries=False, snooker=.3))
    # T4: DEpairs=2, 3 gamma levels with gamma adaptation, normal prior
    trace_case("trace_s2_depairs_gamma", d=6, N=6, G=120, k=5, schedule=2, seed=13, target=("mvn",), prior="normal",
               dream_kwargs=dict(adapt_crossover=True, adapt_gamma=True, gamma_levels=3, DEpairs=2, crossover_burnin=50))
    # T5: C2-shaped, small: 100-D MVN, 8 chains
    trace_case("trace_s2_mvn100", d=100, N=8, G=40, k=5, schedule=2, seed=20260929, target=("mvn",), nseed=160,
               dream_kwargs=dict(adapt_crossover=False, crossover_burnin=10 ** 9))
    # T6: C3-shaped, small: 3-component mixture with crossover adaptation
    trace_case("trace_s2_mix3", d=20, N=8, G=80, k=5, schedule=2, seed=17, target=("mix", (-5, 0, 5), (1 / 6., 1 / 3., 1 / 2.)),
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=30))
    # T7: parallel tempering (core.py:131-248): temperature ladder, one swap attempt per iteration
    pt_case("trace_pt_mvn10", d=10, N=6, G=150, k=5, seed=23)
    # T7b: parallel tempering WITH crossover adaptation (run_dream's default), burn-in inside the run: after an accepted swap
    # the next jump is measured from the swapped state (Dream.py:371-378)
    pt_case("trace_pt_adapt", d=10, N=6, G=100, k=5, seed=29, dream_kwargs=dict(adapt_crossover=True, crossover_burnin=60))
    # T8: restart (core.py:46-62, 255-263; Dream.py:128-141): continues trace_s2_adapt from its history and adapted
    # crossover probabilities, adaptation on again
    trace_case("trace_s2_restart", d=10, N=4, G=60, k=5, schedule=2, seed=12, target=("mvn",), restart_from="trace_s2_adapt",
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=30))
    lag_cases()
    adapt_lag_cases()
    s1_gamma_case()


def adapt_lag_cases():
    # T10 (round 6): adapt_lag -- lockstep S2 whose crossover / gamma-level updates reach the chains' decisions L generations late (the
    # reference's own estimate_* methods, replayed late on the positions of their own generation), the burn-in ending inside the run:
    # L = 1; L = 9 with history_lag = 1 on the 3-component mixture (the schedule bench.py times configs[2] under: whole thin-cycles
    # per launch inside the burn-in); L = 3 with gamma-level adaptation, DEpairs = 2 and a normal prior
    trace_case("trace_s2_adaptlag1", d=10, N=8, G=130, k=5, schedule=2, seed=51, target=("mvn",), adapt_lag=1,
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=80))
    trace_case("trace_s2_adaptlag9_mix", d=20, N=8, G=100, k=5, schedule=2, seed=53, target=("mix", (-5, 0, 5), (1 / 6., 1 / 3., 1 / 2.)),
               adapt_lag=9, history_lag=1, dream_kwargs=dict(adapt_crossover=True, crossover_burnin=55))
    trace_case("trace_s2_adaptlag3_gamma", d=6, N=6, G=110, k=5, schedule=2, seed=57, target=("mvn",), prior="normal", adapt_lag=3,
               dream_kwargs=dict(adapt_crossover=True, adapt_gamma=True, gamma_levels=3, DEpairs=2, crossover_burnin=50))


def s1_gamma_case():
    # T1c: S1 with crossover AND gamma-level adaptation (3 levels, DEpairs = 2): every Dream instance keeps its own copy of both
    # probability vectors between its own updates (Dream.py:375, :383, :409-415)
    trace_case("trace_s1_adapt_gamma", d=6, N=5, G=90, k=5, schedule=1, seed=19, target=("mvn",),
               dream_kwargs=dict(adapt_crossover=True, adapt_gamma=True, gamma_levels=3, DEpairs=2, crossover_burnin=45))


def lag_cases():
    # T9: history_lag = 1 (and 2): lockstep S2 whose appended rows become sampleable one (two) appends late -- the schedule under which
    # the exchange of appended rows between GPUs hides behind the next thin-cycle.  Appends every 5 generations (24 of them), the
    # crossover burn-in ends inside the run; the snooker share is raised so that the lagged archive length shows in many row draws.
    trace_case("trace_s2_lag1", d=10, N=4, G=120, k=5, schedule=2, seed=41, target=("mvn",), history_lag=1,
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=40, history_thin=5, snooker=.2))
    trace_case("trace_s2_lag2_k1", d=6, N=5, G=90, k=1, schedule=2, seed=43, target=("mvn",), history_lag=2,
               dream_kwargs=dict(adapt_crossover=False, crossover_burnin=10 ** 9, history_thin=3))
    # (round 6) history_lag = 3: what bench.py runs at every N from round 6 on -- two appends per launch of the persistent kernels on several GPUs
    # too, the copy engines' pushes of a launch's rows hidden behind the whole next launch (include/dreamzs.h dz_config.history_lag)
    trace_case("trace_s2_lag3", d=10, N=5, G=120, k=5, schedule=2, seed=47, target=("mvn",), history_lag=3,
               dream_kwargs=dict(adapt_crossover=True, crossover_burnin=30, history_thin=4, snooker=.2))


if __name__ == "__main__":
    main()
