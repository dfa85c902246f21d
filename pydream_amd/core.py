# -*- coding: utf-8 -*-
"""run_dream -- same signature and return convention as pydream/core.py:11-86, executed on the GPU."""
import os
import time
from datetime import datetime

import numpy as np

from . import Dream_shared_vars
from . import _capi
from .Dream import Dream
from .model import Model


def run_dream(parameters, likelihood, nchains=5, niterations=50000, start=None, restart=False, verbose=True, nverbose=10, tempering=False, mp_context=None, **kwargs):
    """Run DREAM given a set of parameters with priors and a likelihood function (pydream/core.py:11-44).

    Parameters
    ----------
    parameters: iterable of SampledParam class
    likelihood: function ``f(vec[d]) -> float`` -- or one of ``pydream_amd.likelihoods`` to evaluate it on the device
    nchains, niterations, start, restart, verbose: as in the reference
    nverbose: with verbose on, progress (acceptance rates over all chains) is printed every ``nverbose`` iterations
        rounded up to a multiple that is at least 1000 iterations -- one line per device chunk
    tempering: parallel tempering as in the reference (core.py:131-236): the ladder T_i = 0.001**(i/nchains), one
        temperature-swap attempt per iteration; returns arrays of shape (nchains, 2*niterations, d) and
        (nchains, 2*niterations, 1) with the samples before and after each swap attempt interleaved
    mp_context: accepted for compatibility, ignored (there are no worker processes)
    kwargs: passed to Dream (see Dream).  Extra keys understood here: ``seed`` (int, key of the random
        contract; default drawn from the OS), ``device`` (HIP device ordinal), ``history_lag`` (int, default 0: the rows a
        generation appends to the history are sampled from the next generation on; L >= 1: L appends later -- fewer and
        longer kernel launches, a few percent faster with a device likelihood; see include/dreamzs.h dz_config.history_lag),
        ``adapt_lag`` (int, default 0: every generation of the crossover burn-in decides with the probabilities all earlier generations'
        updates left -- the reference in lockstep; L >= 1: with those of generations <= g - 1 - L, which lets one kernel launch hold
        L + 1 burn-in generations -- the burn-in then runs nearly as fast as the rest of the run; pays wherever the persistent kernels
        run (device likelihoods, d <= 256; sharded: ranks of whole groups of 256 chains), no gain elsewhere; dz_config.adapt_lag).

    Returns
    -------
    sampled_params : list of arrays [niterations, d], one per chain
    log_ps : list of arrays [niterations, 1], log probability of every sampled point
    """
    if restart:
        if start is None:
            raise Exception('Restart run specified but no start positions given.')
        if 'model_name' not in kwargs:
            raise Exception('Restart run specified but no model name to load history and crossover value files from given.')
    if type(parameters) is not list:
        parameters = [parameters]

    model = Model(likelihood=likelihood, sampled_parameters=parameters)

    if restart:
        step_instance = Dream(model=model, variables=parameters,
                              history_file=kwargs['model_name'] + '_DREAM_chain_history.npy',
                              crossover_file=kwargs['model_name'] + '_DREAM_chain_adapted_crossoverprob.npy',
                              gamma_file=kwargs['model_name'] + '_DREAM_chain_adapted_gammalevelprob.npy',
                              verbose=verbose, mp_context=mp_context, **kwargs)
    else:
        step_instance = Dream(model=model, variables=parameters, verbose=verbose, mp_context=mp_context, **kwargs)

    # The examples' convergence loop (dream_ex_ndim_gaussian.py:79-102) calls run_dream(restart=True, model_name=...) again and again:
    # every call would rebuild the engine and upload the whole -- growing -- history from the .npy file the call before has just
    # written.  The engine of a run that saved its history under a model name stays alive instead ("parked"), and a restart under the same
    # name with the same sampler and untouched files continues on it (dz_continue_run): history and adapted probabilities are already
    # in HBM.  The files are still written.  release_engines() frees a parked engine's memory; DREAMZS_KEEP_ENGINE=0 turns this off.
    live = _claim_parked(step_instance, kwargs, nchains, niterations, likelihood, tempering) if restart else None
    # Whatever this call is not going to continue on is released BEFORE the new engine allocates (advisor, round 4): a parked engine holds
    # its whole archive in HBM, and a run under another name (or none, or with save_history off) would otherwise allocate beside it.
    release_engines()
    try:
        pool = _setup_mp_dream_pool(nchains, niterations, step_instance, start_pt=start, mp_context=mp_context,
                                    seed=kwargs.get('seed'), device=kwargs.get('device', 0), history_lag=kwargs.get('history_lag', 0),
                                    adapt_lag=kwargs.get('adapt_lag', 0), live=live)
    except BaseException:
        if live is not None:              # (claimed out of the parked table: nobody else would close it -- bad nchains / start shape, an allocation failing in continue_run)
            live.close()
        raise
    park = False
    try:
        pool._initializer(*pool._initargs)
        if tempering:
            sampled_params, log_ps = _sample_dream_pt_batched(pool.engine, step_instance, nchains, niterations, verbose)
        else:
            sampled_params, log_ps = _sample_dream_batched(pool.engine, step_instance, niterations, verbose, nverbose)
        park = (not tempering and step_instance.save_history and bool(step_instance.model_name) and pool.engine.nl == nchains
                and os.environ.get("DREAMZS_KEEP_ENGINE", "1") != "0" and hasattr(pool.engine, "continue_run"))
    finally:
        pool.close()
        if park:
            _park(pool, step_instance, kwargs, nchains, likelihood)
        else:
            pool.join()
    return sampled_params, log_ps


last_kernel_variant = None      # which kernel instantiation carried the generations of the last run_dream call (dz_last_kernel_variant), for the curious


# ---- the engine kept alive between run_dream calls (the one of the last run that saved its history under a model name) ----
_parked = {}


def _signature(step, kwargs, nchains, likelihood, engine):
    """everything a continued run must have in common with the run whose engine it takes over"""
    dev_prior = step.model.device_prior()
    pri = None if dev_prior is None else tuple(np.asarray(a).tobytes() for a in dev_prior)
    return (nchains, step.total_var_dimension, int(step.multitry), len(step.DEpairs), int(step.nCR), int(step.ngamma), int(step.history_thin),
            bool(step.adapt_crossover), bool(step.adapt_gamma), bool(step.boundaries), float(step.lamb), float(step.zeta), float(step.snooker),
            float(step.p_gamma_unity), int(kwargs.get('device', 0)), int(kwargs.get('history_lag', 0)), int(kwargs.get('adapt_lag', 0)), bool(getattr(step, 'parallel', False)),
            np.asarray(step.mins).tobytes() if step.boundaries else None, np.asarray(step.maxs).tobytes() if step.boundaries else None,
            np.asarray(step.gamma_arr).tobytes(), pri, id(likelihood))


def _file_stamps(prefix):
    out = []
    for tail in ('DREAM_chain_history.npy', 'DREAM_chain_adapted_crossoverprob.npy', 'DREAM_chain_adapted_gammalevelprob.npy'):
        st = os.stat(prefix + '_' + tail)
        out.append((st.st_mtime_ns, st.st_size, st.st_ino))
    return out


def _park(pool, step, kwargs, nchains, likelihood):
    name = step.model_name
    old = _parked.pop(name, None)
    if old is not None and old["engine"] is not pool.engine:
        old["engine"].close()
    while _parked:                          # ONE parked engine per process: a parked archive is gigabytes of HBM that only a restart under
        _parked.popitem()[1]["engine"].close()      # its own model name would use again
    if Dream_shared_vars.engine is pool.engine:
        Dream_shared_vars.engine = None
    try:
        _parked[name] = dict(engine=pool.engine, sig=_signature(step, kwargs, nchains, likelihood, pool.engine), files=_file_stamps(name),
                             likelihood=likelihood)          # (the reference keeps id(likelihood) meaningful)
        pool.engine = None
    except OSError:
        pool.join()


def _claim_parked(step, kwargs, nchains, niterations, likelihood, tempering):
    """the parked engine of this model name if the restart may continue on it, else None (and the parked engine is released)"""
    name = kwargs.get('model_name')
    ent = _parked.pop(name, None)
    if ent is None:
        return None
    try:
        ok = (not tempering and os.environ.get("DREAMZS_KEEP_ENGINE", "1") != "0" and ent["files"] == _file_stamps(name)
              and ent["sig"] == _signature(step, kwargs, nchains, likelihood, ent["engine"]))
    except OSError:
        ok = False
    if not ok:
        ent["engine"].close()
        return None
    return ent["engine"]


def release_engines():
    """free the engine kept for run_dream(restart=True) (its archive stays in HBM until then, or until the process ends)"""
    while _parked:
        _parked.popitem()[1]["engine"].close()


import atexit                                   # noqa: E402
atexit.register(release_engines)                # (a parked engine is destroyed while the HIP runtime is still up, not by a late __del__)


def _sample_dream_pt_batched(eng, step, nchains, niterations, verbose):
    """core._sample_dream_pt (core.py:131-236) on the engine: every chain steps at its own temperature
    (Dream.astep's T), then one random pair attempts a temperature swap; sampled_params[:, 2i] holds the states after
    the steps of iteration i, [:, 2i+1] the states after its swap attempt (log_ps likewise, T*loglike + logprior)."""
    if eng.nl != nchains:
        raise Exception('parallel tempering needs all chains on one GPU')
    d = step.total_var_dimension
    T = np.zeros((nchains))
    T[0] = 1.
    for i in range(nchains):
        T[i] = np.power(.001, (float(i) / nchains))                      # core.py:133-136
    eng.set_temperatures(T, swaps=True)
    sampled_params = np.zeros((nchains, niterations * 2, d))
    log_ps = np.zeros((nchains, niterations * 2, 1))
    chunk = eng.cfg.trace_capacity
    done = 0
    while done < niterations:
        n = min(chunk, niterations - done)
        eng.trace_reset()
        eng.step(n)
        tr = eng.get_trace(0, n)
        sw = eng.get_swaps(0, n)
        X = tr["X"].transpose(1, 0, 2)                                   # [chain, iteration, d], before the swaps
        L = tr["logp"].T
        rows = slice(2 * done, 2 * (done + n), 2)
        sampled_params[:, rows] = X
        log_ps[:, rows, 0] = L
        Xs, Ls = X.copy(), L.copy()
        for i in np.nonzero(sw[:, 2])[0]:                                # accepted swaps exchange the pair (core.py:204-215)
            a, b = sw[i, 0], sw[i, 1]
            Xs[[a, b], i] = X[[b, a], i]
            Ls[[a, b], i] = L[[b, a], i]
        rows = slice(2 * done + 1, 2 * (done + n), 2)
        sampled_params[:, rows] = Xs
        log_ps[:, rows, 0] = Ls
        done += n
        if verbose:
            print('Iteration: ', done, ' overall temp swap acceptance rate: ', float(sw[:, 2].mean()))
    if step.save_history:
        _save_history_to_disc(eng, step)
    return sampled_params, log_ps


def _sample_dream_batched(eng, step, niterations, verbose, nverbose):
    """The per-chain loop of core._sample_dream (core.py:89-129) for all chains at once: advance the
    engine in chunks that fit the device trace buffer and collect trace / log_ps per chain."""
    d = step.total_var_dimension
    nchains = eng.nl                                  # the chains this engine owns (all of them on one GPU)
    chunk = eng.cfg.trace_capacity
    if verbose and nverbose:
        # progress lines come per device chunk: with verbose on, a chunk is the smallest multiple of nverbose that is
        # at least 1000 iterations (the reference prints every nverbose iterations of every chain, core.py:118-126;
        # a line per 10 iterations would cost a device round trip each)
        step_it = int(nverbose) * max(1, -(-1000 // int(nverbose)))
        chunk = max(1, min(chunk, step_it))
    by_chain = hasattr(eng, "get_trace_chains")       # the HIP engine keeps the trace chain by chain: no host transposition
    if by_chain:
        S = np.empty((nchains, niterations, d))
        LP = np.empty((nchains, niterations, 1))
        sampled = [S[c] for c in range(nchains)]      # one (niterations, d) array per chain, as core.py:98/:127 return them
        log_ps = [LP[c] for c in range(nchains)]
    else:
        sampled = [np.empty((niterations, d)) for _ in range(nchains)]
        log_ps = [np.empty((niterations, 1)) for _ in range(nchains)]
    done = 0
    naccepts = 0
    pin = None
    if by_chain and S.nbytes >= (64 << 20):
        # fault in and page-lock the result array on a second thread while the GPU runs the first chunk
        import threading
        pinned = []
        pin = threading.Thread(target=lambda: pinned.append(eng.host_register(S)))
        pin.start()
    nseg = int(os.environ.get("DREAMZS_DOWNLOAD_SEGMENTS", "8"))       # 0: one download after the run
    if pin is not None and nseg > 0 and not verbose and chunk >= niterations and S.nbytes >= (256 << 20) and hasattr(eng, "trace_download_begin"):
        # large quiet runs whose trace fits the device: the run goes in a few segments, and each segment's samples leave for the host
        # (one strided DMA on a stream of its own) while the next segment's generations run
        eng.trace_reset()
        seg = -(-niterations // nseg)
        ok = None
        while done < niterations:
            n = min(seg, niterations - done)
            eng.step(n)
            if ok is None:
                pin.join()                                   # the destination has to be page-locked before the first copy is queued
                ok = bool(pinned and pinned[0])
            if ok:
                eng.trace_download_begin(done, n, S, row0=done)
            done += n
        if ok:
            eng.trace_download_wait()
        eng.get_trace_chains(0, niterations, None if ok else S, logp_out=LP)
    while done < niterations:
        n = min(chunk, niterations - done)
        eng.trace_reset()
        eng.step(n)
        if by_chain:
            if pin is not None and pin.is_alive():
                pin.join()
            eng.get_trace_chains(0, n, S, row0=done, logp_out=LP)
            tr = eng.get_trace(0, n, with_X=False, with_logp=False) if verbose else None     # (the decision flags only feed the progress line)
        else:
            tr = eng.get_trace(0, n)
            for c in range(nchains):
                sampled[c][done:done + n] = tr["X"][:, c, :]
                log_ps[c][done:done + n, 0] = tr["logp"][:, c]
        if tr is not None:
            naccepts += int(tr["moved"].sum())
        done += n
        if verbose:
            print('Iteration: ', done, ' acceptance rate: ', naccepts / float(done * nchains),
                  ' acceptance rate over last %d iterations: ' % n, float(tr["moved"].mean()))
    if pin is not None:
        pin.join()
        if pinned and pinned[0]:
            eng.host_unregister(S)
    from .convergence import SampledList
    sampled = SampledList(sampled)
    if chunk >= niterations >= 2 and eng.nl == eng.N and hasattr(eng, "get_rhat"):
        # the whole run is still in the device trace: the reference's diagnostic (convergence.py:3-20) over all chains, made there
        sampled.gelman_rubin = eng.get_rhat()
    global last_kernel_variant
    last_kernel_variant = eng.last_kernel_variant() if hasattr(eng, "last_kernel_variant") else None
    if step.save_history:
        _save_history_to_disc(eng, step, appended_from=getattr(eng, "continued_from_file_rows", None))
    return sampled, log_ps


_NPY_HEADER_BYTES = 256          # room for any row count: the shape can be rewritten in place when rows are appended


def _npy_header(nvalues):
    """a version-1.0 .npy header for a flat float64 array of `nvalues`, padded to _NPY_HEADER_BYTES (np.load reads any valid length)"""
    body = "{'descr': '<f8', 'fortran_order': False, 'shape': (%d,), }" % nvalues
    pad = _NPY_HEADER_BYTES - 10 - len(body) - 1
    return b"\x93NUMPY\x01\x00" + (_NPY_HEADER_BYTES - 10).to_bytes(2, "little") + (body + " " * pad + "\n").encode("latin1")


def _write_history_file(filename, eng, d, appended_from=None):
    """Dream.save_history_to_disc's history file (Dream.py:947-959: the WHOLE flat history, every time).  A run continued on the live
    engine of the run that wrote `filename` (run_dream checked that the file is untouched since) appends its new rows behind the old
    ones and rewrites the row count in the header instead of downloading and writing gigabytes that are already there."""
    rows = eng.history_rows()
    if appended_from is not None and 0 < appended_from <= rows:
        try:
            with open(filename, "r+b") as f:
                if f.read(_NPY_HEADER_BYTES) == _npy_header(appended_from * d) and os.fstat(f.fileno()).st_size == _NPY_HEADER_BYTES + 8 * appended_from * d:
                    f.seek(0, 2)
                    eng.get_history(appended_from).tofile(f)
                    f.seek(0)
                    f.write(_npy_header(rows * d))
                    return
        except OSError:
            pass
    with open(filename, "wb") as f:
        f.write(_npy_header(rows * d))
        eng.get_history().tofile(f)


def _save_history_to_disc(eng, step, appended_from=None):
    """Dream.save_history_to_disc (Dream.py:947-969): the three .npy files a restart reads back."""
    if not step.model_name:
        prefix = datetime.now().strftime('%Y_%m_%d_%H:%M:%S') + '_'
    else:
        prefix = step.model_name + '_'
    filename = prefix + 'DREAM_chain_history.npy'
    if step.verbose:
        print('Saving history to file: ', filename)
    if isinstance(eng, _capi.Engine):
        _write_history_file(filename, eng, step.total_var_dimension, appended_from)
    else:
        np.save(filename, eng.get_history().reshape(-1))
    cr = eng.get_cr_state()[0]
    filename = prefix + 'DREAM_chain_adapted_crossoverprob.npy'
    if step.verbose:
        print('Saving fitted crossover values: ', cr, ' to file: ', filename)
    np.save(filename, cr)
    gp = eng.get_gamma_state()[0]
    filename = prefix + 'DREAM_chain_adapted_gammalevelprob.npy'
    if step.verbose:
        print('Saving fitted gamma level values: ', gp, ' to file: ', filename)
    np.save(filename, gp)


class _EnginePool:
    """Stands in for the DreamPool that core._setup_mp_dream_pool returns (core.py:307-314): the
    "workers" are the GPU; ``_initializer(*_initargs)`` publishes the engine the way _mp_dream_init
    publishes the shared arrays (core.py:316-327); ``close``/``join`` release it."""

    def __init__(self, engine, nchains, host_eval=None):
        self.engine = engine
        self.host_eval = host_eval
        self._initializer = _mp_dream_init
        self._initargs = (engine, nchains)

    def close(self):
        if self.host_eval is not None:
            self.host_eval.close()

    def join(self):
        if Dream_shared_vars.engine is self.engine:
            Dream_shared_vars.engine = None
        if self.engine is not None:
            self.engine.close()
            self.engine = None


def _mp_dream_init(engine, nchains):
    Dream_shared_vars.engine = engine
    Dream_shared_vars.nchains_counter = nchains
    Dream_shared_vars.host_state = {}
    Dream_shared_vars.temperatures = None


def _setup_mp_dream_pool(nchains, niterations, step_instance, start_pt=None, mp_context=None, seed=None, device=0,
                         chain_offset=0, nchains_local=None, engine_cls=None, history_lag=0, live=None, adapt_lag=0):
    """Validate, size and allocate the shared sampler state (core.py:250-314) -- in HBM.
    `live`: the engine of the run whose files this restart would load (run_dream): continued in place, nothing uploaded."""
    min_njobs = (2 * len(step_instance.DEpairs)) + 1
    if nchains < min_njobs:
        raise Exception('Dream should be run with at least (2*DEpairs)+1 number of chains.  For current algorithmic settings, set njobs>=%s.' % str(min_njobs))
    d = step_instance.total_var_dimension
    if seed is None:
        seed = int.from_bytes(os.urandom(8), 'little')
    rng = np.random.RandomState(seed % (2 ** 32))
    Dream_shared_vars.rng = rng
    nrows_live = live.history_rows() if live is not None else None
    if live is not None:                              # the history file's content is the live engine's archive
        seed_rows = None
        step_instance.nseedchains = nrows_live
    elif step_instance.history_file != False:        # noqa: E712  (core.py:255-259)
        old_history = np.load(step_instance.history_file)
        seed_rows = np.asarray(old_history, dtype=float).reshape(-1, d)
        step_instance.nseedchains = len(seed_rows)
    else:
        seed_rows = None
    min_nseedchains = 2 * len(step_instance.DEpairs) * nchains
    if step_instance.nseedchains < min_nseedchains:
        raise Exception('The size of the seeded starting history is insufficient.  Increase nseedchains>=%s.' % str(min_nseedchains))
    if seed_rows is None and live is None:            # Dream.py:203-214: seed the history with draws from the prior
        seed_rows = np.array([Dream_shared_vars.draw_from_prior(step_instance.variables) for _ in range(int(step_instance.nseedchains))])
    if step_instance.crossover_burnin is None:        # core.py:299-300
        step_instance.crossover_burnin = int(np.floor(niterations / 10))
    if start_pt is not None:
        if step_instance.start_random:
            print('Warning: start position provided but random_start set to True.  Overrode random_start value and starting walk at provided start position.')
            step_instance.start_random = False

    nl = nchains if nchains_local is None else nchains_local
    thin = step_instance.history_thin
    n_appends = (niterations + thin - 1) // thin + 1
    ld = (d + 15) // 16 * 16
    # the device keeps whole chunks of the trace (up to 64 GiB of the 288 GB by default): normally the entire run
    trace_cap = int(max(1, min(niterations, int(os.environ.get('DREAMZS_TRACE_BYTES', 64 << 30)) // (nl * ld * 8))))
    if live is not None:
        live.continue_run(nrows_live + nchains * n_appends, trace_cap, int(seed), int(min(step_instance.crossover_burnin, 2 ** 31 - 1)))
        live.nseed = nrows_live
        live.continued_from_file_rows = nrows_live       # (the history file of the run before holds exactly these rows: run_dream checked)
    eng = live if live is not None else (engine_cls or _capi.Engine)(nchains=nchains, nchains_local=nl, chain_offset=chain_offset, ndim=d, multitry=int(step_instance.multitry),
                       depairs=len(step_instance.DEpairs), ncr=int(step_instance.nCR), ngamma=int(step_instance.ngamma),
                       history_thin=int(thin), crossover_burnin=int(min(step_instance.crossover_burnin, 2 ** 31 - 1)),
                       adapt_crossover=int(bool(step_instance.adapt_crossover)), adapt_gamma=int(bool(step_instance.adapt_gamma)),
                       hardboundaries=int(bool(step_instance.boundaries)), schedule=2, device=int(device), history_lag=int(history_lag), adapt_lag=int(adapt_lag),
                       history_capacity=len(seed_rows) + nchains * n_appends, trace_capacity=trace_cap, seed=int(seed),
                       lamb=float(step_instance.lamb), zeta=float(step_instance.zeta), snooker=float(step_instance.snooker),
                       p_gamma_unity=float(step_instance.p_gamma_unity))
    if live is None:
        eng.nseed = len(seed_rows)
        if step_instance.boundaries:
            eng.set_bounds(step_instance.mins, step_instance.maxs)
        eng.set_gamma_table(step_instance.gamma_arr)
        eng.set_history(seed_rows)
        eng.set_cr_probs(np.asarray(step_instance.CR_probabilities, dtype=float))
        eng.set_gamma_probs(np.asarray(step_instance.gamma_probabilities, dtype=float))

    model = step_instance.model
    dev_prior = model.device_prior() if step_instance.variables is model.sampled_parameters or list(step_instance.variables) == list(model.sampled_parameters) else None
    if dev_prior is not None and live is None:
        eng.set_prior(*dev_prior)
    like = model.likelihood
    host_eval = None
    if hasattr(like, "_dz_apply") and dev_prior is not None:
        like._dz_apply(eng)                           # likelihood AND priors on the device (also on a continued engine: the object may have been
                                                      # changed in place since the run before -- cheap next to the history upload that is saved)
    else:
        from .model import HostEvaluator
        host_eval = HostEvaluator(model, dev_prior is None, nchains, mp_context=mp_context, force=bool(getattr(step_instance, 'parallel', False)))
        eng.set_likelihood_host(host_eval)

    # start positions (core.py:74-78; Dream.py:221-225 for random starts)
    if start_pt is None:
        X0 = np.array([Dream_shared_vars.draw_from_prior(step_instance.variables) for _ in range(nchains)])
    elif type(start_pt) is list:
        X0 = np.array([np.asarray(s, dtype=float).reshape(-1) for s in start_pt])
    else:
        X0 = np.array([np.asarray(start_pt, dtype=float).reshape(-1)] * nchains)
    if len(X0) != nchains:
        raise Exception('start must be one vector or a list of nchains vectors')
    eng.set_state(X0[chain_offset:chain_offset + nl])
    return _EnginePool(eng, nchains, host_eval)
