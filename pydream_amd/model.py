"""Model: the likelihood/prior boundary -- same interface as pydream/model.py."""
import multiprocessing
import os
import time

import numpy as np

_worker_model = None          # set in the worker processes of HostEvaluator


def _worker_init(model):
    global _worker_model
    _worker_model = model


def _worker_eval(args):
    X, with_prior = args
    return _worker_model.batch_logp(X, with_prior)


class HostEvaluator:
    """Evaluates a Python likelihood for a batch of points, over worker processes when that pays.

    The reference runs every chain in its own process (core.py:250-314), so an expensive Python likelihood is
    evaluated nchains-fold in parallel there; here all chains' points arrive in one batch per pass, and the batch is
    split over `workers` processes.  Cheap likelihoods stay in-process: the pool is only started once a serial batch
    has taken longer than `threshold` seconds.  workers: DREAMZS_HOST_WORKERS, else min(nchains, usable cores);
    0 or 1 disables.  mp_context: as in the reference (core.py:11, :307-311); None = the platform default."""

    def __init__(self, model, with_prior, nchains, mp_context=None, force=False, threshold=2e-3):
        self.model, self.with_prior, self.threshold, self.force = model, with_prior, threshold, force
        env = os.environ.get("DREAMZS_HOST_WORKERS")
        if env is not None:
            self.workers = max(0, int(env))
        else:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            self.workers = min(int(nchains), cores)
        self.ctx = mp_context
        self.pool = None
        self.serial_time = 0.0

    def __call__(self, X):
        n = len(X)
        use_pool = self.workers > 1 and n >= 2 and (self.pool is not None or self.force or self.serial_time > self.threshold)
        if not use_pool:
            t0 = time.perf_counter()
            out = self.model.batch_logp(X, self.with_prior)
            self.serial_time = time.perf_counter() - t0
            return out
        if self.pool is None:
            ctx = multiprocessing.get_context(self.ctx) if isinstance(self.ctx, str) or self.ctx is None else self.ctx
            self.pool = ctx.Pool(self.workers, initializer=_worker_init, initargs=(self.model,))
        parts = np.array_split(np.ascontiguousarray(X), min(self.workers, n))
        res = self.pool.map(_worker_eval, [(part, self.with_prior) for part in parts])
        return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None


class Model():
    """The likelihood together with the priors it is sampled under (interface of pydream/model.py:8-32):
    ``Model(likelihood, sampled_parameters)``, attribute ``sampled_parameters`` (always a list), ``total_logp``."""

    def __init__(self, likelihood, sampled_parameters):
        self.likelihood = likelihood
        self.sampled_parameters = sampled_parameters if type(sampled_parameters) is list else [sampled_parameters]

    def total_logp(self, q0):
        """``(log prior, log likelihood)`` of one point.  The point is the concatenation of the parameters' values in
        order; each parameter's prior sees its own slice, the likelihood the whole vector (model.py:17-32)."""
        logprior = 0
        offset = 0
        for param in self.sampled_parameters:
            try:
                piece = q0[offset:offset + param.dsize]
            except IndexError:          # q0 is a bare scalar (one parameter of one dimension)
                piece = q0
            logprior += param.prior(piece)
            offset += param.dsize
        return logprior, self.likelihood(q0)

    # ---- batched host evaluation used by the engine's host-callback path ----
    def device_prior(self):
        """Concatenated (kind, a, b) if EVERY parameter has a device prior, else None."""
        parts = []
        for param in self.sampled_parameters:
            f = getattr(param, "device_prior", None)
            d = f() if f is not None else None
            if d is None:
                return None
            parts.append(d)
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

    def batch_logp(self, X, with_prior):
        """(prior[n], like[n]) for the rows of X; the prior column is zero when the device evaluates it."""
        n = len(X)
        pr, lk = np.zeros(n), np.zeros(n)
        for i in range(n):
            q = X[i]
            if with_prior:
                p = 0.0
                var_start = 0
                for param in self.sampled_parameters:
                    p += param.prior(q[var_start:var_start + param.dsize])
                    var_start += param.dsize
                pr[i] = p
            lk[i] = self.likelihood(q)
        return pr, lk
