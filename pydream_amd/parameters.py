"""Parameter priors -- the interface of pydream/parameters.py (SampledParam, FlatParam): ``dsize``, ``interval``,
``random``, ``prior``.

Additionally each parameter can describe itself to the device (``device_prior``) so that
``scipy.stats.norm`` / ``scipy.stats.uniform`` / flat priors are evaluated inside the HIP kernels
(``dz_set_prior``); any other distribution is evaluated on the host through ``.prior``.
"""
import numpy as np


class SampledParam():
    """A prior given by a frozen SciPy distribution (interface of pydream/parameters.py:6-47).

    ``SampledParam(scipy.stats.norm, loc=..., scale=...)``: the first argument is the SciPy continuous distribution
    class, everything after it is handed to that class.  The dimension of the parameter (``dsize``) is the size of
    one draw from the distribution."""

    def __init__(self, scipy_distribution, *args, **kwargs):
        self.dist = scipy_distribution(*args, **kwargs)
        self.dsize = np.size(self.random())

    def interval(self, alpha=1):
        """Central interval holding a fraction ``alpha`` of the prior mass; ``alpha = 1`` gives the support, which is
        what Dream uses as hard boundaries (parameters.py:23-26, Dream.py:86-105)."""
        return self.dist.interval(alpha)

    def random(self, reseed=False):
        """One draw from the prior.  ``reseed=True`` draws from a freshly seeded generator instead of numpy's
        global one (parameters.py:28-35: forked workers would otherwise all produce the same start)."""
        state = np.random.RandomState() if reseed else None
        return self.dist.rvs(random_state=state)

    def prior(self, q0):
        """Log prior density of the point ``q0``: the log pdf summed over the parameter's dimensions
        (parameters.py:37-47)."""
        return np.sum(self.dist.logpdf(q0))

    def device_prior(self):
        """(kind[d], a[d], b[d]) for dz_set_prior, or None if this distribution has no device form.
        kind 1 = norm(loc=a, scale=b), kind 2 = uniform(loc=a, scale=b)."""
        name = getattr(getattr(self.dist, "dist", None), "name", None)
        if name not in ("norm", "uniform"):
            return None
        try:
            shapes, loc, scale = self.dist.dist._parse_args(*self.dist.args, **self.dist.kwds)
        except Exception:
            return None
        loc = np.broadcast_to(np.asarray(loc, dtype=float), (self.dsize,)).copy()
        scale = np.broadcast_to(np.asarray(scale, dtype=float), (self.dsize,)).copy()
        kind = np.full(self.dsize, 1 if name == "norm" else 2, dtype=np.int32)
        return kind, loc, scale


class FlatParam(SampledParam):
    """An improper flat prior: log density 0 everywhere, unbounded support (interface of
    pydream/parameters.py:49-70).  ``test_value`` is any representative value; only its size is used."""

    def __init__(self, test_value):
        self.dsize = np.size(test_value)

    def prior(self, q0):
        return 0

    def interval(self, alpha=1):
        """(-inf, +inf) in every dimension, whatever ``alpha``."""
        return [[-np.inf] * self.dsize, [np.inf] * self.dsize]

    def device_prior(self):
        return np.zeros(self.dsize, dtype=np.int32), np.zeros(self.dsize), np.ones(self.dsize)
