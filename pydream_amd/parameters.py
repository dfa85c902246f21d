"""Parameter priors -- same interface as pydream/parameters.py (SampledParam, FlatParam).

Additionally each parameter can describe itself to the device (``device_prior``) so that
``scipy.stats.norm`` / ``scipy.stats.uniform`` / flat priors are evaluated inside the HIP kernels
(``dz_set_prior``); any other distribution is evaluated on the host through ``.prior``.
"""
import numpy as np


class SampledParam():
    """A SciPy-based parameter prior class (pydream/parameters.py:6-47).

    Parameters
    ----------
    scipy_distribution: SciPy continuous random variable class
        A SciPy statistical distribution (i.e. scipy.stats.norm)
    args, kwargs:
        Arguments for the SciPy distribution
    """

    def __init__(self, scipy_distribution, *args, **kwargs):
        self.dist = scipy_distribution(*args, **kwargs)
        self.dsize = self.random().size

    def interval(self, alpha=1):
        """Return the interval for a given alpha value (parameters.py:23-26)."""
        return self.dist.interval(alpha)

    def random(self, reseed=False):
        """Return a random value drawn from this prior (parameters.py:28-35)."""
        if reseed:
            random_seed = np.random.RandomState()
        else:
            random_seed = None
        return self.dist.rvs(random_state=random_seed)

    def prior(self, q0):
        """Return the prior log probability given a point (parameters.py:37-47)."""
        logp = np.sum(self.dist.logpdf(q0))
        return logp

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
    """A Flat parameter class (returns 0 at all locations) (pydream/parameters.py:49-70).

    Parameters
    ----------
    test_value: array
        Representative value for the parameter.  Used to infer the parameter dimension.
    """

    def __init__(self, test_value):
        self.dsize = np.asarray(test_value).size

    def prior(self, q0):
        return 0

    def interval(self, alpha=1):
        """Return the interval for a given alpha value."""
        lower = [-np.inf] * self.dsize
        upper = [np.inf] * self.dsize
        return [lower, upper]

    def device_prior(self):
        return np.zeros(self.dsize, dtype=np.int32), np.zeros(self.dsize), np.ones(self.dsize)
