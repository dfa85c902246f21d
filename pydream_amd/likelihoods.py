"""Built-in likelihoods that run on the device ("batched device callback").

Each object is an ordinary Python callable ``f(x[d]) -> float`` (so it also works with the
reference), and carries a descriptor that ``run_dream`` hands to ``dz_set_likelihood_*`` so the
evaluation happens inside the HIP kernels instead of on the host.
"""
import numpy as np


class MVNormalLogLike:
    """log_F - 1/2 (x-mu)^T P (x-mu)  (pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52).

    factorize=True hands the device the upper-triangular factor U of P = U^T U (half the work of the
    dense form; agrees with it to ~1e-12); factorize=False uses P itself."""

    def __init__(self, precision, mu=None, log_F=0.0, factorize=True):
        P = np.asarray(precision, dtype=float)
        self.d = P.shape[0]
        self.precision = P
        self.mu = np.zeros(self.d) if mu is None else np.asarray(mu, dtype=float)
        self.log_F = float(log_F)
        self.factorize = bool(factorize)
        self.U = np.linalg.cholesky((P + P.T) / 2.0).T if self.factorize else None

    def __call__(self, x):
        v = np.asarray(x, dtype=float) - self.mu
        return self.log_F - .5 * np.sum(v * np.dot(self.precision, v))

    def _dz_apply(self, engine):
        if self.factorize:
            engine.set_likelihood_mvn(self.mu, self.U, 1, self.log_F)
        else:
            engine.set_likelihood_mvn(self.mu, self.precision, 0, self.log_F)


class GaussianMixtureLogLike:
    """log sum_j exp(-1/2 |x - mu_j|^2 + log_F_j)  (pydream/examples/mixturemodel/mixturemodel.py:37-48)."""

    def __init__(self, mu, log_F):
        self.mu = np.atleast_2d(np.asarray(mu, dtype=float))
        self.log_F = np.asarray(log_F, dtype=float)
        self.d = self.mu.shape[1]

    @classmethod
    def from_weights(cls, mu, weights):
        mu = np.atleast_2d(np.asarray(mu, dtype=float))
        d = mu.shape[1]
        return cls(mu, np.log(np.asarray(weights, dtype=float)) - (d / 2.) * np.log(2 * np.pi))

    def __call__(self, x):
        log_lh = -.5 * np.sum((np.asarray(x, dtype=float) - self.mu) ** 2, axis=1) + self.log_F
        m = np.max(log_lh)
        return np.log(np.sum(np.exp(log_lh - m))) + m

    def _dz_apply(self, engine):
        engine.set_likelihood_mixture(self.mu, self.log_F)
