"""Model: the likelihood/prior boundary -- same interface as pydream/model.py."""
import numpy as np


class Model():
    """pydream/model.py:8-32"""

    def __init__(self, likelihood, sampled_parameters):
        self.likelihood = likelihood
        if type(sampled_parameters) is list:
            self.sampled_parameters = sampled_parameters
        else:
            self.sampled_parameters = [sampled_parameters]

    def total_logp(self, q0):
        """(prior_logp, loglike) of one point (model.py:17-32)."""
        prior_logp = 0
        var_start = 0
        for param in self.sampled_parameters:
            var_end = param.dsize + var_start
            try:
                prior_logp += param.prior(q0[var_start:var_end])
            except IndexError:
                # raised if q0 is a single scalar
                prior_logp += param.prior(q0)
            var_start += param.dsize
        loglike = self.likelihood(q0)
        return prior_logp, loglike

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
