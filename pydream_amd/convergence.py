"""Gelman-Rubin diagnostic -- same function name, argument and result as pydream/convergence.py:3-20."""
import numpy as np


def Gelman_Rubin(sampled_parameters):
    """Potential scale reduction factor per dimension.

    ``sampled_parameters`` is a sequence of per-chain traces ``[niterations, d]``.  Definition (the reference's,
    convergence.py:3-20): only the second half of every chain enters the moments; W is the mean over chains of the
    population (ddof = 0) variance within a chain, B the population variance of the chain means;
    ``var_est = W (1 - 1/n) + B`` with n the FULL chain length, and the result is ``sqrt(var_est / W)``."""
    traces = np.stack([np.asarray(t, dtype=float) for t in sampled_parameters])      # [chain, iteration, d]
    n = traces.shape[1]
    tail = traces[:, n // 2:, :]
    within = tail.var(axis=1).mean(axis=0)
    between = tail.mean(axis=1).var(axis=0)
    pooled = within * (1.0 - 1.0 / n) + between
    return np.sqrt(pooled / within)
