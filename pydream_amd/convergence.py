"""Gelman-Rubin diagnostic -- same function name, argument and result as pydream/convergence.py:3-20."""
import numpy as np


def _common_base(traces):
    """(base, n) if the traces are the chains of ONE C-contiguous float64 array [chain, iteration, d] in order (what run_dream
    returns: core.py's per-chain views of its result array), else None -- found from the views' addresses, no data touched"""
    t0 = traces[0]
    if not isinstance(t0, np.ndarray) or t0.ndim != 2 or t0.dtype != np.float64 or not t0.flags.c_contiguous or t0.size == 0:
        return None
    base = t0.base
    if not isinstance(base, np.ndarray) or base.ndim != 3 or base.dtype != np.float64 or not base.flags.c_contiguous:
        return None
    if base.shape[0] != len(traces) or base.shape[1:] != t0.shape:
        return None
    a0, step = base.__array_interface__['data'][0], base.strides[0]
    for c, t in enumerate(traces):
        if not isinstance(t, np.ndarray) or t.base is not base or t.shape != t0.shape or t.strides != t0.strides or \
                t.__array_interface__['data'][0] != a0 + c * step:
            return None
    return base


def Gelman_Rubin(sampled_parameters):
    """Potential scale reduction factor per dimension.

    ``sampled_parameters`` is a sequence of per-chain traces ``[niterations, d]``.  Definition (the reference's,
    convergence.py:3-20): only the second half of every chain enters the moments; W is the mean over chains of the
    population (ddof = 0) variance within a chain, B the population variance of the chain means;
    ``var_est = W (1 - 1/n) + B`` with n the FULL chain length, and the result is ``sqrt(var_est / W)``.

    Nothing is copied: the moments are taken chain by chain as the reference takes them (convergence.py:8-14) -- and when the traces are
    the chains of one array (what run_dream returns; 6.5 GB at 4096 chains x 2000 iterations x 100-D, which np.stack would duplicate)
    a block of chains at a time on that array in place.  ``run_dream``'s result also carries the diagnostic made on the device from the
    run's resident trace: ``sampled_params.gelman_rubin`` (``Gelman_Rubin_device``)."""
    nchains = len(sampled_parameters)
    n = len(sampled_parameters[0])
    nb = n // 2
    base = _common_base(sampled_parameters)
    if base is not None:
        d = base.shape[2]
        chain_var, chain_means = np.empty((nchains, d)), np.empty((nchains, d))
        blk = max(1, (32 << 20) // max(1, (n - nb) * d * 8))          # the variance's temporary stays about 32 MB
        for c0 in range(0, nchains, blk):
            tail = base[c0:c0 + blk, nb:, :]
            chain_var[c0:c0 + blk] = tail.var(axis=1)
            chain_means[c0:c0 + blk] = tail.mean(axis=1)
    else:
        chain_var = [np.var(np.asarray(sampled_parameters[c], dtype=float)[nb:, :], axis=0) for c in range(nchains)]
        chain_means = [np.mean(np.asarray(sampled_parameters[c], dtype=float)[nb:, :], axis=0) for c in range(nchains)]
    within = np.mean(chain_var, axis=0)
    between = np.var(chain_means, axis=0)
    pooled = within * (1.0 - 1.0 / n) + between
    return np.sqrt(np.divide(pooled, within))


class SampledList(list):
    """The list of per-chain traces run_dream returns (core.py:127), plus ``gelman_rubin``: the diagnostic of THIS run over all its
    chains, made on the device from the resident trace (k_chain_moments / k_rhat: 0.3 ms where the host pass reads gigabytes), or
    None when the run did not stay resident (chunked runs, tempering)."""
    gelman_rubin = None


def Gelman_Rubin_device(sampled_parameters):
    """the device-made diagnostic carried by a run_dream result; falls back to Gelman_Rubin on anything else"""
    r = getattr(sampled_parameters, "gelman_rubin", None)
    return np.array(r) if r is not None else Gelman_Rubin(sampled_parameters)
