"""Process-wide handle to the device-resident sampler state.

In the reference this (empty) module is the namespace through which every chain process reaches
the multiprocessing shared arrays (pydream/core.py:316-327): history, current_positions, cross_probs,
ncr_updates, delta_m, gamma_level_probs, ngamma_updates, delta_m_gamma, count, nchains.  Here all of
those live in HBM inside ONE engine handle; the module keeps that handle (``engine``) plus the
small amount of host bookkeeping ``Dream.astep`` needs.  The read-only properties below give the
reference's names a meaning for code that inspects them.
"""
import numpy as np

engine = None             # pydream_amd._capi.Engine
nchains_counter = 0       # counts down as Dream instances claim chain ids (Dream.py:198-200)
host_state = {}           # chain id -> last state returned by astep
temperatures = None       # per-chain T once an astep call used T != 1 (parallel tempering driven from the host)
rng = None                # numpy RandomState used for prior draws (history seeding, random starts)


def draw_from_prior(model_vars):
    """One point drawn from the priors of ``model_vars``, as a flat vector (what Dream.draw_from_prior, Dream.py:628-644,
    returns), using this module's RandomState when there is one so that seeded runs are reproducible."""
    pieces = []
    for variable in model_vars:
        try:
            frozen = getattr(variable, "dist", None)
            value = frozen.rvs(random_state=rng) if (rng is not None and frozen is not None) else variable.random()
        except AttributeError:
            raise Exception('Random draw from distribution for variable %s not implemented yet.' % variable)
        pieces.append(np.ravel(value))
    return np.concatenate(pieces) if pieces else np.array([])


def history():
    """Flat float64 view of the Z archive, seed rows first (Dream_shared_vars.history)."""
    return engine.get_history().reshape(-1)


def count():
    """Number of rows appended so far (Dream_shared_vars.count.value)."""
    return engine.history_rows() - engine.nseed


def cross_probs():
    return engine.get_cr_state()[0]


def ncr_updates():
    return engine.get_cr_state()[2]


def delta_m():
    return engine.get_cr_state()[1]


def gamma_level_probs():
    return engine.get_gamma_state()[0]
