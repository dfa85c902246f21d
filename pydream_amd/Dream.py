"""The DREAM step object -- same constructor and ``astep`` signature as pydream/Dream.py.

In the reference a ``Dream`` instance *is* the sampler: ``astep`` advances one chain with numpy.
Here it is the host-side description of the sampler; the transitions themselves run on the GPU
(libdreamzs.so).  ``astep`` remains callable as the single-chain view of the engine (one chain
advanced through ``dz_step_range``), which is how the reference's own tests drive it
(pydream/tests/test_dream.py:507-518).
"""
import numpy as np

from . import Dream_shared_vars


class Dream():
    """An implementation of the MT-DREAM\\ :sub:`(ZS)`\\  algorithm (Laloy & Vrugt 2012) -- options as in
    pydream/Dream.py:12-67 (same names, defaults and meaning).

    Parameters
    ----------
    model : Model, variables : iterable of SampledParam (default: all of the model's)
    nseedchains : int, history seed rows (default 10 * dimension)
    nCR : int, adapt_crossover : bool, adapt_gamma : bool, crossover_burnin : int
    DEpairs : int, lamb : float, zeta : float, history_thin : int, snooker : float, p_gamma_unity : float
    gamma_levels : int, start_random : bool, save_history : bool, history_file / crossover_file / gamma_file : str
    multitry : bool or int, parallel : bool (every try is evaluated in one batch anyway; True makes a Python likelihood
        use worker processes from the first batch on instead of once a batch has proved slow -- model.HostEvaluator)
    verbose : bool, model_name : str, hardboundaries : bool, mp_context : start method of those worker processes
    """

    def __init__(self, model, variables=None, nseedchains=None, nCR=3, adapt_crossover=True, adapt_gamma=False,
                 crossover_burnin=None, DEpairs=1, lamb=.05, zeta=1e-12, history_thin=10, snooker=.10,
                 p_gamma_unity=.20, gamma_levels=1, start_random=True, save_history=True, history_file=False,
                 crossover_file=False, gamma_file=False, multitry=False, parallel=False, verbose=False,
                 model_name=False, hardboundaries=True, mp_context=None, **kwargs):
        # what is sampled
        self.model = model
        self.model_name = model_name
        self.mp_context = mp_context
        self.variables = model.sampled_parameters if variables is None else variables
        self.total_var_dimension = int(sum(var.dsize for var in self.variables))
        d = self.total_var_dimension
        self.logp = model.total_logp

        # hard boundaries = the support of every prior (Dream.py:79-105)
        self.boundaries = hardboundaries
        if hardboundaries:
            self.boundary_mask = True if d == 1 else np.ones(d, dtype=bool)
            lows, highs = [], []
            for var in self.variables:
                lo, hi = var.interval(1)
                lows.extend(np.ravel(lo)[:var.dsize])
                highs.extend(np.ravel(hi)[:var.dsize])
            self.mins = np.asarray(lows, dtype=float)
            self.maxs = np.asarray(highs, dtype=float)

        # crossover values and their selection probabilities (Dream.py:107-134, :146)
        self.nCR = nCR
        if nCR > d:
            self.nCR = d
            print('Warning: the total number of crossover values specified (' + str(nCR) + ') is less than the total dimension of all variables (' + str(d) + ').  Setting the number of crossover values to be equal to the total variable dimension.')
        if d == 1 and adapt_crossover:
            adapt_crossover = False
            print('Warning: the total variable dimension = 1, so crossover values will not be adapted, even though crossover adaptation was requested.')
        self.adapt_crossover = adapt_crossover
        self.crossover_burnin = crossover_burnin
        self.crossover_file = crossover_file
        if crossover_file:
            self.CR_probabilities = np.load(crossover_file)
            self.nCR = len(self.CR_probabilities)
            if adapt_crossover:
                print('Warning: Crossover values loaded and adapt_crossover = True.  Crossover values will be further adapted.')
        else:
            self.CR_probabilities = [1.0 / self.nCR] * self.nCR
        self.CR_values = np.arange(1, self.nCR + 1) / float(self.nCR)

        # gamma levels (Dream.py:120-122, :136-147)
        self.ngamma = gamma_levels
        self.njoint_cr_gamma_probs = nCR * gamma_levels
        self.adapt_gamma = adapt_gamma
        self.gamma_file = gamma_file
        if gamma_file:
            self.gamma_probabilities = np.load(gamma_file)
            if adapt_gamma:
                print('Warning: Gamma values loaded and adapt gamma = True.  Gamma values will be further adapted.')
        else:
            self.gamma_probabilities = [1.0 / gamma_levels] * gamma_levels
        self.gamma_level_values = np.arange(1, gamma_levels + 1)

        # jump parameters
        self.DEpairs = np.arange(1, DEpairs + 1, dtype=int)           # (the reference keeps the list 1..DEpairs, Dream.py:148)
        self.snooker = snooker
        self.p_gamma_unity = p_gamma_unity
        self.lamb = lamb
        self.zeta = zeta
        self.parallel = parallel
        # multitry: False -> 1 try, True -> 5 tries, a number -> that many (Dream.py:155-161)
        # (compared by value like the reference does, so 0 counts as False and 1 as True)
        self.multitry = 1 if multitry == False else (5 if multitry == True else multitry)      # noqa: E712  (1.0 / 0.0 too, as in the reference)
        if self.multitry == 2:
            raise Exception('multitry=2 fails inside the reference (Dream.py:867-868); use 1 or >= 3.')
        # What the reference accepts and this engine does not (any integer there: Dream.py:108-122, :148-161, :81-83): said here, when the
        # sampler is described, with the limit -- not later by dz_create.  The limits are the fixed lane budgets of the GPU kernels: a
        # generation's multi-try weights, crossover / gamma-level bins and DE pairs are held one per lane of a 64-lane wave, and a
        # point's dimensions two per lane in at most eight 128-dimension chunks.
        LIMITS = (('multitry', self.multitry, 32), ('DEpairs', DEpairs, 8), ('nCR', self.nCR, 32), ('gamma_levels', gamma_levels, 32),
                  ('the total dimension of all variables', d, 1024))
        for what, value, limit in LIMITS:
            if value > limit:
                raise Exception('%s = %d: this GPU engine supports at most %d (the reference, pydream/Dream.py, takes any integer).' % (what, value, limit))

        # jump scale 2.38 / sqrt(2 delta d') by (gamma level, number of pairs delta, crossed dimensions d'), halved from one
        # level to the next (Dream.py:172-179)
        dprime = np.arange(1, d + 1, dtype=float)
        self.gamma_arr = np.empty((gamma_levels, DEpairs, d))
        for level in range(gamma_levels):
            for delta in range(1, DEpairs + 1):
                self.gamma_arr[level, delta - 1, :] = (2.38 / np.sqrt(2 * delta * dprime)) / (2 ** level)
        self.gamma = None

        # history archive and bookkeeping
        self.nseedchains = 10 * d if nseedchains is None else nseedchains      # Dream.py:168-170
        self.save_history = save_history
        self.history_file = history_file
        self.history_thin = history_thin
        self.start_random = start_random
        self.verbose = verbose
        self.len_history = 0
        self.iter = 0
        self.chain_n = None
        self.nchains = None
        self.last_logp = None
        self.last_prior = None
        self.last_like = None

    # ------------------------------------------------------------------
    def astep(self, q0, T=1., last_loglike=None, last_logprior=None):
        """One MT-DREAM(ZS) transition of ONE chain (Dream.py:193-422): returns (q_new, last_prior, last_like).

        The engine must have been set up by ``core._setup_mp_dream_pool`` + ``pool._initializer`` (the
        reference's single-process idiom).  The chain is advanced on the GPU with immediate effect on
        the shared state (history append, published position, crossover statistics), i.e. the
        sequential round-robin semantics the reference has when driven this way."""
        eng = getattr(Dream_shared_vars, "engine", None)
        if eng is None:
            raise Exception('Dream should be run with multiple chains in parallel.  Set nchains > 1.')   # Dream.py:236
        if self.chain_n is None:                      # Dream.py:198-200: claim a chain id, counting down
            Dream_shared_vars.nchains_counter -= 1
            self.chain_n = Dream_shared_vars.nchains_counter
            if self.chain_n < 0:
                raise Exception('more Dream instances than chains')
            self.nchains = eng.N
            if q0 is None or self.start_random:       # Dream.py:221-225
                q0 = Dream_shared_vars.draw_from_prior(self.variables)
        q0 = np.asarray(q0, dtype=float).reshape(-1)
        c = self.chain_n
        temps = Dream_shared_vars.temperatures
        if temps is None and T != 1.:
            temps = Dream_shared_vars.temperatures = np.ones(eng.N)
        if temps is not None and temps[eng.cfg.chain_offset + c] != T:     # this chain's T (Dream.py:193, core.py:240-246)
            temps[eng.cfg.chain_offset + c] = T
            eng.set_temperatures(temps, swaps=False)
        cur = Dream_shared_vars.host_state.get(c)
        if last_loglike is not None:                  # Dream.py:240-243
            eng.set_chain_state(c, q0, last_logprior, last_loglike)
        elif cur is None or not np.array_equal(cur, q0):
            eng.set_chain_state(c, q0, None, None)    # log density evaluated on the device (Dream.py:266-268)
        eng.step_range(c, 1)
        q_new, pr, lk = eng.get_chain_state(c)
        Dream_shared_vars.host_state[c] = q_new.copy()
        self.last_prior, self.last_like = float(pr), float(lk)
        self.last_logp = T * self.last_like + self.last_prior       # Dream.py:243
        if (self.adapt_crossover or self.adapt_gamma) and self.crossover_burnin is not None and self.iter <= self.crossover_burnin and hasattr(eng, "get_chain_probs"):
            # this instance's own copies, refreshed by its own updates (Dream.py:375, :383, :409-415)
            cr, gp = eng.get_chain_probs(c)
            if self.adapt_crossover:
                self.CR_probabilities = list(cr)
            if self.adapt_gamma:
                self.gamma_probabilities = list(gp)
        self.iter += 1
        return q_new, self.last_prior, self.last_like
