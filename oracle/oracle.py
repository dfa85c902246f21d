"""ctypes front-end of the CPU oracle (oracle/dreamzs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by
``bench.py``'s ``cpu_baseline`` leg -- never by the ``pydream_amd`` package.  The
method names mirror ``pydream_amd._capi.Engine`` so parity tests can drive both
objects with the same code.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(HERE, "dreamzs_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


class Config(C.Structure):
    _fields_ = [
        ("nchains", C.c_int32), ("nchains_local", C.c_int32), ("chain_offset", C.c_int32), ("ndim", C.c_int32),
        ("multitry", C.c_int32), ("depairs", C.c_int32), ("ncr", C.c_int32), ("ngamma", C.c_int32),
        ("history_thin", C.c_int32), ("crossover_burnin", C.c_int32), ("adapt_crossover", C.c_int32),
        ("adapt_gamma", C.c_int32), ("hardboundaries", C.c_int32), ("schedule", C.c_int32), ("device", C.c_int32),
        ("history_lag", C.c_int32), ("adapt_lag", C.c_int32), ("reserved0", C.c_int32), ("history_capacity", C.c_int64), ("trace_capacity", C.c_int64),
        ("seed", C.c_uint64), ("lamb", C.c_double), ("zeta", C.c_double), ("snooker", C.c_double),
        ("p_gamma_unity", C.c_double), ("temperature", C.c_double),
    ]


LOGP_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
XCHG_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        if "OMP_NUM_THREADS" not in os.environ:       # the cores this process may really use: affinity and cgroup quota
            n = len(os.sched_getaffinity(0))
            try:
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
                if quota != "max":
                    n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
            except (OSError, ValueError):
                pass
            os.environ["OMP_NUM_THREADS"] = str(n)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_generation.restype = C.c_int64
        L.orc_u53.restype = C.c_double
        L.orc_u53.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_u32.restype = C.c_double
        L.orc_u32.argtypes = [C.c_uint32]
        L.orc_normal32.restype = C.c_float
        L.orc_normal32.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_normal32_sin.restype = C.c_float
        L.orc_normal32_sin.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_u16.restype = C.c_double
        L.orc_u16.argtypes = [C.c_uint32]
        L.orc_uniform16.restype = C.c_double
        L.orc_uniform16.argtypes = [C.c_uint32, C.c_double, C.c_double]
        L.orc_exp.restype = C.c_double
        L.orc_exp.argtypes = [C.c_double]
        L.orc_log.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]
        L.orc_stream_id.restype = C.c_uint32
        L.orc_wave_dot.restype = C.c_double
        L.orc_loglike.restype = C.c_double
        L.orc_invcdf.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_philox4x32_10.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_set_likelihood_mvn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double]
        L.orc_step.argtypes = [C.c_void_p, C.c_int64]
        L.orc_get_trace.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6
        L.orc_get_history.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_set_history.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- stateless building blocks (random contract, elementary functions) ----
def philox(seed, c0, c1, c2, c3):
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(C.c_uint64(seed), c0, c1, c2, c3, _p(out))
    return out


K_CTRL, K_PT, K_DIM, K_BND = 0, 1, 2, 3


def stream_id(kind, tr=0, phase=0, rnd=0):
    return int(lib().orc_stream_id(kind, tr, phase, rnd))


def u53(hi, lo):
    return float(lib().orc_u53(int(hi), int(lo)))


def u32(w):
    return float(lib().orc_u32(int(w)))


def normal32(w1, w2):
    return float(lib().orc_normal32(int(w1), int(w2)))


def normal32_sin(w1, w2):
    return float(lib().orc_normal32_sin(int(w1), int(w2)))


def u16(h):
    return float(lib().orc_u16(int(h) & 0xffff))


def uniform16(h, low, high):
    return float(lib().orc_uniform16(int(h) & 0xffff, float(low), float(high)))


def threads():
    """OpenMP threads the oracle's lockstep generation spreads the chains over (OMP_NUM_THREADS)."""
    return int(lib().orc_threads())


def exp(x):
    return float(lib().orc_exp(float(x)))


def log(x):
    return float(lib().orc_log(float(x)))


def invcdf(p, u):
    p = _f64(p)
    return int(lib().orc_invcdf(_p(p), len(p), float(u)))


def sample_distinct(words, M):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.zeros(len(words), dtype=np.uint32)
    rc = lib().orc_sample_distinct(_p(words), len(words), C.c_uint32(M), _p(out))
    if rc:
        raise ValueError("sample_distinct failed")
    return out


def wave_dot(a, b):
    a, b = _f64(a), _f64(b)
    return float(lib().orc_wave_dot(_p(a), _p(b), len(a)))


def gamma_table(ngamma, depairs, d):
    out = np.zeros((ngamma, depairs, d))
    lib().orc_gamma_table(ngamma, depairs, d, _p(out))
    return out


def gelman_rubin(traces):
    """traces: array [nchains, nsamples, d] (convergence.py:3-20)."""
    tr = _f64(traces)
    nch, ns, d = tr.shape
    out = np.zeros(d)
    lib().orc_gelman_rubin(_p(tr), nch, ns, d, _p(out))
    return out


class OracleError(RuntimeError):
    pass


class Engine:
    """One oracle instance.  kwargs are the fields of ``Config``."""

    def __init__(self, **kw):
        self.L = lib()
        cfg = Config()
        defaults = dict(nchains_local=kw.get("nchains"), chain_offset=0, multitry=1, depairs=1, ncr=3, ngamma=1,
                        history_thin=10, crossover_burnin=0, adapt_crossover=0, adapt_gamma=0, hardboundaries=1,
                        schedule=2, device=0, history_lag=0, adapt_lag=0, reserved0=0, trace_capacity=0, seed=0, lamb=0.05, zeta=1e-12,
                        snooker=0.1, p_gamma_unity=0.2, temperature=1.0)
        defaults.update(kw)
        for k, v in defaults.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.N, self.nl, self.d, self.k = cfg.nchains, cfg.nchains_local, cfg.ndim, cfg.multitry
        h = C.c_void_p()
        self._chk(self.L.orc_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._keep = []

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(self.L.orc_last_error().decode())

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bounds(self, mins, maxs):
        mins, maxs = _f64(mins), _f64(maxs)
        self._chk(self.L.orc_set_bounds(self.h, _p(mins), _p(maxs)))

    def set_gamma_table(self, table):
        t = _f64(table)
        self._chk(self.L.orc_set_gamma_table(self.h, _p(t)))

    def set_history(self, Z):
        Z = _f64(Z).reshape(-1, self.d)
        self._chk(self.L.orc_set_history(self.h, _p(Z), Z.shape[0]))

    def set_state(self, X, prior=None, like=None):
        X = _f64(X).reshape(self.nl, self.d)
        pr = None if prior is None else _f64(prior)
        lk = None if like is None else _f64(like)
        self._chk(self.L.orc_set_state(self.h, _p(X), _p(pr), _p(lk)))

    def set_cr_probs(self, p):
        p = _f64(p)
        self._chk(self.L.orc_set_cr_probs(self.h, _p(p), len(p)))

    def set_gamma_probs(self, p):
        p = _f64(p)
        self._chk(self.L.orc_set_gamma_probs(self.h, _p(p), len(p)))

    def set_prior(self, kind, a, b):
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        a, b = _f64(a), _f64(b)
        self._chk(self.L.orc_set_prior(self.h, _p(kind), _p(a), _p(b)))

    def set_likelihood_mvn(self, mu, M, kind=0, log_F=0.0):
        mu, M = _f64(mu), _f64(M)
        self._chk(self.L.orc_set_likelihood_mvn(self.h, _p(mu), _p(M), kind, float(log_F)))

    def set_likelihood_mixture(self, mu, log_F):
        mu, log_F = _f64(mu), _f64(log_F)
        self._chk(self.L.orc_set_likelihood_mixture(self.h, mu.shape[0], _p(mu), _p(log_F)))

    def set_likelihood_host(self, fn):
        """fn(X[n,d]) -> (prior[n], like[n])"""
        d = self.d

        def tramp(Xp, n, dd, pp, lp, user):
            try:
                X = np.ctypeslib.as_array(Xp, shape=(n, d))
                pr, lk = fn(X.copy())
                np.ctypeslib.as_array(pp, shape=(n,))[:] = pr
                np.ctypeslib.as_array(lp, shape=(n,))[:] = lk
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        cb = LOGP_CB(tramp)
        self._keep.append(cb)
        self._chk(self.L.orc_set_likelihood_host(self.h, cb, None))

    def set_temperatures(self, T, swaps=True):
        T = _f64(T)
        assert len(T) == self.N
        self._chk(self.L.orc_set_temperatures(self.h, _p(T), int(bool(swaps))))

    def get_swaps(self, g0, ng):
        out = np.zeros((ng, 3), np.int32)
        self.L.orc_get_swaps.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        self._chk(self.L.orc_get_swaps(self.h, g0, ng, _p(out)))
        return out

    def set_exchange(self, fn):
        """fn(send: bytes-like, nbytes) -> bytes of all ranks' blocks in rank order"""
        def tramp(send, recv, nbytes, user):
            try:
                buf = (C.c_char * nbytes).from_address(send)
                out = fn(bytes(buf), nbytes)
                C.memmove(recv, out, len(out))
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        cb = XCHG_CB(tramp)
        self._keep.append(cb)
        self._chk(self.L.orc_set_exchange(self.h, cb, None))

    def step(self, generations=1):
        self._chk(self.L.orc_step(self.h, int(generations)))

    def trace_reset(self):
        self._chk(self.L.orc_trace_reset(self.h))

    @property
    def generation(self):
        return int(self.L.orc_generation(self.h))

    def get_state(self):
        X = np.zeros((self.nl, self.d)); pr = np.zeros(self.nl); lk = np.zeros(self.nl)
        self._chk(self.L.orc_get_state(self.h, _p(X), _p(pr), _p(lk)))
        return X, pr, lk

    def get_trace(self, g0, ng):
        nl, d = self.nl, self.d
        out = dict(X=np.zeros((ng, nl, d)), logp=np.zeros((ng, nl)), moved=np.zeros((ng, nl), np.uint8),
                   try_idx=np.zeros((ng, nl), np.int32), cr_idx=np.zeros((ng, nl), np.int32),
                   snooker=np.zeros((ng, nl), np.uint8))
        self._chk(self.L.orc_get_trace(self.h, g0, ng, _p(out["X"]), _p(out["logp"]), _p(out["moved"]),
                                       _p(out["try_idx"]), _p(out["cr_idx"]), _p(out["snooker"])))
        return out

    def get_history(self):
        rows = C.c_int64()
        self._chk(self.L.orc_get_history(self.h, None, 0, C.byref(rows)))
        Z = np.zeros((rows.value, self.d))
        self._chk(self.L.orc_get_history(self.h, _p(Z), rows.value, C.byref(rows)))
        return Z

    def get_cr_state(self):
        n = self.cfg.ncr
        p, dm, nu = np.zeros(n), np.zeros(n), np.zeros(n)
        self._chk(self.L.orc_get_cr_state(self.h, _p(p), _p(dm), _p(nu)))
        return p, dm, nu

    def get_gamma_state(self):
        n = self.cfg.ngamma
        p, dm, nu = np.zeros(n), np.zeros(n), np.zeros(n)
        self._chk(self.L.orc_get_gamma_state(self.h, _p(p), _p(dm), _p(nu)))
        return p, dm, nu

    def get_rhat(self):
        r = np.zeros(self.d)
        self._chk(self.L.orc_get_rhat(self.h, _p(r)))
        return r

    def loglike(self, x):
        x = _f64(x)
        return float(self.L.orc_loglike(self.h, _p(x)))

    def debug_propose(self, chain, gen, phase, base, snooker, cr_idx, delta=1, glev=1):
        n = self.k if phase == 0 else self.k - 1
        base = _f64(base)
        pts = np.zeros((n, self.d)); slogp = np.zeros(n); gam = np.zeros(n); zidx = np.zeros((n, 16), np.int64)
        self.L.orc_debug_propose.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.L.orc_debug_propose(self.h, chain, gen, phase, _p(base), int(snooker), cr_idx, delta, glev,
                                           _p(pts), _p(slogp), _p(gam), _p(zidx)))
        return pts, slogp, gam, zidx
