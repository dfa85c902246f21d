"""ctypes binding of libdreamzs.so (include/dreamzs.h) -- the only way the Python host talks to the GPU.

There is no CPU fallback: if the library cannot be loaded, or no MI355X is present, every
entry point raises.  The library is built in-tree by ``python -m pydream_amd.build``.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DREAMZS_LIB") or os.path.join(HERE, "libdreamzs.so")   # DREAMZS_LIB: an alternative build


class DreamZSError(RuntimeError):
    pass


class Config(C.Structure):
    """dz_config (include/dreamzs.h)"""
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

# every symbol include/dreamzs.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "dz_version", "dz_last_error", "dz_device_count", "dz_create", "dz_destroy", "dz_set_bounds", "dz_set_gamma_table",
    "dz_set_history", "dz_set_state", "dz_set_cr_probs", "dz_set_gamma_probs", "dz_set_prior", "dz_set_likelihood_mvn",
    "dz_set_likelihood_mixture", "dz_set_likelihood_host", "dz_set_likelihood_module", "dz_hip_library", "dz_comm_library", "dz_comm_unique_id", "dz_comm_init_rccl", "dz_comm_count", "dz_comm_barrier", "dz_set_exchange", "dz_peer_export", "dz_peer_attach", "dz_peer_detach", "dz_exchange_stats", "dz_exchange_bytes", "dz_set_temperatures", "dz_get_swaps",
    "dz_step", "dz_continue_run", "dz_step_range", "dz_set_chain_state", "dz_get_chain_state", "dz_get_chain_probs", "dz_sync", "dz_trace_reset", "dz_generation", "dz_redraw_rounds", "dz_last_kernel_variant", "dz_get_state", "dz_get_trace", "dz_get_trace_chains", "dz_trace_download_begin", "dz_trace_download_wait", "dz_host_register", "dz_host_unregister", "dz_get_history", "dz_get_history_range", "dz_history_checksum",
    "dz_get_cr_state", "dz_get_gamma_state", "dz_get_rhat", "dz_get_chain_moments", "dz_eval_logp", "dz_debug_propose",
    "dz_profile_enable", "dz_profile_get", "dz_profile_reset", "dz_profile_get_list",
]

_lib = None


def load_library():
    """Load libdreamzs.so; raises DreamZSError (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DreamZSError("%s is missing: build it with `python -m pydream_amd.build` (needs hipcc); "
                           "pydream_amd has no CPU fallback" % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise DreamZSError("cannot load %s: %s" % (LIB_PATH, exc))
    V = C.c_void_p
    L.dz_last_error.restype = C.c_char_p
    L.dz_generation.restype = C.c_int64
    L.dz_generation.argtypes = [V]
    L.dz_redraw_rounds.restype = C.c_int64
    L.dz_redraw_rounds.argtypes = [V]
    L.dz_last_kernel_variant.restype = C.c_char_p
    L.dz_last_kernel_variant.argtypes = [V]
    L.dz_create.argtypes = [C.POINTER(Config), C.POINTER(V)]
    L.dz_destroy.argtypes = [V]
    L.dz_set_bounds.argtypes = [V, V, V]
    L.dz_set_gamma_table.argtypes = [V, V]
    L.dz_set_history.argtypes = [V, V, C.c_int64]
    L.dz_set_state.argtypes = [V, V, V, V]
    L.dz_set_cr_probs.argtypes = [V, V, C.c_int32]
    L.dz_set_gamma_probs.argtypes = [V, V, C.c_int32]
    L.dz_set_prior.argtypes = [V, V, V, V]
    L.dz_set_likelihood_mvn.argtypes = [V, V, V, C.c_int32, C.c_double]
    L.dz_set_likelihood_mixture.argtypes = [V, C.c_int32, V, V]
    L.dz_set_likelihood_host.argtypes = [V, LOGP_CB, V]
    L.dz_set_likelihood_module.argtypes = [V, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, V, C.c_int64]
    L.dz_comm_unique_id.argtypes = [V]
    L.dz_comm_library.restype = C.c_char_p
    L.dz_hip_library.restype = C.c_char_p
    L.dz_comm_init_rccl.argtypes = [V, C.c_int32, C.c_int32, V]
    L.dz_comm_barrier.argtypes = [V]
    L.dz_set_exchange.argtypes = [V, XCHG_CB, V]
    L.dz_peer_export.argtypes = [V, V]
    L.dz_peer_attach.argtypes = [V, C.c_int32, C.c_int32, V]
    L.dz_exchange_stats.argtypes = [V, V, V, V]
    L.dz_exchange_bytes.argtypes = [V, V, V, V]
    L.dz_comm_count.argtypes = [V, V]
    L.dz_peer_detach.argtypes = [V]
    L.dz_step.argtypes = [V, C.c_int64]
    L.dz_sync.argtypes = [V]
    L.dz_continue_run.argtypes = [V, C.c_int64, C.c_int64, C.c_uint64, C.c_int32]
    L.dz_step_range.argtypes = [V, C.c_int32, C.c_int32]
    L.dz_set_chain_state.argtypes = [V, C.c_int32, V, V, V]
    L.dz_get_chain_state.argtypes = [V, C.c_int32, V, V, V]
    L.dz_get_chain_probs.argtypes = [V, C.c_int32, V, V]
    L.dz_trace_reset.argtypes = [V]
    L.dz_get_state.argtypes = [V, V, V, V]
    L.dz_get_trace.argtypes = [V, C.c_int64, C.c_int64] + [V] * 6
    L.dz_get_history.argtypes = [V, V, C.c_int64, V]
    L.dz_history_checksum.argtypes = [V, V, V]
    L.dz_get_history_range.argtypes = [V, C.c_int64, C.c_int64, V]
    L.dz_get_cr_state.argtypes = [V, V, V, V]
    L.dz_get_gamma_state.argtypes = [V, V, V, V]
    L.dz_get_rhat.argtypes = [V, V]
    L.dz_get_chain_moments.argtypes = [V, V, V]
    L.dz_eval_logp.argtypes = [V, V, C.c_int64, V, V]
    L.dz_debug_propose.argtypes = [V, C.c_int32, C.c_int64, C.c_int32, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, V, V]
    L.dz_profile_enable.argtypes = [V, C.c_int32]
    L.dz_profile_get.argtypes = [V, C.c_int32, V, V]
    L.dz_profile_reset.argtypes = [V]
    L.dz_profile_get_list.argtypes = [V, C.c_int32, V, C.c_int64, V]
    L.dz_device_count.argtypes = [V]
    _lib = L
    return L


def device_count():
    n = C.c_int32(0)
    load_library().dz_device_count(C.byref(n))
    return n.value


def hip_library():
    """path of the HIP runtime the engine's calls are bound to"""
    return load_library().dz_hip_library().decode()


def comm_library():
    """path of the librccl the engine uses (next to the HIP runtime libdreamzs.so is linked against)"""
    L = load_library()
    p = L.dz_comm_library()
    if p is None:
        raise DreamZSError(L.dz_last_error().decode())
    return p.decode()


def comm_unique_id():
    buf = C.create_string_buffer(128)
    L = load_library()
    if L.dz_comm_unique_id(buf) != 0:
        raise DreamZSError(L.dz_last_error().decode())
    return buf.raw


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def history_checksum_host(Z):
    """include/dreamzs.h dz_history_checksum on a host array [rows, d]: sum mod 2^64 of mix64(bits + (index + 1) * golden)"""
    Z = _f64(Z)
    with np.errstate(over="ignore"):
        z = Z.reshape(-1).view(np.uint64) + (np.arange(1, Z.size + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(np.sum(z, dtype=np.uint64))


PROFILE_CLASSES = {"propose": 0, "logp": 1, "accept": 2, "adapt": 3, "exchange": 4, "generations": 5, "empty": 6}


class Engine:
    """One dz_engine handle (one GPU).  Keyword arguments are the fields of dz_config."""

    def __init__(self, **kw):
        self.L = load_library()
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
        self.h = C.c_void_p()
        self._keep = []
        self._chk(self.L.dz_create(C.byref(cfg), C.byref(self.h)))

    def _chk(self, rc):
        if rc != 0:
            raise DreamZSError(self.L.dz_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.dz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setters ----
    def set_bounds(self, mins, maxs):
        mins, maxs = _f64(mins), _f64(maxs)
        self._chk(self.L.dz_set_bounds(self.h, _p(mins), _p(maxs)))

    def set_gamma_table(self, table):
        t = None if table is None else _f64(table)
        self._chk(self.L.dz_set_gamma_table(self.h, _p(t)))

    def set_history(self, Z):
        Z = _f64(Z).reshape(-1, self.d)
        self._chk(self.L.dz_set_history(self.h, _p(Z), Z.shape[0]))

    def set_state(self, X, prior=None, like=None):
        X = _f64(X).reshape(self.nl, self.d)
        pr = None if prior is None else _f64(prior)
        lk = None if like is None else _f64(like)
        self._chk(self.L.dz_set_state(self.h, _p(X), _p(pr), _p(lk)))

    def set_cr_probs(self, p):
        p = _f64(p)
        self._chk(self.L.dz_set_cr_probs(self.h, _p(p), len(p)))

    def set_gamma_probs(self, p):
        p = _f64(p)
        self._chk(self.L.dz_set_gamma_probs(self.h, _p(p), len(p)))

    def set_prior(self, kind, a, b):
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        a, b = _f64(a), _f64(b)
        self._chk(self.L.dz_set_prior(self.h, _p(kind), _p(a), _p(b)))

    def set_likelihood_mvn(self, mu, M, kind=0, log_F=0.0):
        mu, M = _f64(mu), _f64(M)
        if M.shape != (self.d, self.d):
            raise ValueError("matrix must be [d,d]")
        self._chk(self.L.dz_set_likelihood_mvn(self.h, _p(mu), _p(M), int(kind), float(log_F)))

    def set_likelihood_mixture(self, mu, log_F):
        mu, log_F = _f64(mu), _f64(log_F)
        self._chk(self.L.dz_set_likelihood_mixture(self.h, mu.shape[0], _p(mu), _p(log_F)))

    def set_likelihood_host(self, fn):
        """fn(X[n,d]) -> (prior[n], like[n]); evaluated on the host for each batch of proposals."""
        d = self.d

        def tramp(Xp, n, dd, pp, lp, user):
            try:
                X = np.ctypeslib.as_array(Xp, shape=(n, d))
                pr, lk = fn(X.copy())
                np.ctypeslib.as_array(pp, shape=(n,))[:] = pr
                np.ctypeslib.as_array(lp, shape=(n,))[:] = lk
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = LOGP_CB(tramp)
        self._keep.append(cb)
        self._chk(self.L.dz_set_likelihood_host(self.h, cb, None))

    def set_likelihood_module(self, code_object_path, kernel_name, lanes_per_point=1, data=None, always_finite=False):
        """A user-built gfx950 code object's kernel as the likelihood (include/dreamzs.h dz_set_likelihood_module); data: bytes / a numpy
        array copied to the device and handed to the kernel."""
        blob = b"" if data is None else (data if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data).tobytes())
        buf = C.create_string_buffer(bytes(blob), len(blob)) if blob else None
        self._chk(self.L.dz_set_likelihood_module(self.h, os.fsencode(code_object_path), kernel_name.encode(), int(lanes_per_point),
                                                   1 if always_finite else 0, buf, len(blob)))

    def set_exchange(self, fn):
        """fn(send_bytes, nbytes) -> bytes of all ranks' blocks in rank order (host-staged all-gather)."""
        def tramp(send, recv, nbytes, user):
            try:
                buf = (C.c_char * nbytes).from_address(send)
                out = fn(bytes(buf), nbytes)
                C.memmove(recv, out, len(out))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = XCHG_CB(tramp)
        self._keep.append(cb)
        self._chk(self.L.dz_set_exchange(self.h, cb, None))

    def comm_init_rccl(self, rank, world, unique_id):
        self._chk(self.L.dz_comm_init_rccl(self.h, rank, world, C.c_char_p(unique_id)))

    PEER_BLOB_BYTES = 512

    def peer_export(self):
        """this rank's blob for the peer transport (IPC handles of its archive, position buffers and flag words)"""
        buf = C.create_string_buffer(self.PEER_BLOB_BYTES)
        self._chk(self.L.dz_peer_export(self.h, buf))
        return buf.raw

    def peer_attach(self, rank, world, blobs):
        """blobs: the ranks' exports in rank order (bytes, world x PEER_BLOB_BYTES)"""
        assert len(blobs) == world * self.PEER_BLOB_BYTES
        self._chk(self.L.dz_peer_attach(self.h, rank, world, C.c_char_p(blobs)))

    def peer_detach(self):
        self._chk(self.L.dz_peer_detach(self.h))

    def exchange_stats(self):
        """(exchanges queued, gates passed, microseconds the gates spent waiting) -- the exposed part of the peer exchange"""
        n, g, us = C.c_int64(), C.c_int64(), C.c_double()
        self._chk(self.L.dz_exchange_stats(self.h, C.byref(n), C.byref(g), C.byref(us)))
        return n.value, g.value, us.value

    def comm_count(self):
        """ncclCommCount of the engine's RCCL communicator (0: none)"""
        n = C.c_int32()
        self._chk(self.L.dz_comm_count(self.h, C.byref(n)))
        return n.value

    def exchange_bytes(self):
        """bytes handed to the transport for EACH other rank so far: (history rows, published positions, adaptation group sums)"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.L.dz_exchange_bytes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def comm_barrier(self):
        self._chk(self.L.dz_comm_barrier(self.h))

    def set_temperatures(self, T, swaps=True):
        T = _f64(T)
        assert len(T) == self.N
        self._chk(self.L.dz_set_temperatures(self.h, _p(T), int(bool(swaps))))

    def get_swaps(self, g0, ng):
        out = np.zeros((ng, 3), np.int32)
        self._chk(self.L.dz_get_swaps(self.h, C.c_int64(g0), C.c_int64(ng), _p(out)))
        return out

    # ---- stepping ----
    def step(self, generations=1):
        self._chk(self.L.dz_step(self.h, int(generations)))

    def continue_run(self, history_capacity, trace_capacity, seed, crossover_burnin):
        """a new run on the live engine, as run_dream(restart=True) starts one (dz_continue_run); follow with set_state"""
        self._chk(self.L.dz_continue_run(self.h, int(history_capacity), int(trace_capacity), int(seed), int(crossover_burnin)))
        self.cfg.history_capacity, self.cfg.trace_capacity = max(self.cfg.history_capacity, int(history_capacity)), max(self.cfg.trace_capacity, int(trace_capacity))
        self.cfg.seed, self.cfg.crossover_burnin = int(seed), int(crossover_burnin)

    def step_range(self, chain0, nchains=1):
        self._chk(self.L.dz_step_range(self.h, int(chain0), int(nchains)))

    def set_chain_state(self, chain, x, prior=None, like=None):
        x = _f64(x).reshape(self.d)
        pr = None if prior is None else np.array([prior], dtype=np.float64)
        lk = None if like is None else np.array([like], dtype=np.float64)
        self._chk(self.L.dz_set_chain_state(self.h, int(chain), _p(x), _p(pr), _p(lk)))

    def get_chain_state(self, chain):
        x = np.zeros(self.d); pr = np.zeros(1); lk = np.zeros(1)
        self._chk(self.L.dz_get_chain_state(self.h, int(chain), _p(x), _p(pr), _p(lk)))
        return x, float(pr[0]), float(lk[0])

    def get_chain_probs(self, chain):
        """(CR_probabilities, gamma_probabilities) of the Dream instance that drives `chain` (its own copies under step_range)"""
        cr, gp = np.zeros(self.cfg.ncr), np.zeros(self.cfg.ngamma)
        self._chk(self.L.dz_get_chain_probs(self.h, int(chain), _p(cr), _p(gp)))
        return cr, gp

    def sync(self):
        self._chk(self.L.dz_sync(self.h))

    def trace_reset(self):
        self._chk(self.L.dz_trace_reset(self.h))

    @property
    def generation(self):
        return int(self.L.dz_generation(self.h))

    def redraw_rounds(self):
        """Redraw launches so far (Dream.py:281-289: proposal sets whose tries were all impossible, drawn again)."""
        return int(self.L.dz_redraw_rounds(self.h))

    def last_kernel_variant(self):
        """what the most recent generations ran as (template instantiation of the persistent kernel, or "multi-kernel path")"""
        return self.L.dz_last_kernel_variant(self.h).decode()

    # ---- getters ----
    def get_state(self):
        X = np.zeros((self.nl, self.d)); pr = np.zeros(self.nl); lk = np.zeros(self.nl)
        self._chk(self.L.dz_get_state(self.h, _p(X), _p(pr), _p(lk)))
        return X, pr, lk

    def get_trace(self, g0, ng, with_X=True, with_logp=True):
        nl, d = self.nl, self.d
        out = dict(X=np.zeros((ng, nl, d)) if with_X else None, logp=np.zeros((ng, nl)) if with_logp else None,
                   moved=np.zeros((ng, nl), np.uint8), try_idx=np.zeros((ng, nl), np.int32), cr_idx=np.zeros((ng, nl), np.int32),
                   snooker=np.zeros((ng, nl), np.uint8))
        self._chk(self.L.dz_get_trace(self.h, g0, ng, _p(out["X"]), _p(out["logp"]), _p(out["moved"]),
                                      _p(out["try_idx"]), _p(out["cr_idx"]), _p(out["snooker"])))
        return out

    def get_trace_chains(self, g0, ng, out, row0=0, logp_out=None):
        """samples [g0, g0+ng) of every local chain into out[c, row0:row0+ng, :] (out: C-contiguous [nl, rows, d]) and,
        optionally, their log probabilities into logp_out[c, row0:row0+ng] (C-contiguous [nl, rows] or [nl, rows, 1])."""
        base, rows = None, None
        if out is not None:                     # (out=None: the log probabilities only)
            assert out.flags.c_contiguous and out.dtype == np.float64 and out.shape[0] == self.nl and out.shape[2] == self.d
            base, rows = C.c_void_p(out.ctypes.data + row0 * self.d * 8), out.shape[1]
        lp = None
        if logp_out is not None:
            assert logp_out.flags.c_contiguous and logp_out.dtype == np.float64 and logp_out.shape[0] == self.nl and logp_out.shape[1] == (rows or logp_out.shape[1])
            lp, rows = C.c_void_p(logp_out.ctypes.data + row0 * 8), logp_out.shape[1]
        self._chk(self.L.dz_get_trace_chains(self.h, C.c_int64(g0), C.c_int64(ng), base, C.c_int64(rows), lp))

    def trace_download_begin(self, g0, ng, out, row0=0):
        """as get_trace_chains for the samples, but queued behind the generations stepped so far and returning at once; `out` must be
        page-locked (host_register) and left alone until trace_download_wait()."""
        assert out.flags.c_contiguous and out.dtype == np.float64 and out.shape[0] == self.nl and out.shape[2] == self.d
        self._chk(self.L.dz_trace_download_begin(self.h, C.c_int64(g0), C.c_int64(ng), C.c_void_p(out.ctypes.data + row0 * self.d * 8), C.c_int64(out.shape[1])))

    def trace_download_wait(self):
        self._chk(self.L.dz_trace_download_wait(self.h))

    def host_register(self, arr):
        """page-lock a result array ahead of the download; returns False if the runtime refuses (the copy still works)."""
        return self.L.dz_host_register(C.c_void_p(arr.ctypes.data), C.c_int64(arr.nbytes)) == 0

    def host_unregister(self, arr):
        self.L.dz_host_unregister(C.c_void_p(arr.ctypes.data))

    def get_history(self, row0=0):
        """the archive [rows, d] (from row `row0` on)"""
        rows = C.c_int64()
        self._chk(self.L.dz_get_history(self.h, None, 0, C.byref(rows)))
        if row0:
            Z = np.empty((rows.value - row0, self.d))
            self._chk(self.L.dz_get_history_range(self.h, int(row0), rows.value - row0, _p(Z)))
            return Z
        Z = np.empty((rows.value, self.d))
        self._chk(self.L.dz_get_history(self.h, _p(Z), rows.value, C.byref(rows)))
        return Z

    def history_checksum(self):
        """(64-bit checksum, rows) of the archive as get_history() would return it, made on the device (dz_history_checksum);
        `history_checksum_host` is the same sum in numpy."""
        h, rows = C.c_uint64(), C.c_int64()
        self._chk(self.L.dz_history_checksum(self.h, C.byref(h), C.byref(rows)))
        return int(h.value), int(rows.value)

    def history_rows(self):
        rows = C.c_int64()
        self._chk(self.L.dz_get_history(self.h, None, 0, C.byref(rows)))
        return rows.value

    def get_cr_state(self):
        n = self.cfg.ncr
        p, dm, nu = np.zeros(n), np.zeros(n), np.zeros(n)
        self._chk(self.L.dz_get_cr_state(self.h, _p(p), _p(dm), _p(nu)))
        return p, dm, nu

    def get_gamma_state(self):
        n = self.cfg.ngamma
        p, dm, nu = np.zeros(n), np.zeros(n), np.zeros(n)
        self._chk(self.L.dz_get_gamma_state(self.h, _p(p), _p(dm), _p(nu)))
        return p, dm, nu

    def get_rhat(self):
        r = np.zeros(self.d)
        self._chk(self.L.dz_get_rhat(self.h, _p(r)))
        return r

    def get_chain_moments(self):
        m, v = np.zeros((self.nl, self.d)), np.zeros((self.nl, self.d))
        self._chk(self.L.dz_get_chain_moments(self.h, _p(m), _p(v)))
        return m, v

    def eval_logp(self, X):
        X = _f64(X).reshape(-1, self.d)
        pr, lk = np.zeros(len(X)), np.zeros(len(X))
        self._chk(self.L.dz_eval_logp(self.h, _p(X), len(X), _p(pr), _p(lk)))
        return pr, lk

    def loglike(self, x):
        return float(self.eval_logp(np.asarray(x)[None, :])[1][0])

    def debug_propose(self, chain, gen, phase, base, snooker, cr_idx, delta=1, glev=1):
        n = self.k if phase == 0 else self.k - 1
        base = _f64(base)
        pts = np.zeros((n, self.d)); slogp = np.zeros(n)
        self._chk(self.L.dz_debug_propose(self.h, chain, gen, phase, _p(base), int(snooker), cr_idx, delta, glev, _p(pts), _p(slogp)))
        return pts, slogp, None, None

    # ---- HIP-event kernel timing ----
    def profile_enable(self, on=True, prealloc_pairs=0):
        """prealloc_pairs > 1 pre-creates that many HIP event pairs so the timed region only records them."""
        self._chk(self.L.dz_profile_enable(self.h, (max(2, int(prealloc_pairs)) if prealloc_pairs else 1) if on else 0))

    def profile_reset(self):
        self._chk(self.L.dz_profile_reset(self.h))

    def profile_get_list(self, which, cap=65536):
        """milliseconds of every launch of the class since the last reset, in launch order"""
        ms, n = np.zeros(cap), C.c_int64()
        self._chk(self.L.dz_profile_get_list(self.h, PROFILE_CLASSES[which], _p(ms), cap, C.byref(n)))
        return ms[:min(cap, n.value)]

    def profile_get(self, which):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self.L.dz_profile_get(self.h, PROFILE_CLASSES[which], C.byref(ms), C.byref(n)))
        return ms.value, n.value
