"""Multi-GPU runs: one process per GPU, chains sharded contiguously, Z / positions replicated.

Rank r owns chains [r*N/W, (r+1)*N/W).  Every rank holds the whole Z archive; at the end of an
appending generation each rank contributes its N/W rows and an all-gather puts them into Z's tail
in global chain order (this replaces the reference's lock-protected multiprocessing arrays,
pydream/core.py:281-297, pydream/Dream.py:919-938).  During crossover burn-in the published
positions are all-gathered the same way (Dream.py:424-449) and every rank evaluates the adaptation
statistics of ALL chains redundantly, so the adapted probabilities are bit-identical everywhere and
independent of W.  Random streams are keyed by the GLOBAL chain id, so results do not depend on W.

Transports for the all-gather:
  * "peer"  -- every rank maps the other ranks' buffers (HIP IPC) and its copy engines push its rows into them; with
               ``history_lag=1`` the transfer hides behind the next thin-cycle (include/dreamzs.h dz_peer_attach);
  * "rccl"  -- ncclAllGather on device buffers over xGMI inside libdreamzs.so, between two launches;
  * "host"  -- a host-staged all-gather through torch.distributed (gloo); used by the CPU tests
               and on boxes with fewer GPUs than ranks.
The control plane (rendezvous, handle / unique-id exchange, the host transport) is either torch.distributed (gloo) -- pass its
process group, or None for the default one -- or the built-in ``SocketGroup`` below (plain TCP on one node, no torch in the process:
what bench.py and the GPU tests use; importing torch costs minutes on a freshly started box).
"""
import os
import socket
import struct
import time

import numpy as np

from . import _capi
from .Dream import Dream
from .core import _sample_dream_batched, _setup_mp_dream_pool
from .model import Model


def shard(nchains, rank, world):
    """(chain_offset, nchains_local) of `rank`; chains must divide evenly."""
    if nchains % world:
        raise Exception('nchains (%d) must be a multiple of the number of ranks (%d)' % (nchains, world))
    nl = nchains // world
    return rank * nl, nl


# ---- wire format of the control plane ------------------------------------------------------------------------------------------
# What the ranks exchange is a handful of plain values: None (a barrier), byte strings (IPC blobs, the RCCL unique id, staged rows),
# text (error messages), integers (seeds, checksums), floats (block times), float/integer arrays (per-chain moments) and lists /
# tuples of those.  They travel in a fixed tagged framing -- nothing on the wire can name code (no pickle).
_MAX_MSG = 1 << 32
_ARRAY_DTYPES = ("<f8", "<f4", "<i8", "<i4", "<u8", "<u4", "|u1", "|i1", "|b1")


def _encode(obj, out, depth=0):
    if depth > 8:
        raise ValueError("control plane: value nested too deeply")
    if obj is None:
        out += b"N"
    elif isinstance(obj, (bool, np.bool_)):
        out += b"T" if obj else b"f"
    elif isinstance(obj, (int, np.integer)):
        out += b"I" + int(obj).to_bytes(16, "little", signed=True)
    elif isinstance(obj, (float, np.floating)):
        out += b"F" + struct.pack("<d", float(obj))
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out += b"B" + struct.pack("<q", len(b)) + b
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out += b"S" + struct.pack("<q", len(b)) + b
    elif isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        dt = a.dtype.newbyteorder("<").str if a.dtype.byteorder == ">" else a.dtype.str
        if dt not in _ARRAY_DTYPES:
            raise TypeError("control plane: arrays of dtype %s do not travel" % a.dtype)
        a = a.astype(dt, copy=False)
        out += b"A" + dt.encode("ascii").ljust(4) + struct.pack("<b", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape) + a.tobytes()
    elif isinstance(obj, (list, tuple)):
        out += (b"L" if isinstance(obj, list) else b"U") + struct.pack("<q", len(obj))
        for x in obj:
            _encode(x, out, depth + 1)
    else:
        raise TypeError("control plane: values of type %s do not travel" % type(obj).__name__)


def _decode(buf, pos=0, depth=0):
    if depth > 8:
        raise ValueError("control plane: value nested too deeply")
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag in (b"T", b"f"):
        return tag == b"T", pos
    if tag == b"I":
        return int.from_bytes(buf[pos:pos + 16], "little", signed=True), pos + 16
    if tag == b"F":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"B", b"S"):
        n = struct.unpack_from("<q", buf, pos)[0]
        pos += 8
        if n < 0 or pos + n > len(buf):
            raise ValueError("control plane: bad length")
        b = bytes(buf[pos:pos + n])
        return (b if tag == b"B" else b.decode("utf-8")), pos + n
    if tag == b"A":
        dt = bytes(buf[pos:pos + 4]).decode("ascii").strip()
        if dt not in _ARRAY_DTYPES:
            raise ValueError("control plane: bad array type")
        nd = struct.unpack_from("<b", buf, pos + 4)[0]
        if not 0 <= nd <= 8:
            raise ValueError("control plane: bad array rank")
        shape = struct.unpack_from("<%dq" % nd, buf, pos + 5)
        pos += 5 + 8 * nd
        if any(x < 0 for x in shape):
            raise ValueError("control plane: bad array shape")
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize if nd else np.dtype(dt).itemsize
        if pos + nbytes > len(buf):
            raise ValueError("control plane: bad array length")
        return np.frombuffer(buf, dtype=dt, count=nbytes // np.dtype(dt).itemsize, offset=pos).reshape(shape).copy(), pos + nbytes
    if tag in (b"L", b"U"):
        n = struct.unpack_from("<q", buf, pos)[0]
        pos += 8
        if n < 0 or n > len(buf):
            raise ValueError("control plane: bad count")
        items = []
        for _ in range(n):
            x, pos = _decode(buf, pos, depth + 1)
            items.append(x)
        return (items if tag == b"L" else tuple(items)), pos
    raise ValueError("control plane: unknown tag %r" % tag)


_HELLO = b"DZRDV1"
_TOKEN_BYTES = 32


class SocketGroup:
    """A process group for ONE node over loopback TCP, rank 0 as the hub: all_gather_object / broadcast_object / barrier /
    all_reduce_max -- all the control plane of a sharded run needs.  Every collective is: each rank sends its contribution to the hub,
    the hub answers everybody with the list in rank order.

    The hub listens on 127.0.0.1 only.  A connection counts only if it opens with the job's `token` (32 random bytes every rank got
    from the launcher: DZ_RDV_TOKEN, or the rendezvous file of `socket_group_from_env`) and a rank number in 1..world-1 that has not
    connected yet; anything else is dropped.  Values travel in the tagged framing above, never as pickles."""

    @property
    def group(self):
        return self

    def __init__(self, rank, world, addr="127.0.0.1", port=29500, timeout=600.0, token=None, listener=None, connect_timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = []
        self.hub = None
        if self.world == 1:                # (a single rank needs no token: nothing connects)
            return
        token = _token_bytes(token)
        if addr not in ("127.0.0.1", "localhost"):
            raise ValueError("SocketGroup is a one-node control plane: it binds and connects on 127.0.0.1 only (got %r)" % (addr,))
        addr = "127.0.0.1"
        deadline = time.time() + (timeout if connect_timeout is None else connect_timeout)
        if self.rank == 0:
            srv = listener
            if srv is None:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, int(port)))
                srv.listen(max(8, 2 * self.world))
            conns = {}
            try:
                while len(conns) < self.world - 1:
                    srv.settimeout(max(0.1, deadline - time.time()))
                    c, _ = srv.accept()
                    try:
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(10.0)
                        hello = self._recvn(c, len(_HELLO) + 4 + _TOKEN_BYTES)
                        r = struct.unpack_from("<i", hello, len(_HELLO))[0]
                        import hmac
                        ok = hello[:len(_HELLO)] == _HELLO and hmac.compare_digest(hello[len(_HELLO) + 4:], token) and 1 <= r < self.world and r not in conns
                    except (OSError, ConnectionError, struct.error):
                        ok = False
                    if not ok:                  # a stranger, a wrong token, a rank out of range or already seated
                        c.close()
                        continue
                    c.sendall(b"\x01")
                    c.settimeout(timeout)
                    conns[r] = c
            finally:
                srv.close()
            self.peers = [conns[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    c = socket.create_connection((addr, int(port)), timeout=5.0)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(timeout)
                    c.sendall(_HELLO + struct.pack("<i", self.rank) + token)
                    if self._recvn(c, 1) != b"\x01":
                        raise ConnectionError("rendezvous refused")
                    break
                except (OSError, ConnectionError):
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            self.hub = c

    @staticmethod
    def _recvn(c, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("peer closed the control connection")
            buf += chunk
        return bytes(buf)

    @classmethod
    def _send(cls, c, obj):
        b = bytearray()
        _encode(obj, b)
        c.sendall(struct.pack("<q", len(b)) + b)

    @classmethod
    def _recv(cls, c):
        n = struct.unpack("<q", cls._recvn(c, 8))[0]
        if n < 1 or n > _MAX_MSG:
            raise ConnectionError("control plane: bad message length")
        buf = cls._recvn(c, n)
        obj, pos = _decode(buf)
        if pos != n:
            raise ConnectionError("control plane: trailing bytes")
        return obj

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [self._recv(c) for c in self.peers]
            for c in self.peers:
                self._send(c, out)
            return out
        self._send(self.hub, obj)
        return self._recv(self.hub)

    def broadcast_object(self, obj, src=0):
        return self.all_gather_object(obj if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather_object(None)

    def all_reduce_max(self, values):
        rows = self.all_gather_object([float(v) for v in values])
        return [max(r[i] for r in rows) for i in range(len(values))]

    def close(self):
        for c in self.peers + ([self.hub] if self.hub is not None else []):
            try:
                c.close()
            except OSError:
                pass


def _token_bytes(token):
    """the job's rendezvous token as 32 bytes: given (bytes or hex text), or DZ_RDV_TOKEN from the launcher"""
    import os
    if token is None:
        token = os.environ.get("DZ_RDV_TOKEN")
    if token is None:
        raise ValueError("SocketGroup needs the job's rendezvous token (token=..., or DZ_RDV_TOKEN in the environment)")
    if isinstance(token, str):
        token = bytes.fromhex(token)
    if len(token) != _TOKEN_BYTES:
        raise ValueError("the rendezvous token must be %d bytes" % _TOKEN_BYTES)
    return bytes(token)


def new_token():
    """a fresh rendezvous token (hex text) for a launcher to hand to its ranks as DZ_RDV_TOKEN"""
    import os
    return os.urandom(_TOKEN_BYTES).hex()


def _rendezvous_dir():
    """a directory only this user can read or write: $XDG_RUNTIME_DIR/dreamzs or <tmp>/dreamzs-<uid>, created 0700, refused if it is
    a symlink, someone else's, or open to others"""
    import os
    import stat
    import tempfile
    base = os.environ.get("XDG_RUNTIME_DIR")
    path = os.path.join(base, "dreamzs") if base and os.path.isdir(base) else os.path.join(tempfile.gettempdir(), "dreamzs-%d" % os.getuid())
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise Exception("rendezvous directory %s is not a private directory of this user" % path)
    return path


def socket_group_from_env(timeout=600.0):
    """The SocketGroup of a job started the torch.distributed.run way: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
    environment, one process per rank on ONE node.  MASTER_PORT itself belongs to the launcher's own store, so rank 0 opens a
    listening socket on a loopback port the system picks -- and keeps THAT socket: no close-and-bind-again window -- and leaves the
    port and the job's token in a file the other ranks read.  The file lives in a directory private to the user (`_rendezvous_dir`),
    is created with O_EXCL | O_NOFOLLOW under a name keyed by MASTER_PORT and the launcher's pid, and is removed once everybody is
    seated.  A launcher that sets DZ_RDV_TOKEN and DZ_RDV_PORT itself (bench.py's own) needs no file at all."""
    import os
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return SocketGroup(0, 1, token=b"\0" * _TOKEN_BYTES)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if addr not in ("127.0.0.1", "localhost"):
        raise Exception("the socket control plane is for one node: MASTER_ADDR must be 127.0.0.1 (got %s); use --control torch across nodes" % addr)
    if os.environ.get("DZ_RDV_TOKEN") and os.environ.get("DZ_RDV_PORT"):
        return SocketGroup(rank, world, "127.0.0.1", int(os.environ["DZ_RDV_PORT"]), timeout)
    key = os.path.join(_rendezvous_dir(), "rdv_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.bind(("127.0.0.1", 0))
        srv.listen(max(8, 2 * world))
        port = srv.getsockname()[1]
        token = os.urandom(_TOKEN_BYTES)
        try:
            os.unlink(key)                       # (a leftover of an earlier job with the same launcher pid: ours to remove, the directory is private)
        except OSError:
            pass
        fd = os.open(key, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as f:
            f.write("%d %s\n" % (port, token.hex()))
        try:
            return SocketGroup(0, world, "127.0.0.1", port, timeout, token=token, listener=srv)
        finally:
            try:
                os.unlink(key)
            except OSError:
                pass
    deadline = time.time() + timeout
    while True:
        try:
            fd = os.open(key, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            with os.fdopen(fd) as f:
                txt = f.read().split()
            port, token = int(txt[0]), bytes.fromhex(txt[1])
            return SocketGroup(rank, world, "127.0.0.1", port, timeout, token=token, connect_timeout=min(timeout, 10.0))
        except (OSError, ValueError, IndexError, ConnectionError):
            if time.time() > deadline:
                raise Exception("no rendezvous with rank 0 (%s)" % key)
            time.sleep(0.05)


def _dist():
    import torch.distributed as dist
    if not dist.is_initialized():
        raise Exception('torch.distributed is not initialised (init_process_group first)')
    return dist


def _rank_world(group):
    if isinstance(group, SocketGroup):
        return group.rank, group.world
    dist = _dist()
    return dist.get_rank(group), dist.get_world_size(group)


def _gather_objects(obj, group):
    """every rank's object, in rank order"""
    if isinstance(group, SocketGroup):
        return group.all_gather_object(obj)
    dist = _dist()
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def _barrier(group):
    if isinstance(group, SocketGroup):
        group.barrier()
    else:
        _dist().barrier(group=group)


class HostExchange:
    """all-gather of equal-sized byte blocks through the control plane (torch.distributed CPU tensors, or the socket group)."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, send, nbytes):
        if isinstance(self.group, SocketGroup):
            return b"".join(self.group.all_gather_object(bytes(send)))
        import torch
        dist = _dist()
        world = dist.get_world_size(self.group)
        src = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        out = torch.empty(world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, src, group=self.group)
        return out.numpy().tobytes()


def attach_transport(engine, rank, world, transport="rccl", group=None):
    """Give a sharded engine its all-gather."""
    if world == 1:
        return
    if transport not in ("peer", "rccl", "host"):
        raise ValueError("transport must be 'peer', 'rccl' or 'host'")
    # Every rank makes the SAME collectives whether its own attach works or not -- (1) the bootstrap data, (2) the outcome -- and
    # every rank raises if any rank failed: a rank that left early would otherwise pair its next collective with the others' (2).
    err, boot = "", None
    try:
        boot = engine.peer_export() if transport == "peer" else (_capi.comm_unique_id() if transport == "rccl" and rank == 0 else None)
    except Exception as ex:
        err = "%s" % ex
    # ... and (advisor, round 5) what decides WHAT the ranks exchange must be the same everywhere: the sampler's configuration, and the
    # environment switches the engine read at dz_create (whether the burn-in exchanges group sums or positions, how many appends a launch
    # holds) -- ranks that disagree would issue all-gathers of different sizes and kinds, which hangs or corrupts silently
    cfg = getattr(engine, "cfg", None)
    sig = tuple(int(getattr(cfg, f)) for f in ("nchains", "nchains_local", "ndim", "multitry", "depairs", "ncr", "ngamma", "history_thin", "crossover_burnin",
                                                "adapt_crossover", "adapt_gamma", "history_lag", "adapt_lag", "seed") if cfg is not None and hasattr(cfg, f)) + \
        tuple(os.environ.get(v, "") for v in ("DZ_ADAPT_GROUPS", "DZ_ADAPT_FUSED", "DZ_MEGA", "DZ_MEGA_BURNIN"))
    got = _gather_objects((boot if not err else None, sig), group)
    boots = [g[0] for g in got]
    if not err and any(g[1] != sig for g in got):
        err = "the ranks' engines were configured differently (sampler options or DZ_* environment switches): %r on rank %d, %r here" % (
            next(g[1] for g in got if g[1] != sig), next(i for i, g in enumerate(got) if g[1] != sig), sig)
    if not err:
        try:
            if transport == "peer":
                if any(b is None for b in boots):
                    raise Exception("a rank could not export its buffers")
                engine.peer_attach(rank, world, b"".join(boots))
            elif transport == "rccl":
                if boots[0] is None:
                    raise Exception("rank 0 could not make the RCCL unique id")
                engine.comm_init_rccl(rank, world, boots[0])
            else:
                engine.set_exchange(HostExchange(group))
        except Exception as ex:
            err = "%s" % ex
    errs = _gather_objects(err, group)              # (also the barrier: nobody pushes before everybody has mapped everybody)
    if any(errs):
        r = next(i for i, x in enumerate(errs) if x)
        if transport == "peer" and not err:
            try:
                engine.peer_detach()                # (ranks on which it did attach must not keep using it)
            except Exception:
                pass
        raise Exception("%s transport: rank %d: %s" % (transport, r, errs[r]))


def broadcast_seed(seed, group=None):
    return _gather_objects(seed, group)[0]


def run_dream_sharded(parameters, likelihood, nchains=8, niterations=1000, start=None, verbose=False, nverbose=10,
                      transport="rccl", device=None, engine_cls=None, group=None, restart=False, **kwargs):
    """run_dream with the chains sharded over the ranks of the (already initialised) process group.

    Returns this rank's slice: (sampled_params, log_ps) lists for chains
    [rank*N/W, (rank+1)*N/W), in the same format as run_dream.  `seed` must be the same on all
    ranks (pass it, or leave it None to have rank 0 draw and broadcast one).
    restart=True continues an earlier run the way run_dream does (pydream/core.py:46-62): every rank loads the history and the adapted
    crossover / gamma-level probabilities from the three files `model_name` names (written by rank 0 of the run before: the archive is
    replicated) and takes its own slice of `start` -- a list of nchains vectors, the last states of ALL chains.  The files are read by
    EVERY rank: they must lie on a filesystem all ranks see (one node: any directory; several nodes: a shared one)."""
    import os
    rank, world = _rank_world(group)
    if restart:
        if start is None:
            raise Exception('Restart run specified but no start positions given.')
        if 'model_name' not in kwargs:
            raise Exception('Restart run specified but no model name to load history and crossover value files from given.')
        kwargs = dict(kwargs, history_file=kwargs['model_name'] + '_DREAM_chain_history.npy',
                      crossover_file=kwargs['model_name'] + '_DREAM_chain_adapted_crossoverprob.npy',
                      gamma_file=kwargs['model_name'] + '_DREAM_chain_adapted_gammalevelprob.npy')
    off, nl = shard(nchains, rank, world)
    seed = kwargs.pop('seed', None)
    if seed is None:
        seed = broadcast_seed(int.from_bytes(os.urandom(8), 'little') if rank == 0 else None, group)
    if type(parameters) is not list:
        parameters = [parameters]
    model = Model(likelihood=likelihood, sampled_parameters=parameters)
    step = Dream(model=model, variables=parameters, verbose=verbose, **kwargs)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    pool = _setup_mp_dream_pool(nchains, niterations, step, start_pt=start, seed=seed, device=device,
                                chain_offset=off, nchains_local=nl, engine_cls=engine_cls, history_lag=kwargs.get('history_lag', 0),
                                adapt_lag=kwargs.get('adapt_lag', 0))
    try:
        attach_transport(pool.engine, rank, world, transport, group)
        save = step.save_history
        step.save_history = save and rank == 0          # the archive is replicated: one writer is enough
        # (advisor, round 5) a rank that fails -- rank 0 writing the files to a full disk, a likelihood raising on one shard -- must not
        # leave the others waiting in a barrier: every rank reports its outcome (the gather is the barrier) and every rank raises if any failed
        failure, out = "", None
        try:
            out = _sample_dream_batched(pool.engine, step, niterations, verbose and rank == 0, nverbose)
        except Exception as ex:
            if world == 1:
                raise
            failure = "%s: %s" % (type(ex).__name__, ex)
        if world > 1:
            outcomes = _gather_objects(failure, group)   # also: the files are complete before any rank returns (a restart reads them on every rank)
            if any(outcomes):
                r = next(i for i, x in enumerate(outcomes) if x)
                raise Exception("run_dream_sharded: rank %d failed: %s" % (r, outcomes[r]))
        if transport == "peer" and world > 1:
            pool.engine.sync()
            _barrier(group)                             # a rank's buffers stay mapped until no peer can still be writing into them
        return out
    finally:
        pool.close()
        pool.join()


def gelman_rubin_sharded(engine, nsamples, group=None):
    """R-hat (pydream/convergence.py:3-20) over ALL chains of a sharded run: per-chain second-half
    moments are computed on each GPU and all-gathered; the final [d] reduction runs on the host."""
    mean, var = engine.get_chain_moments()
    if isinstance(group, SocketGroup):
        out = group.all_gather_object((mean, var))
        means = np.concatenate([o[0] for o in out])
        vars_ = np.concatenate([o[1] for o in out])
    else:
        import torch
        dist = _dist()
        world = dist.get_world_size(group)
        loc = torch.from_numpy(np.stack([mean, var]))
        out = [torch.empty_like(loc) for _ in range(world)]
        dist.all_gather(out, loc, group=group)
        means = np.concatenate([o[0].numpy() for o in out])
        vars_ = np.concatenate([o[1].numpy() for o in out])
    W = np.mean(vars_, axis=0)
    B = np.var(means, axis=0)
    return np.sqrt((W * (1 - 1. / nsamples) + B) / W)
