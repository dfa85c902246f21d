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
import pickle
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


class SocketGroup:
    """A process group for ONE node over TCP, rank 0 as the hub: all_gather_object / broadcast_object / barrier / all_reduce_max --
    all the control plane of a sharded run needs.  Every collective is: each rank sends its pickled contribution to the hub, the hub
    answers everybody with the list in rank order."""

    @property
    def group(self):
        return self

    def __init__(self, rank, world, addr="127.0.0.1", port=29500, timeout=600.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = []
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                r = struct.unpack("<i", self._recvn(c, 4))[0]
                conns[r] = c
            srv.close()
            self.peers = [conns[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    c = socket.create_connection((addr, int(port)), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            c.sendall(struct.pack("<i", self.rank))
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
        b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        c.sendall(struct.pack("<q", len(b)) + b)

    @classmethod
    def _recv(cls, c):
        n = struct.unpack("<q", cls._recvn(c, 8))[0]
        return pickle.loads(cls._recvn(c, n))

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
        for c in self.peers + ([self.hub] if self.rank else []):
            try:
                c.close()
            except OSError:
                pass


def socket_group_from_env(timeout=600.0):
    """The SocketGroup of a job started the torch.distributed.run way: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
    environment, one process per rank on ONE node.  MASTER_PORT itself belongs to the launcher's own store, so rank 0 listens on a port
    the system picks and leaves its number in a file under /tmp keyed by MASTER_PORT and the launcher's pid; the others read it (and
    read it again if nothing answers there: a leftover of an earlier job)."""
    import os
    import tempfile
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return SocketGroup(0, 1)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if addr in ("localhost",):
        addr = "127.0.0.1"
    key = os.path.join(tempfile.gettempdir(), "dreamzs_rdv_%s_%d.port" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
    if rank == 0:
        probe = socket.socket()
        probe.bind((addr, 0))
        port = probe.getsockname()[1]
        probe.close()
        tmp = key + ".%d" % os.getpid()
        with open(tmp, "w") as f:
            f.write("%d" % port)
        os.replace(tmp, key)
        try:
            return SocketGroup(0, world, addr, port, timeout)
        finally:
            try:
                os.unlink(key)
            except OSError:
                pass
    deadline = time.time() + timeout
    while True:
        try:
            port = int(open(key).read())
            g = SocketGroup.__new__(SocketGroup)
            g.rank, g.world, g.peers = rank, world, []
            c = socket.create_connection((addr, port), timeout=2.0)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            c.sendall(struct.pack("<i", rank))
            g.hub = c
            return g
        except (OSError, ValueError):
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
    if transport == "peer":
        blobs = _gather_objects(engine.peer_export(), group)
        engine.peer_attach(rank, world, b"".join(blobs))
        _barrier(group)                             # nobody pushes before everybody has mapped everybody
    elif transport == "rccl":
        uid = _gather_objects(_capi.comm_unique_id() if rank == 0 else None, group)[0]
        engine.comm_init_rccl(rank, world, uid)
    elif transport == "host":
        engine.set_exchange(HostExchange(group))
    else:
        raise ValueError("transport must be 'peer', 'rccl' or 'host'")


def broadcast_seed(seed, group=None):
    return _gather_objects(seed, group)[0]


def run_dream_sharded(parameters, likelihood, nchains=8, niterations=1000, start=None, verbose=False, nverbose=10,
                      transport="rccl", device=None, engine_cls=None, group=None, **kwargs):
    """run_dream with the chains sharded over the ranks of the (already initialised) process group.

    Returns this rank's slice: (sampled_params, log_ps) lists for chains
    [rank*N/W, (rank+1)*N/W), in the same format as run_dream.  `seed` must be the same on all
    ranks (pass it, or leave it None to have rank 0 draw and broadcast one)."""
    import os
    rank, world = _rank_world(group)
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
                                chain_offset=off, nchains_local=nl, engine_cls=engine_cls, history_lag=kwargs.get('history_lag', 0))
    try:
        attach_transport(pool.engine, rank, world, transport, group)
        save = step.save_history
        step.save_history = save and rank == 0          # the archive is replicated: one writer is enough
        out = _sample_dream_batched(pool.engine, step, niterations, verbose and rank == 0, nverbose)
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
