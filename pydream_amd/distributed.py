"""Multi-GPU runs: one process per GPU, chains sharded contiguously, Z / positions replicated.

Rank r owns chains [r*N/W, (r+1)*N/W).  Every rank holds the whole Z archive; at the end of an
appending generation each rank contributes its N/W rows and an all-gather puts them into Z's tail
in global chain order (this replaces the reference's lock-protected multiprocessing arrays,
pydream/core.py:281-297, pydream/Dream.py:919-938).  During crossover burn-in the published
positions are all-gathered the same way (Dream.py:424-449) and every rank evaluates the adaptation
statistics of ALL chains redundantly, so the adapted probabilities are bit-identical everywhere and
independent of W.  Random streams are keyed by the GLOBAL chain id, so results do not depend on W.

Transports for the all-gather:
  * "rccl"  -- ncclAllGather on device buffers over xGMI inside libdreamzs.so (production);
  * "host"  -- a host-staged all-gather through torch.distributed (gloo); used by the CPU tests
               and on boxes with fewer GPUs than ranks.
torch.distributed is used only as the control plane (rendezvous, unique-id broadcast, host exchange).
"""
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


def _dist():
    import torch.distributed as dist
    if not dist.is_initialized():
        raise Exception('torch.distributed is not initialised (init_process_group first)')
    return dist


class HostExchange:
    """all-gather of equal-sized byte blocks through torch.distributed CPU tensors."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, send, nbytes):
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
    if transport == "rccl":
        dist = _dist()
        ids = [_capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, group=group)
        engine.comm_init_rccl(rank, world, ids[0])
    elif transport == "host":
        engine.set_exchange(HostExchange(group))
    else:
        raise ValueError("transport must be 'rccl' or 'host'")


def broadcast_seed(seed, group=None):
    dist = _dist()
    box = [seed]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


def run_dream_sharded(parameters, likelihood, nchains=8, niterations=1000, start=None, verbose=False, nverbose=10,
                      transport="rccl", device=None, engine_cls=None, group=None, **kwargs):
    """run_dream with the chains sharded over the ranks of the (already initialised) process group.

    Returns this rank's slice: (sampled_params, log_ps) lists for chains
    [rank*N/W, (rank+1)*N/W), in the same format as run_dream.  `seed` must be the same on all
    ranks (pass it, or leave it None to have rank 0 draw and broadcast one)."""
    import os
    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
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
                                chain_offset=off, nchains_local=nl, engine_cls=engine_cls)
    try:
        attach_transport(pool.engine, rank, world, transport, group)
        save = step.save_history
        step.save_history = save and rank == 0          # the archive is replicated: one writer is enough
        return _sample_dream_batched(pool.engine, step, niterations, verbose and rank == 0, nverbose)
    finally:
        pool.close()
        pool.join()


def gelman_rubin_sharded(engine, nsamples, group=None):
    """R-hat (pydream/convergence.py:3-20) over ALL chains of a sharded run: per-chain second-half
    moments are computed on each GPU and all-gathered; the final [d] reduction runs on the host."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    mean, var = engine.get_chain_moments()
    loc = torch.from_numpy(np.stack([mean, var]))
    out = [torch.empty_like(loc) for _ in range(world)]
    dist.all_gather(out, loc, group=group)
    means = np.concatenate([o[0].numpy() for o in out])
    vars_ = np.concatenate([o[1].numpy() for o in out])
    W = np.mean(vars_, axis=0)
    B = np.var(means, axis=0)
    return np.sqrt((W * (1 - 1. / nsamples) + B) / W)
