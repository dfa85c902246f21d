#!/usr/bin/env python3
"""bench.py -- proposals/s of the MT-DREAM(ZS) hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one generation: every chain makes one MT-DREAM(ZS) transition (multitry proposals,
reference set, Metropolis accept, Z append every `thin` generations).  Workload at N GPUs (weak
scaling, BASELINE.json north_star / configs[3]): 4096 chains per GPU on the 100-D correlated MVN of
pydream/examples/ndim_gaussian, multitry=5, DE+snooker, reference defaults otherwise.  Inputs
(seed archive, start states, precision matrix) are resident in HBM before the timed region.

For N>1 the driver launches one rank per GPU with torch.distributed.run; torch is used ONLY for the
rendezvous (gloo: barrier, max-reduce of the time, broadcast of the RCCL unique id) -- the engine
itself is libdreamzs.so (HIP + RCCL), loaded through ctypes.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def mvn_precision(d):
    """pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:30-38 at dimension d"""
    i = np.arange(1, d + 1, dtype=float)
    C = (.5 * np.identity(d) + .5 * np.ones((d, d))) * np.sqrt(np.outer(i, i))
    return np.linalg.inv(C)


def setup_engine(Cls, args, n_global, n_local, offset, steps_total, device=0, **extra):
    d, k = args.dim, args.multitry
    rng = np.random.default_rng(args.seed)
    m0 = max(10 * d, 2 * n_global)                        # Dream.py:168-170, core.py:270-273
    Z0 = rng.uniform(-5.0, 15.0, (m0, d))                 # seed archive, iid U(-5,15)
    cap = m0 + n_global * (steps_total // args.thin + 2)
    kw = dict(nchains=n_global, nchains_local=n_local, chain_offset=offset, ndim=d, multitry=k,
              history_thin=args.thin, history_capacity=cap, trace_capacity=max(args.steps, args.warmup, 2),
              seed=args.seed, device=device, adapt_crossover=0, crossover_burnin=0, snooker=args.snooker)
    kw.update(extra)
    e = Cls(**kw)
    e.set_history(Z0)
    e.set_state(Z0[offset:offset + n_local])              # starts = first N seed rows (dream_ex_ndim_gaussian.py:54)
    if args.target == "mvn":
        P = mvn_precision(d)
        if args.mvn_kind == "tri":
            e.set_likelihood_mvn(np.zeros(d), np.linalg.cholesky((P + P.T) / 2).T, 1, 0.0)
        else:
            e.set_likelihood_mvn(np.zeros(d), P, 0, 0.0)
    else:
        mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
        logF = np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi)
        e.set_likelihood_mixture(mu, logF)
    return e


def algorithmic_bytes(args, n_local):
    """Per-launch algorithmic HBM bytes of each kernel class of the multi-kernel path (from
    SURVEY.md section 8(d)'s per-unit figures; rows unpadded, snooker fraction s).  The persistent kernel's figure
    (whole generations, B = 176 d + 152 per chain-generation) is added in main()."""
    d, k, s = args.dim, args.multitry, args.snooker
    row = 8.0 * d
    rows_z = 2 * (1 - s) + 3 * s
    pts = n_local * (2 * k - 1)
    return {
        # average of the two propose launches of a generation: Z gathers + proposal write per point, base row per chain
        "propose": (pts * row * (rows_z + 1.0) + 2.0 * n_local * row) / 2.0,
        # average of the two logp launches: proposal read + 2 scalars written
        "logp": pts * (row + 16.0) / 2.0,
        # accept: state read + write, trace row, append/thin, logp scalars
        "accept": n_local * (row * (2.0 + 1.0 + 1.0 / args.thin) + 16.0 * (2 * k - 1) + 8.0),
    }


def measured_traffic(args, n_local, kernel_class, gens_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same workload
    (profiles/r01_traffic.json, made by tools/collect_profiles.sh + tools/traffic_from_pmc.py), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    default = (n_local == 4096 and args.dim == 100 and args.multitry == 5 and args.target == "mvn" and args.mvn_kind == "tri"
               and args.snooker == 0.1 and args.thin == 10)
    if not (default and kernel_class == "generations" and os.path.exists(path)):
        return None, None
    t = json.load(open(path))
    if "k_generations" not in t["kernel"]:
        return None, None
    return t["bytes_per_generation"] * gens_per_launch, "profiles/r01_traffic.json"


def cpu_baseline(args):
    """The CPU restatement (oracle/dreamzs_oracle.c: scalar C, the chains of a generation spread over the host's cores
    with OpenMP, as the reference spreads them over processes) on a bounded sample of the same workload: the same
    chain count, as many generations as fit in about 10 s."""
    from oracle import oracle as O
    nc = args.cpu_chains
    cores = O.threads()
    probe = 4
    e = setup_engine(O.Engine, args, nc, nc, 0, 8100, schedule=2, trace_capacity=0)      # (the archive is calloc-ed: untouched rows cost nothing)
    e.step(2)
    t0 = time.perf_counter()
    e.step(probe)
    per_gen = (time.perf_counter() - t0) / probe
    g = args.cpu_steps if args.cpu_steps > 0 else int(max(10, min(8000, args.cpu_seconds / per_gen)))
    t0 = time.perf_counter()
    e.step(g)
    dt = time.perf_counter() - t0
    return {"value": nc * args.multitry * g / dt, "unit": "proposals/s", "cores": cores, "kind": "port",
            "sample": "%d chains x %d generations of the same %d-D %s target, multitry=%d, oracle/dreamzs_oracle.c with %d OpenMP "
                      "thread(s) over the chains, %.1f s" % (nc, g, args.dim, args.target, args.multitry, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--spinup", type=int, default=2000,
                    help="untimed generations run before the W warm-up steps so that the GPU leaves its idle clock "
                         "state (a cold MI355X needs ~0.1 s of load; without it short runs vary 3x); more are added, up "
                         "to 4x, until --spinup-seconds of load have passed (small workloads finish 2000 generations sooner)")
    ap.add_argument("--spinup-seconds", type=float, default=0.15)
    ap.add_argument("--chains-per-gpu", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--multitry", type=int, default=5)
    ap.add_argument("--thin", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20260929)
    ap.add_argument("--target", choices=["mvn", "mix3"], default="mvn")
    ap.add_argument("--mvn-kind", choices=["dense", "tri"], default="tri",
                    help="form of the MVN whitening matrix: tri = Cholesky factor of the precision (what "
                         "pydream_amd.likelihoods.MVNormalLogLike builds), dense = full precision matrix")
    ap.add_argument("--snooker", type=float, default=0.1, help="snooker probability (reference default 0.1)")
    ap.add_argument("--cpu-chains", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=0, help="generations of the CPU baseline (0: as many as fit --cpu-seconds)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not time individual kernels with HIP events")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from pydream_amd import _capi            # loads libdreamzs.so (system ROCm runtime) BEFORE torch
    _capi.load_library()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)

    n_local = args.chains_per_gpu
    n_global = n_local * world
    # spin-up (single GPU: extended by time up to 4x) + warm-up + timed pass + HIP-event pass: sizes the replicated archive
    total = (4 if world == 1 else 1) * args.spinup + 2 * args.steps + args.warmup
    # (DZ_BENCH_TRANSPORT=host and DZ_BENCH_DEVICE exist so that the multi-rank control flow can be rehearsed on a
    #  one-GPU box: ranks share the device and exchange through the host; measurements use RCCL, one rank per GPU)
    device = int(os.environ.get("DZ_BENCH_DEVICE", local_rank))
    e = setup_engine(_capi.Engine, args, n_global, n_local, rank * n_local, total, device=device)
    transport_note = None
    if world > 1:
        from pydream_amd.distributed import attach_transport
        transport = os.environ.get("DZ_BENCH_TRANSPORT", "rccl")
        err = ""
        try:
            attach_transport(e, rank, world, transport=transport)
        except Exception as ex:                       # e.g. ncclCommInitRank refused: every rank must learn of it
            err = "%s" % ex
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if any(errs):
            # a number over the host-staged all-gather (named as such in the JSON line) beats no number
            transport_note = "host (rccl unavailable: %s)" % next(x for x in errs if x)
            attach_transport(e, rank, world, transport="host")
        else:
            transport_note = transport

    def barrier():
        e.sync()
        if dist is not None:
            dist.barrier()

    chunk = max(1, min(args.spinup, max(args.steps, args.warmup, 2)))
    spun, t_spin = 0, time.perf_counter()
    # (with several ranks every e.step is a collective sequence, so the count must not depend on a local clock)
    extend = lambda: world == 1 and spun < 4 * args.spinup and time.perf_counter() - t_spin < args.spinup_seconds
    while spun < args.spinup or extend():
        n = min(chunk, (args.spinup if spun < args.spinup else 4 * args.spinup) - spun)      # clock spin-up (untimed, not part of W or K)
        e.trace_reset()
        e.step(n)
        e.sync()
        spun += n
    barrier()
    e.trace_reset()
    e.step(args.warmup)
    barrier()
    e.trace_reset()
    barrier()
    # ---- timed region: exactly K generations, nothing else on the stream ----
    t0 = time.perf_counter()
    e.step(args.steps)
    t_enqueued = time.perf_counter() - t0
    e.sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    tr = e.get_trace(0, args.steps, with_X=False)
    acc = float(tr["moved"].mean())
    rhat = e.get_rhat() if world == 1 else None
    if world > 1:
        from pydream_amd.distributed import gelman_rubin_sharded
        rhat = gelman_rubin_sharded(e, args.steps)

    # ---- second pass of K generations with HIP events around every kernel launch (per-kernel durations
    #      for the roofline; kept out of the timed region because each event record costs a few us) ----
    prof = {}
    if not args.no_events:
        e.trace_reset()
        e.profile_enable(True, prealloc_pairs=8 * args.steps)
        e.profile_reset()
        e.step(args.steps)
        e.sync()
        e.profile_enable(False)
        # propose / logp / accept launches carry their own start/stop events (hipExtLaunchKernelGGL: the dispatch's
        # begin/end timestamps, the same figures rocprofv3's kernel trace reports); adapt / exchange / generations
        # are bracketed by event records, which adds about `event_bracket_us` to each of those
        for name in ("generations", "propose", "logp", "accept", "adapt", "exchange"):
            ms, n = e.profile_get(name)
            prof[name] = {"total_ms": ms, "launches": n, "avg_us": (1e3 * ms / n) if n else None}
        ms0, n0 = e.profile_get("empty")
        prof["event_bracket_us"] = (1e3 * ms0 / n0) if n0 else None
        if dist is not None:
            dist.barrier()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = n_global * args.multitry * args.steps / dt
    out = {
        "metric": "proposals/sec (all chains), 100D MVN logpdf, MT-DREAM(ZS) multitry=%d" % args.multitry,
        "value": value, "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d chains/GPU x %d-D %s target (%s), multitry=%d, DE+snooker(%g), nCR=3, history_thin=%d, "
                               "seed archive max(10d,2N) rows U(-5,15); BASELINE north_star target / configs[3] per-GPU shard"
                               % (n_local, args.dim, "correlated MVN" if args.target == "mvn" else "3-Gaussian mixture",
                                  args.mvn_kind if args.target == "mvn" else "identity cov", args.multitry, args.snooker, args.thin),
                   "chains_global": n_global, "chains_per_gpu": n_local, "ndim": args.dim, "multitry": args.multitry,
                   "parallelism": ("chains sharded x%d, Z appends all-gathered, transport: %s" % (world, transport_note))
                                  if world > 1 else "single GPU"},
        "logp_points_per_s": n_global * (2 * args.multitry - 1) * args.steps / dt,
        "host_enqueue_ms_per_step": 1e3 * t_enqueued / args.steps, "spinup_generations": spun,
        "acceptance_rate": acc, "rhat_max": float(np.max(rhat)),
    }
    if prof:
        ab = algorithmic_bytes(args, n_local)
        if prof.get("generations", {}).get("launches"):
            # the persistent kernel covers whole generations: SURVEY.md section 8(d) B = 176 d + 152 bytes per chain-generation
            ab["generations"] = n_local * (176.0 * args.dim + 152.0) * args.steps / prof["generations"]["launches"]
        cand = {k: v for k, v in prof.items() if k in ab and v["launches"]}
        dom = max(cand, key=lambda k: cand[k]["total_ms"])
        avg_s = cand[dom]["avg_us"] * 1e-6
        achieved = ab[dom] / avg_s / 1e9
        traffic, tsrc = measured_traffic(args, n_local, dom, args.steps / cand[dom]["launches"])
        out["roofline"] = {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                           "algorithmic_bytes_per_launch": ab[dom], "avg_launch_us": avg_s * 1e6}
        out["kernel_times"] = prof
        gen_bytes = n_local * (176.0 * args.dim + 152.0)        # B = 176 d + 152 per chain-generation
        out["generation_hbm"] = {"algorithmic_bytes_per_generation": gen_bytes,
                                 "achieved_GBps": gen_bytes * args.steps / dt / 1e9,
                                 "frac_of_8TBps": gen_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
