#!/usr/bin/env python3
"""bench.py -- proposals/s of the MT-DREAM(ZS) hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one generation: every chain makes one MT-DREAM(ZS) transition (multitry proposals,
reference set, Metropolis accept, Z append every `thin` generations).  Workload at N GPUs (weak
scaling, BASELINE.json north_star / configs[3]): 4096 chains per GPU on the 100-D correlated MVN of
pydream/examples/ndim_gaussian, multitry=5, DE+snooker, reference defaults otherwise.  Inputs
(seed archive, start states, precision matrix) are resident in HBM before the timed region.

Sequence (every part but the timed blocks is untimed):
  1. convergence run from the over-dispersed starts, in chunks, with the whole run traced on the device: after every
     chunk R-hat (pydream/convergence.py:3-20: second half of the run so far) over all chains; reports the first
     generation count at which max R-hat < 1.2 (the reference example's stopping rule, dream_ex_ndim_gaussian.py:80)
     -- this also takes the GPU out of its idle clock state;
  2. R-hat over a fixed window of --rhat-window generations after that (independent of --steps);
  3. W warm-up generations, then blocks of EXACTLY K generations, each bracketed by barrier + device sync on both sides
     and timed on its own (max over ranks), repeated until at least --min-timed-ms have been timed:
     value = N k K / median block time (a single block at the default K lasts 30 ms, at K = 20 0.6 ms);
  4. a pass of max(K, 200) generations with HIP events on every launch (per-kernel durations for the roofline: the median of
     at least 20 launches of the persistent kernel, capped by what the timed blocks allow);
  5. (one GPU) the same timed blocks with the reference example's own formula, the dense precision matrix: `dense_value`;
  6. (rank 0) the CPU restatement on the host cores.

For N>1 there is one rank per GPU: either a launcher started them (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment), or -- `python bench.py --gpus N` as ONE process, no WORLD_SIZE -- bench.py starts its own N ranks
(rank i on device i) and fails if the node has fewer than N devices: it never runs fewer GPUs than it was asked for.  The ranks
rendezvous over loopback TCP (pydream_amd.distributed.SocketGroup: barrier, max-reduce of the block times, exchange of the IPC
handles / the RCCL unique id) -- no torch in the process (its first import on a fresh box takes minutes); --control torch uses
torch.distributed/gloo instead.  The engine itself is libdreamzs.so (HIP; copy-engine peer pushes or RCCL), through ctypes.
After the timed blocks of an N>1 run every rank reduces its replica of the archive to a 64-bit checksum on the device
(dz_history_checksum); the line carries "replicas_identical" and the exit code is 3 if they are not.
history_lag (appended rows become sampleable L appends late) is 3 at EVERY N and a launch of the persistent kernels holds two history
appends (20 generations) at EVERY N (round 6; --appends-per-launch), so that the 1 -> 8 curve is one algorithm AND one launch structure:
with the copy engines pushing a launch's rows while the next launch computes, no generation of that next launch may sample them, which
two appends per launch meet from lag 3 on (rounds 3-5 ran lag 1: 20 generations per launch on one GPU, 10 on several).  At N = 1 the
lockstep schedule (lag 0, 10 generations per launch) is timed beside it: `value_history_lag0`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # several ranks on one node: each rank's BLAS would otherwise start a thread per core the box SHOWS (256 on the GPU boxes, of which
    # the job's cgroup grants 16) -- eight ranks inverting a 1000 x 1000 matrix at set-up then take minutes instead of a second
    for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_v, "2")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6 # v_mfma_f64_16x16x4_f64: 64 cycles per instruction and SIMD (tools/micro/mfma_f64_rate.hip measured 77.4 TFLOP/s);
                             # equal to the FP64 vector rate -- the guide's table has no FP64 row
PROFILE_TAG = next((t for t in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", t + "_pmc_summary.json"))), "r04")
                             # profiles/<tag>_traffic.json, profiles/<tag>_pmc_summary.json (tools/collect_profiles.sh)
STRONG_CHAINS = 32768        # BASELINE configs[3]: "8xMI355X: 32768 chains, 100D MVN ... report 1/2/4/8 scaling"
ENGINE_CLOCK_HZ = 2.4e9      # MI355X peak engine clock (MI355X_MICROARCH.md); the headline kernel's cycle stamps give 2.35 GHz under load


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
    if getattr(args, "history_lag", 0):
        kw["history_lag"] = int(args.history_lag)
    if getattr(args, "adapt", False):        # BASELINE configs[2]: crossover adaptation on (the reference's default), burn-in = a tenth of the run (core.py:299-300)
        kw.update(adapt_crossover=1, crossover_burnin=int(args.burnin_generations), adapt_lag=int(getattr(args, "adapt_lag", 0) or 0))
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
    elif args.target == "user":          # a user's wave-level device function compiled into the persistent kernel (DeviceFunctionLogLike; needs hipcc at run time)
        from pydream_amd.likelihoods import DeviceFunctionLogLike
        c = np.linspace(-2.0, 2.0, d); w = 0.5 + np.arange(d) % 7 / 7.0
        DeviceFunctionLogLike(USER_SRC, "weighted_sq", d, data=np.concatenate([c, w]), always_finite=True)._dz_apply(e)
    else:
        mu = np.array([np.full(d, m) for m in (-5.0, 0.0, 5.0)])
        logF = np.log(np.array([1 / 6., 1 / 3., 1 / 2.])) - (d / 2.) * np.log(2 * np.pi)
        e.set_likelihood_mixture(mu, logF)
    return e


TARGET_NAMES = {"mvn": ("MVN", "correlated MVN"), "mix3": ("3-Gaussian mixture", "3-Gaussian mixture"),
                "user": ("user device function (separable quartic)", "user-supplied device function -1/2 sum_j [w_j t_j^2 + 0.001 t_j^4], t = x - c, compiled into the persistent kernel at run time")}
USER_SRC = r'''
__device__ double weighted_sq(const double* x, int d, const void* data, int lane)
{
    const double* c = (const double*)data; const double* w = c + d;
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) { const double t = x[j] - c[j]; acc = acc + w[j] * (t * t) + 0.001 * ((t * t) * (t * t)); }
    return -0.5 * dz_wave_sum(acc);
}'''


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


def generation_bytes(args):
    """algorithmic HBM bytes per chain-generation,
    B = 8d (2k-1)(2(1-s)+3s) [Z gathers] + 16d [state read + write] + 8d + 8 [trace] + 8d/thin [append] + 16(2k-1) [logp scalars];
    k = 5, s = 0.1, thin = 10: 176 d + 152."""
    d, k, s = float(args.dim), args.multitry, args.snooker
    return 8.0 * d * ((2 * k - 1) * (2.0 * (1.0 - s) + 3.0 * s) + 2.0 + 1.0 + 1.0 / args.thin) + 8.0 + 16.0 * (2 * k - 1)


def measured_traffic(args, n_local, kernel_class, gens_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same workload
    (profiles/<tag>_traffic.json, made by tools/collect_profiles.sh + tools/traffic_from_pmc.py), or None."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_traffic.json")
    default = (n_local == 4096 and args.dim == 100 and args.multitry == 5 and args.target == "mvn" and args.mvn_kind == "tri"
               and args.snooker == 0.1 and args.thin == 10)
    if not (default and kernel_class == "generations" and os.path.exists(path)):
        return None, None
    t = json.load(open(path))
    if "k_generations" not in t["kernel"]:
        return None, None
    return t["bytes_per_generation"] * gens_per_launch, "profiles/%s_traffic.json" % PROFILE_TAG


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


def measured_pmc(args, n_local):
    """VALU / matrix-pipe busy fractions of the dominant kernel from the committed PMC summary, or None."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_summary.json")
    default = (n_local == 4096 and args.dim == 100 and args.multitry == 5 and args.target == "mvn" and args.mvn_kind == "tri")
    if not (default and os.path.exists(path)):
        return None
    summ = json.load(open(path))
    cand = {k: v for k, v in summ.items() if "k_generations" in k and "SQ_BUSY_CYCLES" in v}
    if not cand:
        return None
    v = max(cand.values(), key=lambda x: x.get("SQ_WAVE_CYCLES", 0))
    out = {"source": "profiles/%s_pmc_summary.json" % PROFILE_TAG}
    # units (checked against the instruction counts): SQ_WAVE_CYCLES and SQ_ACTIVE_INST_VALU count quad-cycles summed over waves,
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs
    # generations per dispatch of the profiled run (its average: a launch holds thin generations, or 2 thin at history_lag 1, and the
    # convergence chunks end ragged): from the traffic file made of the same run (tools/traffic_from_pmc.py), else one thin-cycle
    gpd = float(args.thin)
    tpath = os.path.join(ROOT, "profiles", PROFILE_TAG + "_traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("dispatches") and t.get("generations_in_run"):
            gpd = t["generations_in_run"] / float(t["dispatches"])
    if v.get("SQ_WAVES") and v.get("SQ_INSTS_VALU") is not None:
        out["valu_insts_per_wave_generation"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"] / gpd
        if v.get("SQ_INSTS_MFMA") is not None:
            out["mfma_insts_per_wave_generation"] = v["SQ_INSTS_MFMA"] / v["SQ_WAVES"] / gpd
        out["generations_per_profiled_dispatch"] = gpd
    if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAVES"):
        nsimd = 1024.0
        waves_per_simd = v["SQ_WAVES"] / nsimd
        if "SQ_ACTIVE_INST_VALU" in v:
            out["valu_busy"] = v["SQ_ACTIVE_INST_VALU"] * waves_per_simd / v["SQ_WAVE_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            out["mfma_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (nsimd * 4.0 * v["SQ_WAVE_CYCLES"] / v["SQ_WAVES"])
    return out


def timed_blocks(e, K, min_ms, barrier, dist, max_blocks=400):
    """Blocks of exactly K generations, each bracketed by barrier + sync on both sides; returns the block times (s, max over ranks)."""
    times = []
    while True:
        e.trace_reset()
        barrier()
        t0 = time.perf_counter()
        e.step(K)
        e.sync()
        dt = time.perf_counter() - t0              # this rank's K generations (its all-gathers waited for every other rank's)
        if dist is not None:
            dist.barrier()                         # the far side's barrier; the block's time is the slowest rank's
            dt = dist.all_reduce_max([dt])[0]
        times.append(dt)
        if 1e3 * sum(times) >= min_ms or len(times) >= max_blocks:      # (the times are rank-maxima: every rank stops at the same block)
            return times


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="the block of K timed generations is repeated until this much has been timed")
    ap.add_argument("--rhat-chunk", type=int, default=500, help="generations between R-hat evaluations of the convergence run")
    ap.add_argument("--rhat-max-generations", type=int, default=None,
                    help="the convergence run stops here if R-hat has not passed 1.2 (default: max(10000, 16 * dim) -- the 1000-D shard needs ~13000)")
    ap.add_argument("--rhat-min-generations", type=int, default=2000,
                    help="the convergence run lasts at least this long (it doubles as the clock spin-up: a cold MI355X needs ~0.1 s of load)")
    ap.add_argument("--rhat-window", type=int, default=4000, help="generations of the fixed R-hat window after the convergence run")
    ap.add_argument("--spinup", type=int, default=None, help="(deprecated) alias of --rhat-min-generations")
    ap.add_argument("--chains-per-gpu", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--multitry", type=int, default=5)
    ap.add_argument("--thin", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20260929)
    ap.add_argument("--target", choices=["mvn", "mix3", "user"], default="mvn")
    ap.add_argument("--mvn-kind", choices=["dense", "tri"], default="tri",
                    help="form of the MVN whitening matrix: tri = Cholesky factor of the precision (what "
                         "pydream_amd.likelihoods.MVNormalLogLike builds), dense = full precision matrix")
    ap.add_argument("--snooker", type=float, default=0.1, help="snooker probability (reference default 0.1)")
    ap.add_argument("--cpu-chains", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=0, help="generations of the CPU baseline (0: as many as fit --cpu-seconds)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-matrix (reference formula) pass")
    ap.add_argument("--no-events", action="store_true", help="do not time individual kernels with HIP events")
    ap.add_argument("--event-generations", type=int, default=400,
                    help="generations of the event-timed pass (at least --steps): 400 generations are 20 launches of the persistent kernel (20 generations each: two history appends per launch)")
    ap.add_argument("--history-lag", type=int, default=None,
                    help="dz_config.history_lag: appended rows become sampleable this many appends late.  Default 3 at every N (on several "
                         "GPUs the row exchange then hides behind a thin-cycle; one GPU runs the same schedule so that the scaling curve is "
                         "one algorithm, and times lag 0 -- the lockstep schedule -- beside it: value_history_lag0)")
    ap.add_argument("--appends-per-launch", type=int, default=None,
                    help="history appends one launch of the persistent kernels may hold (DZ_MEGA_SEGS).  Default (history_lag + 1) // 2: what the peer "
                         "transport allows on several GPUs, applied at every N")
    ap.add_argument("--no-lag0", action="store_true", help="one GPU: skip the extra lag-0 pass")
    ap.add_argument("--adapt", action="store_true",
                    help="crossover adaptation on (BASELINE configs[2]; the reference's default): the first --burnin-generations generations "
                         "publish positions and adapt the crossover probabilities; timed on their own as `burnin_value`")
    ap.add_argument("--burnin-generations", type=int, default=800, help="crossover_burnin with --adapt (the reference: niterations / 10)")
    ap.add_argument("--adapt-lag", type=int, default=None,
                    help="dz_config.adapt_lag with --adapt: generation g of the burn-in decides with the crossover probabilities as they were after "
                         "the updates of generations <= g - 1 - L, so a launch holds L + 1 burn-in generations.  Default appends_per_launch x thin - 1 (the "
                         "launches inside the burn-in hold as many generations as those behind it); the lockstep adaptation (L = 0, one generation per launch) is timed beside it: burnin_value_adapt_lag0")
    ap.add_argument("--control", choices=["socket", "torch"], default=os.environ.get("DZ_BENCH_CONTROL", "socket"),
                    help="rendezvous of a multi-GPU run: plain TCP (default) or torch.distributed/gloo")
    ap.add_argument("--transport", choices=["peer", "rccl", "host"], default=None,
                    help="row exchange of a multi-GPU run (default: DZ_BENCH_TRANSPORT or peer; falls back peer -> rccl -> host, loudly)")
    ap.add_argument("--no-configs", action="store_true",
                    help="one GPU, default workload: skip the `configs` block (BASELINE configs[1], configs[2] as written, the configs[4] shard, each measured "
                         "the same way as the headline in the same run)")
    ap.add_argument("--no-strong", action="store_true", help="several GPUs: skip the strong-scaling leg (BASELINE configs[3] as written: 32768 chains over the N GPUs)")
    ap.add_argument("--no-rccl-leg", action="store_true", help="several GPUs: skip the second set of timed blocks over the RCCL all-gather (`rccl_value`)")
    ap.add_argument("--rccl-leg", action="store_true", help="one GPU: run the RCCL leg too (a communicator of one rank)")
    args = ap.parse_args(argv)
    args.rhat_max_given = args.rhat_max_generations is not None
    if args.rhat_max_generations is None:
        args.rhat_max_generations = max(10000, 16 * args.dim)
    if args.spinup is not None:
        args.rhat_min_generations = args.spinup
        args.rhat_max_generations = max(args.spinup, min(args.rhat_max_generations, 4 * args.spinup))
    if args.history_lag is None:
        args.history_lag = 3
    if args.appends_per_launch is None:
        args.appends_per_launch = max(1, (args.history_lag + 1) // 2)
    # (the engine's cap on history appends per launch -- one GPU alone could hold history_lag + 1: the same launches at every N)
    os.environ["DZ_MEGA_SEGS"] = str(args.appends_per_launch)
    if args.adapt_lag is None:      # as many burn-in generations per launch as the launches behind the burn-in hold generations -- at every N whose ranks own whole
        # groups of 256 chains (their groups' sums of a whole launch travel in one exchange); other shards run one burn-in generation per launch at any lag
        args.adapt_lag = max(0, args.appends_per_launch * args.thin - 1) if (args.gpus == 1 or args.chains_per_gpu % 256 == 0) else 0
    return args


def workload_label(args, n_local, world):
    """which BASELINE.json configuration a workload is"""
    if args.target == "mvn" and args.dim == 1000 and n_local == 512:
        return "BASELINE configs[4] per-GPU shard (4096 chains x 1000-D correlated MVN over 8 GPUs)"
    if args.target == "mvn" and args.dim == 100 and n_local == 1024 and world == 1:
        return "BASELINE configs[1]"
    if args.target == "mix3" and args.dim == 100 and n_local == 4096 and world == 1 and getattr(args, "adapt", False):
        return "BASELINE configs[2] as written"
    if args.target == "mvn" and args.dim == 100 and n_local == 4096:
        return "BASELINE north_star target / configs[3] per-GPU shard"
    if args.target == "mvn" and args.dim == 100 and n_local * world == 32768:
        return "BASELINE configs[3] as written (32768 chains) on %d GPU%s: the strong-scaling reading" % (world, "" if world == 1 else "s")
    if args.target == "user":
        return "not a BASELINE configuration: north_star's `batched device callback` for a density that is not built in"
    if args.target == "mvn" and args.dim == 200:
        return "the reference example's own dimension (dream_ex_ndim_gaussian.py:29), not a BASELINE configuration"
    return "variant of the BASELINE workloads"


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))        # one process asked for N GPUs: it starts its own N ranks (or fails)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from pydream_amd import _capi            # loads libdreamzs.so (system ROCm runtime) BEFORE torch
    _capi.load_library()
    dist = None
    if world > 1:
        if args.control == "torch":
            dist = TorchGroup(rank, world)
        else:
            from pydream_amd.distributed import socket_group_from_env
            dist = socket_group_from_env()
    out, replicas = measure(args, dist, world, rank)
    if rank != 0:
        if dist is not None:
            dist.close()
        return
    default_workload = (args.target == "mvn" and args.dim == 100 and args.chains_per_gpu == 4096 and args.multitry == 5 and not args.adapt)
    if world == 1 and default_workload and not args.no_configs:
        out["configs"] = baseline_configs(args)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.close()
    if replicas is not None and not replicas["identical"]:
        sys.stderr.write("bench.py: the ranks' archive replicas DIFFER (see replica_check in the line): the number above is not valid\n")
        sys.exit(3)


def baseline_configs(args):
    """The other single-GPU BASELINE.json configurations (and the reference example's own d = 200), each measured in this same run the way the headline is (convergence run with
    the reference's R-hat rule, warm-up, blocks of exactly K generations, an event-timed pass for the roofline) -- without the dense /
    lag-0 / CPU legs.  -> {"configs[1]": {...}, "configs[2]": {...}, "configs[4] shard": {...}}"""
    import copy
    res = {}
    for key, over in (("configs[1]", dict(chains_per_gpu=1024)),
                      ("configs[2]", dict(target="mix3", adapt=True, burnin_generations=800)),
                      # configs[3] as written is 32768 chains over 8 GPUs: its N = 1 point (the strong-scaling anchor; the N > 1 lines carry
                      # `strong_scaling`: the same 32768 chains over their N GPUs)
                      ("configs[3] @ 1 GPU", dict(chains_per_gpu=32768, rhat_cap=2000)),
                      ("configs[4] shard", dict(chains_per_gpu=512, dim=1000)),
                      # not a BASELINE configuration: the reference example's own dimension (dream_ex_ndim_gaussian.py:29), 4096 chains
                      ("example d=200", dict(dim=200, rhat_cap=4000)),
                      # not a BASELINE configuration either: a user's device likelihood inside the persistent kernel (north_star: "the user likelihood is evaluated as a batched device callback")
                      ("user device likelihood", dict(target="user", rhat_cap=2000))):
        a = copy.copy(args)
        cap = over.pop("rhat_cap", None)
        for k_, v_ in over.items():
            setattr(a, k_, v_)
        if not args.rhat_max_given:
            a.rhat_max_generations = max(10000, 16 * a.dim) if cap is None else cap      # (d = 200: the device trace of the convergence run is 6.8 MB per generation)
        a.burnin_generations = min(a.burnin_generations, max(60, a.rhat_max_generations // 4))
        t0 = time.perf_counter()
        try:
            o, _ = measure(a, None, 1, 0, sub=True)
        except Exception as ex:           # a sub-configuration that fails must not take the headline with it: it is reported as failed
            res[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            continue
        rf = o.get("roofline", {})
        c = {"workload": o["config"]["workload"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
             "kernel_variant": o.get("kernel_variant"), "acceptance_rate": o["acceptance_rate"], "timed_blocks": o["timing"]["timed_blocks"],
             "rhat_run_so_far": o["convergence"]["history"][-1][1] if o["convergence"]["history"] else None,
             "rhat_generations": o["convergence"]["generations_run"],
             "generations_to_rhat_below_1p2": o["convergence"]["generations_to_rhat_below_1p2"],
             "roofline": {k_: rf.get(k_) for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_us", "whole_generation_frac", "other_kernels_hbm") if k_ in rf},
             "generation_hbm_frac_of_8TBps": o.get("generation_hbm", {}).get("frac_of_8TBps"),
             "seconds": None}
        if "burnin_value" in o:
            c["burnin_value"] = o["burnin_value"]
            c["burnin"] = {k_: o["burnin"][k_] for k_ in ("burnin_generations", "adapt_lag", "ms_per_step", "kernel_variant", "cr_probs_after_burnin")}
            if "burnin_value_adapt_lag0" in o:
                c["burnin_value_adapt_lag0"] = o["burnin_value_adapt_lag0"]
                c["burnin"]["adapt_lag0"] = o["burnin"]["adapt_lag0"]
        c["seconds"] = time.perf_counter() - t0
        res[key] = c
    return res


def measure(args, dist, world, rank, sub=False):
    """The whole measurement sequence of one workload (module docstring, parts 1-5) on engines of this process; -> (the line as a dict on
    rank 0 / None elsewhere, the replica check or None).  sub: one of the `configs` block's workloads -- no dense / lag-0 legs, no fixed
    R-hat window (the run-so-far R-hat of the convergence run is what is reported)."""
    from pydream_amd import _capi
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    n_local = args.chains_per_gpu
    n_global = n_local * world
    K = args.steps
    chunk = max(1, args.rhat_chunk)
    conv_cap = max(chunk, args.rhat_max_generations)
    est_block_s = max(K * 15e-6 * max(1.0, (args.dim / 100.0) ** 2) * max(1.0, n_local / 4096.0), 1e-5)
    max_blocks = int(min(400, max(2, args.min_timed_ms * 1e-3 / est_block_s + 2)))
    total = conv_cap + args.rhat_window + args.warmup + K * (max_blocks + 2) + max(K, args.event_generations) + 2 * args.thin + (args.burnin_generations + 40 if args.adapt else 0)
    # The transport north_star names -- an RCCL all-gather of the appended rows -- gets its own number in every N > 1 line: after the blocks
    # over the default transport (peer: copy-engine pushes) the SAME engines are re-attached to RCCL and the same blocks are timed again
    # (`rccl_value`).  --rccl-leg runs it with one rank as well (a world-1 communicator: what a one-GPU box can exercise).
    want_rccl = not sub and ((world > 1 and not args.no_rccl_leg) or getattr(args, "rccl_leg", False))
    if want_rccl:
        total += K * (max_blocks + 2) + max(K, 100) + 4 * args.thin
        if local_rank == 0:       # librccl.so is 570 MB: one sequential read puts it in the page cache while the first legs run (dlopen's page faults on a cold file took minutes)
            import threading

            def _warm(path=os.path.join(os.path.dirname(os.path.realpath(_capi.hip_library())), "librccl.so.1")):
                try:
                    with open(path, "rb", buffering=0) as f:
                        while f.read(8 << 20):
                            pass
                except OSError:
                    pass
            threading.Thread(target=_warm, daemon=True).start()
    # (DZ_BENCH_TRANSPORT=host and DZ_BENCH_DEVICE exist so that the multi-rank control flow can be rehearsed on a
    #  one-GPU box: ranks share the device and exchange through the host; measurements use RCCL, one rank per GPU)
    device = int(os.environ.get("DZ_BENCH_DEVICE", local_rank))

    def attach(e):
        """-> (transport in use, note about refused ones).  Every rank must end up on the same transport: a refusal anywhere (e.g.
        hipIpcOpenMemHandle or ncclCommInitRank failing) moves all ranks on to the next one -- a number over a slower transport,
        NAMED as such at the top level of the JSON line, beats no number."""
        if world == 1:
            return None, None
        from pydream_amd.distributed import attach_transport
        first = args.transport or os.environ.get("DZ_BENCH_TRANSPORT", "peer")
        order = [first] + [t for t in ("peer", "rccl", "host") if t != first and ("peer", "rccl", "host").index(t) > ("peer", "rccl", "host").index(first)]
        notes = []
        for transport in order:
            try:        # (attach_transport makes the same collectives on every rank and raises on EVERY rank if any rank failed)
                attach_transport(e, rank, world, transport=transport, group=dist.group)
                return transport, "; ".join(notes) or None
            except Exception as ex:
                notes.append("%s unavailable: %s" % (transport, ex))
        raise SystemExit("no transport could be attached: " + "; ".join(notes))

    def rhat_now(e, nsamples):
        if world == 1:
            return e.get_rhat()
        from pydream_amd.distributed import gelman_rubin_sharded
        return gelman_rubin_sharded(e, nsamples, group=dist.group)

    def converge_and_time(e, with_rhat):
        """parts 1-3 of the sequence on engine e; returns (block times, convergence record, acceptance of the last block)"""
        def barrier():
            e.sync()
            if dist is not None:
                dist.barrier()
                e.comm_barrier()                   # the ranks leave a device collective closer together than a host barrier
        conv = {"chunk": chunk, "criterion": "max R-hat < 1.2 (reference rule: second half of the run so far, all %d chains)" % n_global,
                "generations_to_rhat_below_1p2": None, "history": []}
        done = 0
        t_load = time.perf_counter()
        e.trace_reset()
        while done < conv_cap:
            n = min(chunk, conv_cap - done)
            if not with_rhat:
                e.trace_reset()                       # (no diagnostic on this engine: its trace buffer holds one chunk)
            e.step(n)
            done += n
            if with_rhat:
                r = float(np.max(rhat_now(e, done)))
                conv["history"].append([done, r])
                if r < 1.2 and conv["generations_to_rhat_below_1p2"] is None:
                    conv["generations_to_rhat_below_1p2"] = done
                    conv["rhat_at_that_point"] = r
            else:
                e.sync()
            # (an engine without the diagnostic -- the dense leg -- has no R-hat window after this loop, and with few chains its 2000
            #  generations are over before an idle GPU has raised its clocks (1024 chains: 40 ms; seen as 67 M/s instead of 272): it
            #  keeps the GPU loaded for at least 0.3 s, bounded by the convergence cap)
            # (small populations finish the minimum before an idle GPU has raised its clocks -- 1024 chains x 2000 generations are
            #  35 ms, seen as 64 M/s instead of 297 --: the run also lasts at least 0.3 s, bounded by the convergence cap)
            if done >= args.rhat_min_generations and time.perf_counter() - t_load >= 0.3 and \
                    (conv["generations_to_rhat_below_1p2"] is not None if with_rhat else True):
                break
        conv["generations_run"] = done
        if with_rhat and not sub:
            e.trace_reset()
            e.step(args.rhat_window)
            conv["rhat_window_generations"] = args.rhat_window
            conv["rhat_window_max"] = float(np.max(rhat_now(e, args.rhat_window)))
        barrier()
        e.trace_reset()
        e.step(args.warmup)
        # the timed blocks start right after a history append (the W warm-up generations are followed by the 0..thin-1 more that
        # complete the current thin-cycle), so that a block of K generations is K / thin whole launches of the persistent kernel
        g_now = e.generation() if callable(e.generation) else e.generation
        align = (-(g_now - 1)) % args.thin
        if align:
            e.trace_reset()
            e.step(align)
        barrier()
        times = timed_blocks(e, K, args.min_timed_ms, barrier, dist, max_blocks)
        acc = float(e.get_trace(0, K, with_X=False)["moved"].mean())
        return times, conv, acc

    trace_cap = max(conv_cap, args.rhat_window, K, args.warmup, 2)
    e = setup_engine(_capi.Engine, args, n_global, n_local, rank * n_local, total, device=device, trace_capacity=trace_cap)
    transport, transport_note = attach(e)
    burn = None
    if args.adapt:
        # The crossover burn-in, timed on its own: blocks of K generations inside it (after 20 generations: the adaptation window opens
        # at generation 11, Dream.py:371).  The GPU is taken out of its idle clock state first by a throw-away engine (an idle MI355X
        # needs ~0.1 s of load), because the burn-in is by definition the START of the measured engine's run.
        import copy
        aw = copy.copy(args); aw.adapt = False; aw.thin = 10 ** 6          # (appends once: the archive stays at its seed size however long it runs)
        ew = setup_engine(_capi.Engine, aw, n_local, n_local, 0, 20000, device=device, trace_capacity=0)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.4:
            ew.trace_reset(); ew.step(500); ew.sync()
        ew.close()

        def time_burnin(eb, finish):
            eb.trace_reset(); eb.step(20 + (-(20 - 1)) % args.thin); eb.sync()      # (the blocks start right behind a history append, like the timed blocks after the burn-in: whole launches)
            bt = []
            Kb = max(1, min(K, (args.burnin_generations - 30) // 4))
            while eb.generation + Kb < args.burnin_generations - 1 and (1e3 * sum(bt) < args.min_timed_ms or len(bt) < 3) and len(bt) < 400:
                eb.trace_reset()
                eb.sync()
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                eb.step(Kb)
                eb.sync()
                dt = time.perf_counter() - t0
                if dist is not None:
                    dt = dist.all_reduce_max([dt])[0]
                bt.append(dt)
            bvar = eb.last_kernel_variant()
            rest = args.burnin_generations + 1 - eb.generation
            if finish and rest > 0:
                done_b = 0
                while done_b < rest:
                    n = min(rest - done_b, trace_cap); eb.trace_reset(); eb.step(n); done_b += n
            mb = float(np.median(bt)) if bt else None
            return {"burnin_generations": args.burnin_generations, "adapt_lag": int(eb.cfg.adapt_lag), "block_generations": Kb, "timed_blocks": len(bt),
                    "ms_per_step": (1e3 * mb / Kb) if mb else None, "value": (n_global * args.multitry * Kb / mb) if mb else None,
                    "kernel_variant": bvar, "cr_probs_after_burnin": [float(x) for x in eb.get_cr_state()[0]] if finish else None}
        burn = time_burnin(e, True)
        if world == 1 and args.adapt_lag > 0:      # beside it: the lockstep adaptation (adapt_lag 0: one burn-in generation per launch), the same blocks
            a0 = copy.copy(args); a0.adapt_lag = 0
            e0 = setup_engine(_capi.Engine, a0, n_global, n_local, 0, args.burnin_generations + 40, device=device, trace_capacity=max(K, 20 + args.thin))
            b0 = time_burnin(e0, False)
            e0.close()
            burn["adapt_lag0"] = {k_: b0[k_] for k_ in ("value", "ms_per_step", "kernel_variant", "timed_blocks")}
    t_enq0 = time.perf_counter()
    times, conv, acc = converge_and_time(e, True)
    med = float(np.median(times))

    # ---- a pass of K generations with HIP events around every kernel launch (per-kernel durations
    #      for the roofline; kept out of the timed blocks because each event record costs a few us) ----
    prof = {}
    kernel_variant = e.last_kernel_variant()
    xstats = None
    xbytes = None
    if world > 1:
        e.sync()
        xstats = e.exchange_stats()
        xbytes = e.exchange_bytes()
    if not args.no_events:
        KE = max(K, args.event_generations)
        KE -= KE % args.thin if KE >= args.thin else 0          # whole thin-cycles: every launch of the persistent kernel is a full one
        done_ev = 0
        e.profile_enable(True, prealloc_pairs=8 * min(KE, trace_cap))
        e.profile_reset()
        while done_ev < KE:
            n = min(KE - done_ev, trace_cap)
            e.trace_reset()
            e.step(n)
            done_ev += n
        e.sync()
        e.profile_enable(False)
        # propose / logp / accept / generations launches carry their own start/stop events (hipExtLaunchKernelGGL: the
        # dispatch's begin/end timestamps, the same figures rocprofv3's kernel trace reports); adapt / exchange are
        # bracketed by event records, which adds about `event_bracket_us` to each of those
        for name in ("generations", "propose", "logp", "accept", "adapt", "exchange"):
            each = 1e3 * e.profile_get_list(name)
            n = len(each)
            prof[name] = {"total_ms": float(each.sum()) * 1e-3, "launches": n, "avg_us": float(each.mean()) if n else None,
                          "median_us": float(np.median(each)) if n else None, "min_us": float(each.min()) if n else None,
                          "max_us": float(each.max()) if n else None}
        prof["event_pass_generations"] = KE
        ms0, n0 = e.profile_get("empty")
        prof["event_bracket_us"] = (1e3 * ms0 / n0) if n0 else None
    replicas = None
    if dist is not None:
        # Every rank holds a replica of the archive (and made the adapted crossover probabilities for itself): reduced to a 64-bit
        # checksum on the device (dz_history_checksum waits for every rank's last rows first) and compared across ranks.  A transport
        # that lost, misplaced or served a stale row would otherwise produce a plausible number from a wrong chain.
        h, rows = e.history_checksum()
        crb = e.get_cr_state()[0].tobytes()
        got = dist.all_gather_object([h, rows, crb])
        replicas = {"identical": all(g[0] == got[0][0] and g[1] == got[0][1] and g[2] == got[0][2] for g in got),
                    "archive_rows": [int(g[1]) for g in got], "archive_checksums": ["%016x" % g[0] for g in got],
                    "what": "dz_history_checksum of every rank's archive replica (all appended rows of all ranks, sum mod 2^64 of a "
                            "position-keyed hash of every element) and the bytes of its crossover probabilities, all-gathered after the run"}
        e.sync()
        dist.barrier()           # every rank is past its last exchange: only now may a rank unmap its buffers
    strong = None
    if world > 1 and not sub and not getattr(args, "no_strong", False) and STRONG_CHAINS % world == 0 and args.target == "mvn" and args.dim == 100:
        # BASELINE configs[3] as written -- 32768 chains, 100-D MVN, "report 1/2/4/8 scaling" -- read as STRONG scaling: the same 32768 chains over
        # this run's N GPUs (the N = 1 point is `configs["configs[3] @ 1 GPU"]` of the one-GPU line); a second set of engines on the transport
        # the weak blocks used, a short warm-up from the seed archive, blocks of K generations
        import copy
        try:
            ns_local = STRONG_CHAINS // world
            a_s = copy.copy(args); a_s.chains_per_gpu = ns_local
            sblocks = 24
            e_s = setup_engine(_capi.Engine, a_s, STRONG_CHAINS, ns_local, rank * ns_local, 400 + K * (sblocks + 3) + 4 * args.thin, device=device, trace_capacity=max(K, 200))
            from pydream_amd.distributed import attach_transport
            attach_transport(e_s, rank, world, transport=transport, group=dist.group)

            def sbarrier():
                e_s.sync()
                dist.barrier()
                e_s.comm_barrier()
            e_s.trace_reset(); e_s.step(200); e_s.trace_reset(); e_s.step(190 + (-(390 - 1)) % args.thin)      # (ends a thin-cycle: the blocks are whole launches)
            sbarrier()
            st = timed_blocks(e_s, K, min(args.min_timed_ms, 25.0), sbarrier, dist, sblocks)
            smed = float(np.median(st))
            hs, rows_s = e_s.history_checksum()
            got = dist.all_gather_object([hs, rows_s])
            strong = {"chains_global": STRONG_CHAINS, "chains_per_gpu": ns_local, "n_gpus": world, "value": STRONG_CHAINS * args.multitry * K / smed, "unit": "proposals/s",
                      "ms_per_step": 1e3 * smed / K, "timed_blocks": len(st), "block_generations": K, "kernel_variant": e_s.last_kernel_variant(), "transport": transport,
                      "replicas_identical": all(g[0] == got[0][0] and g[1] == got[0][1] for g in got),
                      "what": "BASELINE configs[3] as written (32768 chains x 100-D MVN) over this run's GPUs: total work fixed as N grows (the top-level "
                              "`value` is the weak-scaling reading, 4096 chains per GPU); its N = 1 point is configs[\"configs[3] @ 1 GPU\"] of the one-GPU line"}
            e_s.sync()
            dist.barrier()
            e_s.close()
        except Exception as ex:
            strong = {"value": None, "note": "strong-scaling leg not run: %s: %s" % (type(ex).__name__, ex)}
    dense = lag0 = rccl = None
    def assemble():
        """the JSON line (rank 0) from what has been measured so far"""
        value = n_global * args.multitry * K / med
        flops_gen = n_local * (2 * args.multitry - 1) * (1.0 if args.mvn_kind == "tri" else 2.0) * float(args.dim) ** 2 if args.target == "mvn" else None
        out = {
            "metric": "proposals/sec (all chains), %dD %s logpdf, MT-DREAM(ZS) multitry=%d, history_lag=%d"
                      % (args.dim, TARGET_NAMES[args.target][0], args.multitry, args.history_lag),
            "value": value, "unit": "proposals/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * med / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d chains/GPU x %d-D %s target (%s), multitry=%d, DE+snooker(%g), nCR=3, history_thin=%d, "
                                   "seed archive max(10d,2N) rows U(-5,15); %s"
                                   % (n_local, args.dim, TARGET_NAMES[args.target][1],
                                      {"mvn": args.mvn_kind, "mix3": "identity cov", "user": "one wave per point"}[args.target], args.multitry, args.snooker, args.thin,
                                      workload_label(args, n_local, world)),
                       "chains_global": n_global, "chains_per_gpu": n_local, "ndim": args.dim, "multitry": args.multitry,
                       "history_lag": args.history_lag,
                       "parallelism": ("chains sharded x%d, history appends replicated on every GPU (transport: %s, history_lag %d)" % (world, transport, args.history_lag))
                                      if world > 1 else "single GPU"},
            "timing": {"timed_blocks": len(times), "block_generations": K, "block_ms_median": 1e3 * med, "block_ms_min": 1e3 * min(times),
                       "block_ms_max": 1e3 * max(times), "timed_ms_total": 1e3 * sum(times),
                       "value_from": "median block", "value_best_block": n_global * args.multitry * K / min(times),
                       "value_worst_block": n_global * args.multitry * K / max(times)},
            "logp_points_per_s": n_global * (2 * args.multitry - 1) * K / med,
            "acceptance_rate": acc,
            "generations_executed": conv["generations_run"] + args.rhat_window + args.warmup + K * len(times) + (0 if args.no_events else prof.get("event_pass_generations", 0)),
            "rhat_max": conv.get("rhat_window_max") if not sub else (conv["history"][-1][1] if conv["history"] else None),
            "convergence": conv,
        }
        if dense is not None:
            out["dense_value"] = dense["value"]
            out["dense"] = dense
        if lag0 is not None:
            out["value_history_lag0"] = lag0["value"]
            out["history_lag0"] = lag0
        if burn is not None:
            out["burnin_value"] = burn["value"]
            out["burnin"] = burn
            out["config"]["workload"] += "; crossover adaptation ON, crossover_burnin %d, adapt_lag %d (%s): `burnin_value` is the rate inside the burn-in, `value` after it" % (
                args.burnin_generations, burn["adapt_lag"], "the updates reach the chains' decisions that many generations late: up to adapt_lag + 1 burn-in generations per launch" if burn["adapt_lag"] else "lockstep adaptation: one burn-in generation per launch")
            out["config"]["adapt_lag"] = burn["adapt_lag"]
            if "adapt_lag0" in burn:
                out["burnin_value_adapt_lag0"] = burn["adapt_lag0"]["value"]
        out["kernel_variant"] = kernel_variant
        out["history_lag"] = args.history_lag
        if replicas is not None:
            out["replicas_identical"] = replicas["identical"]
            out["replica_check"] = replicas
        if strong is not None:
            out["strong_scaling"] = strong
        if rccl is not None:
            out["rccl_value"] = rccl.get("value")
            out["rccl_ranks"] = rccl.get("ranks")
            out["rccl_exchange_exposed_us_per_cycle"] = rccl.get("exchange_exposed_us_per_cycle")
            out["rccl"] = rccl
        if world > 1:
            # top level, not buried in config: what carried the rows, and how much of the exchange the generations had to wait for
            if xbytes is not None:
                out["exchange_bytes_to_each_peer"] = {"history_rows": xbytes[0], "positions": xbytes[1], "adaptation_group_sums": xbytes[2],
                                                      "per_burnin_generation": (xbytes[2] / (args.burnin_generations + 1)) if (args.adapt and xbytes[2]) else None,
                                                      "what": "bytes rank 0 handed to the transport for each other rank over the whole run (dz_exchange_bytes); with crossover "
                                                              "adaptation a rank that owns whole groups of 256 chains sends its groups' column sums every burn-in generation, "
                                                              "its positions only once"}
            out["transport"] = transport if transport != "host" else "host-fallback"
            if transport_note:
                out["transport_note"] = transport_note
            if xstats is not None and transport == "peer":
                nx, ngate, wait_us = xstats
                out["exchange"] = {"exchanges": nx, "gates": ngate, "gate_wait_us_total": wait_us,
                                   "exchange_exposed_us_per_cycle": (wait_us / ngate) if ngate else None,
                                   "what": "time the one-wave gate kernels in front of the launches spent waiting for the other ranks' rows "
                                           "(rank 0, whole run incl. convergence and warm-up); bytes per rank and cycle: %d" % (n_local * 8 * ((args.dim + 15) // 16 * 16))}
                out["exchange_exposed_us_per_cycle"] = out["exchange"]["exchange_exposed_us_per_cycle"]
        if prof:
            KE = prof["event_pass_generations"]
            ab = algorithmic_bytes(args, n_local)
            if prof.get("generations", {}).get("launches"):
                # the persistent kernel covers whole generations: SURVEY.md section 8(d) B = 176 d + 152 bytes per chain-generation
                ab["generations"] = n_local * generation_bytes(args) * KE / prof["generations"]["launches"]
            cand = {k: v for k, v in prof.items() if isinstance(v, dict) and k in ab and v["launches"]}
            dom = max(cand, key=lambda k: cand[k]["total_ms"])
            gens_per_launch = KE / cand[dom]["launches"] if dom == "generations" else 1.0
            # Duration of one launch of the dominant kernel: the MEDIAN over the event-timed launches (>= 20 of them), never more than the
            # wall clock allows -- a kernel cannot take longer than the timed block around it (the event pass runs a few percent slower
            # than the un-instrumented blocks: round 2's line had 314 us per launch inside blocks of 303.6 us per 10 generations).
            launch_s = cand[dom]["median_us"] * 1e-6
            launches_per_block = (K / gens_per_launch) if dom == "generations" else K * {"propose": 2.0, "logp": 2.0, "accept": 1.0}.get(dom, 1.0)
            wall_cap_s = med / max(launches_per_block, 1e-9)
            events_exceed_wall = launch_s > wall_cap_s
            if events_exceed_wall and dom == "generations":
                launch_s = wall_cap_s
            tmeta = {"launch_us": launch_s * 1e6, "launch_us_from": ("timed blocks (median block / launches per block): the event-timed median %.1f us exceeds it"
                                                                      % cand[dom]["median_us"]) if (events_exceed_wall and dom == "generations")
                     else "median of %d event-timed launches" % cand[dom]["launches"],
                     "launch_us_event_median": cand[dom]["median_us"], "launch_us_event_min": cand[dom]["min_us"],
                     "launches_timed": cand[dom]["launches"], "generations_per_launch": gens_per_launch}
            compute_bound = args.dim > 128 and args.target == "mvn" and dom in ("logp", "generations")
            if compute_bound:
                # d > 128: the batched quadratic form dominates and is FP64-matrix-bound (~100 flop/B at d = 1000)
                fl = flops_gen * (gens_per_launch if dom == "generations" else 0.5)          # logp: two launches per generation
                achieved = fl / launch_s / 1e12
                whole = flops_gen * K / med / 1e12                                            # every kernel of the generation in the denominator
                out["roofline"] = {"bound": "fp64_mfma", "kernel": "k_" + dom, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                                   "algorithmic_flops_per_launch": fl,
                                   "whole_generation_achieved": whole, "whole_generation_frac": whole / FP64_MFMA_PEAK_TFLOPS}
                # the other third of a streamed generation, priced as what it is -- HBM traffic: algorithmic bytes / the AVERAGE event-timed
                # launch (no ramp subtracted).  Per try: base row + the archive rows read, the proposal written = 8d (rows_z + 2); the
                # Metropolis step's own bytes ride in the launch that carries the next proposal set (k_accept_propose).
                rows_z = 2 * (1 - args.snooker) + 3 * args.snooker
                try_b = 8.0 * args.dim * (rows_z + 2.0)
                others = {}
                for cls, name, nbytes in (("accept", "k_accept_propose / k_accept (Metropolis step + the next generation's proposal set)", ab["accept"] + n_local * args.multitry * try_b),
                                          ("propose", "k_propose_stream (reference set)", n_local * (args.multitry - 1) * try_b)):
                    v = prof.get(cls)
                    if v and v["launches"]:
                        gbps = nbytes / (v["avg_us"] * 1e-6) / 1e9
                        others[cls] = {"kernel": name, "algorithmic_bytes_per_launch": nbytes, "avg_us": v["avg_us"], "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBS}
                out["roofline"]["other_kernels_hbm"] = others
            else:
                achieved = ab[dom] / launch_s / 1e9
                traffic, tsrc = measured_traffic(args, n_local, dom, gens_per_launch)
                out["roofline"] = {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                                   "algorithmic_bytes_per_launch": ab[dom]}
                pm = measured_pmc(args, n_local)
                if pm:
                    out["roofline"].update({k: v for k, v in pm.items() if k != "source"})
                    out["roofline"]["pmc_source"] = pm["source"]
                    if dom == "generations" and "mfma_insts_per_wave_generation" in pm:
                        # What bounds this kernel is instruction issue, not HBM: the vector ALU and the FP64 matrix pipe of a SIMD do not
                        # overlap on gfx950 (profiles/r02_mfma_valu_overlap.txt), so a generation cannot take fewer cycles than its four
                        # waves per SIMD need to issue their VALU instructions (4 cycles each) and their FP64 MFMAs (64 cycles each).
                        wps = 4.0
                        floor_cycles = wps * (pm["valu_insts_per_wave_generation"] * 4.0 + pm["mfma_insts_per_wave_generation"] * 64.0)
                        gen_cycles = launch_s / gens_per_launch * ENGINE_CLOCK_HZ
                        out["roofline"]["issue_floor_frac"] = floor_cycles / gen_cycles
                        out["roofline"]["issue_floor"] = {
                            "floor_cycles_per_generation": floor_cycles, "measured_cycles_per_generation": gen_cycles, "clock_hz": ENGINE_CLOCK_HZ,
                            "formula": "4 waves/SIMD x (VALU instructions x 4 cycles + FP64 MFMA instructions x 64 cycles) per wave-generation "
                                       "(instruction counts: committed PMC file) / (measured launch duration / generations per launch x clock)",
                            "reading": "the HBM `frac` above is what BASELINE asks for; this is the fraction of the kernel's time its SIMDs need "
                                       "just to issue its instructions -- the rest is dependent-latency stalls at four waves per SIMD"}
                if flops_gen:
                    out["roofline"]["fp64_matrix_tflops"] = flops_gen * K / med / 1e12
            out["roofline"].update(tmeta)
            out["roofline"]["kernel_variant"] = kernel_variant
            out["kernel_times"] = prof
            gen_bytes = n_local * generation_bytes(args)             # B = 176 d + 152 per chain-generation at k = 5, s = 0.1
            out["generation_hbm"] = {"algorithmic_bytes_per_generation": gen_bytes,
                                     "achieved_GBps": gen_bytes * K / med / 1e9,
                                     "frac_of_8TBps": gen_bytes * K / med / 1e9 / HBM_PEAK_GBS}
        return out

    watchdog = None
    if want_rccl and rank == 0 and world > 1:
        # The RCCL leg must never cost the line: if the communicator does not come up (ncclCommInitRank cannot be given a deadline) the
        # line measured so far is printed without it and the process leaves.
        import threading
        T = float(os.environ.get("DZ_BENCH_RCCL_DEADLINE_S", "240"))

        def bail():
            nonlocal rccl
            rccl = {"value": None, "note": "RCCL leg not run: it did not finish within %g s (DZ_BENCH_RCCL_DEADLINE_S); the line is printed without it" % T}
            o = assemble()
            print(json.dumps(o), flush=True)
            os._exit(0 if (replicas is None or replicas["identical"]) else 3)
        watchdog = threading.Timer(T, bail)
        watchdog.daemon = True
        watchdog.start()
    if want_rccl:
        rccl = {"what": "the same engines, after the blocks above, re-attached to RCCL (in-place ncclAllGather of the appended rows on the engine's "
                        "stream, between two launches) and the same blocks of K generations timed again"}
        try:
            if transport == "rccl":
                rccl.update(value=n_global * args.multitry * K / med, ms_per_step=1e3 * med / K, note="RCCL carried the blocks above: rccl_value == value")
            else:
                if transport == "peer":
                    e.peer_detach()
                if world > 1:
                    from pydream_amd.distributed import attach_transport
                    attach_transport(e, rank, world, transport="rccl", group=dist.group)      # (raises on EVERY rank if any rank failed)
                else:
                    e.comm_init_rccl(0, 1, _capi.comm_unique_id())

                def rbarrier():
                    e.sync()
                    if dist is not None:
                        dist.barrier()
                    e.comm_barrier()
                e.trace_reset(); e.step(2 * args.thin)
                rbarrier()
                rt = timed_blocks(e, K, args.min_timed_ms, rbarrier, dist, max_blocks)
                rmed = float(np.median(rt))
                rccl.update(value=n_global * args.multitry * K / rmed, ms_per_step=1e3 * rmed / K, timed_blocks=len(rt), kernel_variant=e.last_kernel_variant())
                KR = max(K, 100); KR -= KR % args.thin if KR >= args.thin else 0
                e.profile_enable(True, prealloc_pairs=8 * min(KR, trace_cap)); e.profile_reset()
                e.trace_reset(); e.step(KR); e.sync()
                e.profile_enable(False)
                ex = 1e3 * e.profile_get_list("exchange")
                rccl["exchange_exposed_us_per_cycle"] = float(ex.mean()) if len(ex) else None
                rccl["exchanges_timed"] = int(len(ex))
            rccl["ranks"] = e.comm_count()
            rccl["library"] = _capi.comm_library()
            if dist is not None:
                h, rows = e.history_checksum()
                got = dist.all_gather_object([h, rows])
                rccl["replicas_identical"] = all(g[0] == got[0][0] and g[1] == got[0][1] for g in got)
                e.sync()
                dist.barrier()
        except Exception as ex:         # (attach_transport fails on all ranks together; a one-GPU rehearsal cannot give RCCL two ranks on one device)
            rccl.update(value=None, note="RCCL leg not run: %s" % ex)
    if watchdog is not None:
        watchdog.cancel()
    e.close()

    if world == 1 and args.target == "mvn" and args.mvn_kind == "tri" and not args.no_dense and not sub:
        import copy
        a2 = copy.copy(args)
        a2.mvn_kind = "dense"
        e2 = setup_engine(_capi.Engine, a2, n_global, n_local, 0, total, device=device, trace_capacity=max(K, args.warmup, chunk, 2))
        t2, _, acc2 = converge_and_time(e2, False)
        e2.close()
        m2 = float(np.median(t2))
        dense = {"value": n_global * args.multitry * K / m2, "ms_per_step": 1e3 * m2 / K, "timed_blocks": len(t2),
                 "formula": "log_F - x.(invC.x)/2 with the dense precision matrix (dream_ex_ndim_gaussian.py:49-52)", "acceptance_rate": acc2}
    if world == 1 and args.history_lag != 0 and not args.no_lag0 and not sub:
        import copy
        a3 = copy.copy(args)
        a3.history_lag = 0
        e3 = setup_engine(_capi.Engine, a3, n_global, n_local, 0, total, device=device, trace_capacity=max(K, args.warmup, chunk, 2))
        t3, _, acc3 = converge_and_time(e3, False)
        v3 = e3.last_kernel_variant()
        e3.close()
        m3 = float(np.median(t3))
        lag0 = {"value": n_global * args.multitry * K / m3, "ms_per_step": 1e3 * m3 / K, "timed_blocks": len(t3), "acceptance_rate": acc3,
                "kernel_variant": v3, "what": "the same timed blocks with history_lag = 0: the lockstep schedule (an append is sampled from the "
                                              "next generation on) that most of the reference-made fixtures pin"}
    if rank != 0:
        return None, replicas
    return assemble(), replicas


def self_launch(args):
    """`python bench.py --gpus N` as ONE process with no launcher's environment: start the N ranks here -- rank i on device i, the
    same command line, torch.distributed.run's environment variables -- and return the exit code (0 only if every rank returned 0).
    Never fewer ranks than asked for: with fewer than N devices on the node this is an error, not a smaller run (DZ_BENCH_DEVICE=<i>
    puts every rank on device i: the control-flow rehearsal on a one-GPU box)."""
    import signal
    import subprocess
    n = args.gpus
    if "DZ_BENCH_DEVICE" not in os.environ:
        from pydream_amd import _capi
        have = _capi.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but this node shows %d HIP device(s); refusing to run fewer ranks than asked for\n" % (n, have))
            return 2
    if args.control == "torch":          # gloo's store needs a real port (the socket control plane does not use MASTER_PORT as a port)
        import socket
        probe = socket.socket(); probe.bind(("127.0.0.1", 0)); port = probe.getsockname()[1]; probe.close()
    else:
        port = 20000 + os.getpid() % 20000
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc = 0
    try:
        live = list(procs)
        while live:
            for pr in list(live):
                code = pr.poll()
                if code is None:
                    continue
                live.remove(pr)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    # (a rank that died leaves the others waiting in a collective: end them -- exactly these processes)
                    deadline = time.time() + (30.0 if code == 3 else 5.0)       # (3 = rank 0's replica check failed: the others are already leaving)
                    while time.time() < deadline and any(q.poll() is None for q in live):
                        time.sleep(0.05)
                    for q in live:
                        if q.poll() is None:
                            q.send_signal(signal.SIGTERM)
            time.sleep(0.02)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


class TorchGroup:
    """--control torch: torch.distributed / gloo behind the few calls bench.py makes (the SocketGroup's interface)"""

    def __init__(self, rank, world):
        import torch.distributed as td
        td.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
        self.td, self.rank, self.world, self.group = td, rank, world, None

    def barrier(self):
        self.td.barrier()

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.td.all_gather_object(out, obj)
        return out

    def all_reduce_max(self, values):
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return [float(x) for x in t]

    def close(self):
        self.td.destroy_process_group()


if __name__ == "__main__":
    main()
