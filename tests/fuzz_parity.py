"""Differential fuzzing of the HIP engine against the CPU oracle: random configurations (sizes off every tile and strip boundary,
multitry on / off, several DE pairs, gamma levels, crossover / gamma adaptation, snooker rates, priors with and without hard boundaries,
redraw rounds, history thinning and lag, dense / triangular MVN, mixture and host likelihoods, parallel tempering, chain-by-chain
stepping, sharded engines), same seeded inputs to both, everything
compared bit for bit -- decision sequences, states, log densities, archive, adaptation state.

Test infrastructure (it drives the oracle): `tests/test_gpu_fuzz.py` runs a fixed set of seeds; as a script it runs as many as asked,
    python tests/fuzz_parity.py --n 400 --seed 1          (on the GPU box)
and prints the configuration of every mismatch.  References: the step is Dream.astep (/root/reference/pydream/Dream.py:193-422) under
the lockstep schedule S2 of DESIGN.md section 5."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

DIMS = [2, 3, 5, 10, 16, 17, 31, 32, 33, 48, 64, 65, 100, 100, 100, 112, 127, 128, 129, 200, 333]
CHAINS = [3, 4, 5, 15, 16, 17, 48, 63, 64, 65, 100, 250, 256, 1000, 1024, 1100, 2048]


def draw_config(rng, long=False, dims=None, chains=None, adapt_lag_arm=False, sharded_arm=False):
    d = int(rng.choice(dims or DIMS))
    N = int(rng.choice(chains or (CHAINS + ([3000, 4096, 4096] if long else []))))
    if d > 128 and not chains:                                # (oracle time; --chains lifts it: the k_generations_d2 regime is 128 < d <= 228)
        N = min(N, 256)
    if rng.random() < 0.04:
        d, N = 1000, int(rng.choice([16, 130]))
    k = int(rng.choice([1, 1, 3, 4, 5, 5, 5, 6, 9, 16, 20]))      # (round 6: 16..32 tries run k_generations_d2 above 1024 chains with the triangular factor; below, and otherwise, the multi-kernel path)
    depairs = int(rng.choice([1, 1, 1, 2, 3]))
    ngamma = int(rng.choice([1, 1, 2, 4]))
    ncr = int(min(d, rng.choice([1, 2, 3, 3, 5])))
    adapt_cr = int(rng.random() < (0.85 if adapt_lag_arm else 0.35))
    adapt_g = int(ngamma > 1 and rng.random() < 0.5)
    if adapt_lag_arm and not (adapt_cr or adapt_g):
        adapt_cr = 1
    n = int(rng.integers(12, 42)) * (3 if long else 1)
    burnin = int(rng.choice([6, 15, n + 5])) if (adapt_cr or adapt_g) else 0
    lk = str(rng.choice(["mvn_dense", "mvn_tri", "mvn_tri", "mix"]))
    if N <= 256 and d <= 128 and rng.random() < 0.1:          # a host likelihood (the reference's usual case: a Python function), batch callback
        lk = "host"
    prior = str(rng.choice(["flat", "flat", "normal", "uniform", "uniform_narrow", "uniform_open"]))
    if prior == "uniform_open" and k == 1:
        prior = "uniform"
    cfg = dict(d=d, N=N, k=k, depairs=depairs, ngamma=ngamma, ncr=ncr, adapt_cr=adapt_cr, adapt_g=adapt_g, burnin=burnin, n=n, lk=lk, prior=prior,
               thin=int(rng.choice([1, 2, 5, 10, 10])), lag=int(rng.choice([0, 0, 0, 1, 2, 3])), snooker=float(rng.choice([0.0, 0.1, 0.1, 0.4])),
               pgu=float(rng.choice([0.0, 0.2, 0.2, 0.6])), lamb=float(rng.choice([0.05, 0.2])), zeta=float(rng.choice([1e-12, 1e-6])),
               zero_mean=int(rng.random() < 0.5), J=int(rng.choice([2, 3])), extra_rows=int(rng.integers(0, 40)), seed=int(rng.integers(1, 2 ** 31 - 1)))
    # parallel tempering (core.py:131-248): the reference's ladder T_i = 0.001^(i/N), one swap attempt per generation (S2 only, no lag)
    cfg["pt"] = int(rng.random() < 0.08 and cfg["lag"] == 0 and prior != "uniform_open" and not adapt_lag_arm)
    # chains sharded over W engines (one per rank in production; here W threads of one process, rows exchanged through the host
    # transport): the result must not depend on W
    cfg["world"] = 1
    if not cfg["pt"] and (sharded_arm or rng.random() < 0.12):
        ws = [w for w in (2, 3, 4) if N % w == 0 and N // w >= 2]
        if ws:
            cfg["world"] = int(rng.choice(ws))
    # schedule S1 (Dream.astep driven chain by chain, every chain with its own copy of the adapted probabilities): few chains, no lag
    cfg["s1"] = int(N <= 17 and rng.random() < 0.5 and not cfg["pt"] and cfg["world"] == 1 and not adapt_lag_arm)
    if cfg["s1"]:
        cfg["lag"] = 0
    # adapt_lag (round 6): the adaptation's updates reach the chains' decisions L generations late (lockstep generations only)
    cfg["adapt_lag"] = 0
    if (adapt_cr or adapt_g) and not cfg["pt"] and not cfg["s1"] and (adapt_lag_arm or rng.random() < 0.6):
        cfg["adapt_lag"] = int(rng.choice([1, 2, 4, 9, 19]))
        cfg["burnin"] = int(rng.choice([6, 15, max(13, n - 8), n + 5]))
    return cfg


def build(Cls, c, device_kw):
    d, N, k, n = c["d"], c["N"], c["k"], c["n"]
    rng = np.random.default_rng(c["seed"])
    M0 = max(10 * d, 2 * c["depairs"] * N) + c["extra_rows"]
    Z0 = rng.uniform(-5.0, 15.0, (M0, d))
    kw = dict(nchains=N, ndim=d, multitry=k, depairs=c["depairs"], ncr=c["ncr"], ngamma=c["ngamma"], history_thin=c["thin"],
              crossover_burnin=c["burnin"], adapt_crossover=c["adapt_cr"], adapt_gamma=c["adapt_g"], hardboundaries=0 if c["prior"] == "uniform_open" else 1,
              history_lag=c["lag"], adapt_lag=c.get("adapt_lag", 0), history_capacity=M0 + N * (n // c["thin"] + 2), trace_capacity=n, seed=c["seed"] & 0x7fffffff,
              lamb=c["lamb"], zeta=c["zeta"], snooker=c["snooker"], p_gamma_unity=c["pgu"])
    if c.get("s1"):
        kw["schedule"] = 1 if Cls.__module__.startswith("oracle") else 2      # (the device engine is driven chain by chain instead: dz_step_range)
        kw["trace_capacity"] = 0
    kw.update(device_kw)
    e = Cls(**kw)
    table = np.array([[2.38 / np.sqrt(2.0 * (dl + 1) * np.arange(1, d + 1)) / 2.0 ** lev for dl in range(c["depairs"])] for lev in range(c["ngamma"])])
    e.set_gamma_table(table)
    if c["prior"] == "normal":
        e.set_prior(np.full(d, 1, np.int32), np.linspace(-1.0, 2.0, d), np.full(d, 30.0))
    elif c["prior"] in ("uniform", "uniform_open"):
        e.set_prior(np.full(d, 2, np.int32), np.full(d, -6.0), np.full(d, 22.0))
        if c["prior"] == "uniform":
            e.set_bounds(np.full(d, -6.0), np.full(d, 16.0))
    elif c["prior"] == "uniform_narrow":                      # boundaries inside the support, and mixed kinds: every third dimension flat
        kind = np.full(d, 2, np.int32); kind[::3] = 0
        e.set_prior(kind, np.full(d, -8.0), np.full(d, 30.0))
        e.set_bounds(np.full(d, -6.0), np.full(d, 16.0))
    if c["lk"].startswith("mvn"):
        i = np.arange(1, d + 1.0)
        P = np.linalg.inv((.5 * np.eye(d) + .5) * np.sqrt(np.outer(i, i)))
        tri = c["lk"] == "mvn_tri"
        Mx = np.linalg.cholesky((P + P.T) / 2).T if tri else P
        e.set_likelihood_mvn(np.zeros(d) if c["zero_mean"] else np.linspace(-1, 1, d), Mx, 1 if tri else 0, 0.0)
    elif c["lk"] == "host":
        e.set_likelihood_host(lambda X: (np.zeros(len(X)), -0.5 * np.sum((X - 2.0) ** 2, axis=1) / 9.0))
    else:
        J = c["J"]
        mu = np.array([np.full(d, m) for m in np.linspace(-4.0, 6.0, J)])
        logF = np.log(np.arange(1, J + 1) / np.arange(1, J + 1).sum()) - (d / 2.) * np.log(2 * np.pi)
        e.set_likelihood_mixture(mu, logF)
    e.set_history(Z0)
    if c.get("pt"):
        e.set_temperatures(np.array([np.power(.001, float(i) / N) for i in range(N)]))
    off, nl = device_kw.get("chain_offset", 0), device_kw.get("nchains_local", N)
    e.set_state(Z0[off:off + nl])
    return e


class ThreadExchange:
    """all-gather of the ranks' byte blocks between W threads of this process (what HostExchange does between processes)"""

    def __init__(self, world):
        import threading
        self.bar, self.buf = threading.Barrier(world), [None] * world

    def callback(self, rank):
        def cb(send, nbytes):
            self.buf[rank] = bytes(send)
            self.bar.wait()
            out = b"".join(self.buf)
            self.bar.wait()
            return out
        return cb


def run_sharded(G, c):
    """-> (trace dict with the ranks' columns side by side, archive, crossover state) of W sharded HIP engines"""
    import threading
    W, N, n = c["world"], c["N"], c["n"]
    nl = N // W
    xch = ThreadExchange(W)
    res, err = [None] * W, []

    def work(r):
        try:
            e = build(G.Engine, c, dict(nchains_local=nl, chain_offset=r * nl))
            e.set_exchange(xch.callback(r))
            half = n // 2
            e.step(half)
            if r == 0:
                tally(e.last_kernel_variant())
            e.step(n - half)
            if r == 0:
                tally(e.last_kernel_variant())
            res[r] = (e.get_trace(0, n), e.get_history(), e.get_cr_state(), e.get_gamma_state(), e.get_state())
            e.close()
        except Exception as ex:
            err.append(ex)
            xch.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join() for t in th]
    if err:
        raise err[0]
    tr = {key: np.concatenate([res[r][0][key] for r in range(W)], axis=1) for key in ("snooker", "cr_idx", "try_idx", "moved", "X", "logp")}
    for r in range(1, W):                                       # archive and adaptation state are replicated
        assert np.array_equal(res[r][1], res[0][1]) and all(np.array_equal(u, v) for u, v in zip(res[r][2], res[0][2]))
    state = tuple(np.concatenate([res[r][4][i] for r in range(W)]) for i in range(len(res[0][4])))
    return tr, res[0][1], res[0][2], res[0][3], state


KINDS = {}


def tally(variant):
    """which kind of launch ran last (the summary line says how much of each the drawn configurations reached)"""
    kind = ("ring" if variant.endswith("+ring") else "multi" if "multi>" in variant else "multi-kernel" if variant.startswith("multi-kernel") else
            variant.split("<")[0] if variant else "none")
    KINDS[kind] = KINDS.get(kind, 0) + 1


def run_one(G, O, c):
    """-> None when the two agree on everything, else a description of the first difference."""
    out = []
    for Cls in (G.Engine, O.Engine):
        if Cls is G.Engine and c.get("world", 1) > 1:
            out.append(run_sharded(G, c))
            continue
        e = build(Cls, c, {})
        if c.get("s1"):                                         # the HIP engine steps chain by chain (dz_step_range), the oracle runs its schedule S1
            if Cls is G.Engine:
                for _ in range(c["n"]):
                    for ch in range(c["N"]):
                        e.step_range(ch, 1)
            else:
                e.step(c["n"])
            out.append(({}, e.get_history(), e.get_cr_state(), e.get_gamma_state(), e.get_state()))
            e.close()
            continue
        half = c["n"] // 2                                      # two step calls: launch segmentation restarts in between
        e.step(half)
        if Cls is G.Engine:
            tally(e.last_kernel_variant())
        e.step(c["n"] - half)
        if Cls is G.Engine:
            tally(e.last_kernel_variant())
        out.append((e.get_trace(0, c["n"]), e.get_history(), e.get_cr_state(), e.get_gamma_state(), e.get_state()) + ((e.get_swaps(0, c["n"]),) if c.get("pt") else ()))
        e.close()
    a, b = out
    if c.get("pt") and not np.array_equal(a[5], b[5]):
        return "temperature swaps differ"
    for key in (() if c.get("s1") else ("snooker", "cr_idx", "try_idx", "moved", "X", "logp")):
        if not np.array_equal(a[0][key], b[0][key]):
            bad = np.argwhere(np.asarray(a[0][key]) != np.asarray(b[0][key]))
            return "trace[%s] differs first at %s" % (key, bad[0].tolist())
    if not np.array_equal(a[1], b[1]):
        return "archive differs"
    for name, x, y in (("cr_state", a[2], b[2]), ("gamma_state", a[3], b[3]), ("state", a[4], b[4])):
        for u, v in zip(x, y):
            if not np.array_equal(u, v):
                return name + " differs"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this long (0: run all --n)")
    ap.add_argument("--long", action="store_true", help="three times the generations, populations up to 4096 chains")
    ap.add_argument("--adapt-lag", action="store_true", help="every configuration with an adaptation and an adapt_lag (lockstep, no tempering): the launches that hold several burn-in generations")
    ap.add_argument("--sharded", action="store_true", help="every configuration that can be sharded is (2..4 engines of one process, rows exchanged through the host transport); with --chains 512,1024,2048 the ranks own whole groups of 256 chains")
    ap.add_argument("--dims", default="", help="comma-separated dimensions to draw from instead of the built-in list")
    ap.add_argument("--chains", default="", help="comma-separated chain counts to draw from (also lifts the 256-chain cap of d > 128)")
    args = ap.parse_args()
    dims = [int(x) for x in args.dims.split(",") if x] or None
    chains = [int(x) for x in args.chains.split(",") if x] or None
    from pydream_amd import _capi as G
    from oracle import oracle as O
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); bad = 0; done = 0
    for i in range(args.n):
        c = draw_config(rng, args.long, dims, chains, args.adapt_lag, args.sharded)
        try:
            r = run_one(G, O, c)
        except Exception as ex:                                 # an engine refusing a configuration must refuse it on both sides: report
            r = "exception: %s" % ex
        done += 1
        if r is not None:
            bad += 1
            print("MISMATCH #%d: %s\n   %s" % (i, r, c), flush=True)
        if args.seconds and time.time() - t0 > args.seconds:
            break
    print("fuzz: %d configurations, %d mismatches, %.0f s (seed %d)" % (done, bad, time.time() - t0, args.seed))
    print("last launch after each half, by kind: " + ", ".join("%s %d" % kv for kv in sorted(KINDS.items())))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
