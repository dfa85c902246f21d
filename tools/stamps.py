#!/usr/bin/env python3
"""Summarise gpurun_out/stamps.bin (DZ_EXP_STAMPS build): per-wave cycle stamps of the last two proposal launches."""
import numpy as np, sys
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64)
nl = a.size // 64
MG = a[3 * nl * 16:].reshape(-1, 16).astype(np.int64)
L = a[2 * nl * 16:3 * nl * 16].reshape(-1, 16).astype(np.int64)
a = a[:2 * nl * 16].reshape(2, nl, 16).astype(np.int64)
for ph in (0, 1):
    s = a[ph]
    ok = s[:, 0] > 0
    s = s[ok]
    if not len(s):
        continue
    t0 = s[:, 0].min()
    n = 5 if ph == 0 else 4
    print("phase %d" % ph)
    idx = [1] + [2 + j for j in range(2 * n)] + [15]
    de = np.all(np.diff(s[:, idx], axis=1) >= 0, axis=1) & (s[:, 15] - s[:, 0] < 200000)     # DE waves of this launch (snooker waves leave stale stamps)
    sd = s[de]
    t0 = sd[:, 0].min()
    print("  span of DE waves: %d cycles; start spread p50 %d p99 %d" % (sd[:, 15].max() - t0, *np.percentile(sd[:, 0] - t0, [50, 99])))
    print("  DE waves %d: preamble mean %d" % (len(sd), (sd[:, 1] - sd[:, 0]).mean()))
    print("    wait for first rows (before tries -> rows of try 0): mean %d" % (sd[:, 2] - sd[:, 1]).mean())
    for i in range(n):
        comp = sd[:, 3 + 2 * i] - sd[:, 2 + 2 * i]
        nxt = (sd[:, 4 + 2 * i] - sd[:, 3 + 2 * i]) if i + 1 < n else (sd[:, 15] - sd[:, 3 + 2 * i])
        print("    try %d: arithmetic %d cycles, then wait %d" % (i, comp.mean(), nxt.mean()))
    print("  whole wave: mean %d p95 %d cycles" % ((sd[:, 15] - sd[:, 0]).mean(), np.percentile(sd[:, 15] - sd[:, 0], 95)))

L = L[(L[:, 0] > 0) & (L[:, 5] > L[:, 0]) & (L[:, 5] - L[:, 0] < 200000)]
if len(L):
    names = ["fetch+staging loads issued, own part stored", "block barrier", "tile -> LDS", "A reads + MFMAs", "epilogue (butterflies, store)"]
    print("logp (last launch): %d waves, span %d cycles" % (len(L), L[:, 5].max() - L[:, 0].min()))
    for i, nm in enumerate(names):
        dtt = L[:, i + 1] - L[:, i]
        print("   %-46s mean %6d  p95 %6d" % (nm, dtt.mean(), np.percentile(dtt, 95)))
    print("   whole wave mean %d" % (L[:, 5] - L[:, 0]).mean())

MG = MG[(MG[:, 0] > 0) & (MG[:, 9] > MG[:, 0]) & (MG[:, 9] - MG[:, 0] < 2000000)]
if len(MG):
    names = ["propose 0 (k tries)", "barrier", "likelihood tiles (MFMA)", "barrier", "select + propose 1 (k-1 tries)", "barrier", "likelihood tiles (MFMA)", "barrier", "Metropolis step, trace"]
    print("persistent kernel, last generation of the last launch: %d waves; mean generation %d cycles" % (len(MG), (MG[:, 9] - MG[:, 0]).mean()))
    for i, nm in enumerate(names):
        dtt = MG[:, i + 1] - MG[:, i]
        print("   %-34s mean %6d  p5 %6d  p95 %6d" % (nm, dtt.mean(), np.percentile(dtt, 5), np.percentile(dtt, 95)))
    if MG[:, 10].min() > 0:
        seq = [(4, 10, "phase 1: slot draws (Philox)"), (10, 11, "  Q sums, lp"), (11, 12, "  mt_select_vals"), (12, 13, "  base row, tile 0, gamma row"),
               (13, 5, "  k-1 tries"), (8, 14, "Metropolis: operands"), (14, 15, "  mt_log_ratio, dlog(u), decision"), (15, 9, "  state, trace, history rows")]
        for a, b, nm in seq:
            dtt = MG[:, b] - MG[:, a]
            print("   %-34s mean %6d  p5 %6d  p95 %6d" % (nm, dtt.mean(), np.percentile(dtt, 5), np.percentile(dtt, 95)))
