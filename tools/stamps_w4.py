#!/usr/bin/env python3
"""Cycle stamps of k_generations_w4 (instrumented build, -DDZ_EXPERIMENTS): last generation of the last launch, the chain's first wave
(selection, Metropolis step) beside its last wave (two pre-tries per set)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64)
nl = a.size // 64
A = a[3 * nl * 16:].reshape(-1, 16).astype(np.int64)      # wave 0 of every chain
W = a[:nl * 16].reshape(-1, 16).astype(np.int64)          # wave 3 of every chain
ok = (A[:, 0] > 0) & (A[:, 9] > A[:, 0]) & (W[:, 0] > 0) & (W[:, 9] > W[:, 0]) & (A[:, 9] - A[:, 0] < 2000000)
A, W = A[ok], W[ok]
t0 = A[:, 0]
print("%d chains; generation (wave 0): mean %d cycles" % (len(A), (A[:, 9] - A[:, 0]).mean()))
names = [(0, "generation start"), (1, "phase-0 tries written, ref rows requested"), (2, "barrier (points visible)"), (3, "likelihood units done"), (4, "barrier (q visible)"),
         (11, "w0: Q sums done / w3: ref pre-tries done"), (12, "w0: selection done"), (13, "w0: base in tile 0 / w3: at barrier"), (5, "ref tries written, next draws + rows requested"),
         (6, "barrier"), (7, "likelihood units done"), (8, "barrier"), (14, "w0: Metropolis operands / w3: next set's pre-tries done"), (15, "w0: decision"), (9, "end of generation")]
def mean_valid(S, i):          # (a stamp inside a branch is taken only by the chains that pass through it: a snooker chain leaves the slot stale)
    v = S[:, i] - t0
    ok = (v >= 0) & (v < 1000000)
    return ("%7d" % v[ok].mean()) if ok.any() else "      -"
for i, nm in names:
    print("  %-58s wave 0 %s   wave 3 %s" % (nm, mean_valid(A, i), mean_valid(W, i)))
