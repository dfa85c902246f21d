"""Timeline of k_logp_mvn_gemm's blocks (launch over all k tries) from gpurun_out/stamps.bin (build with -DDZ_EXPERIMENTS)."""
import sys
import numpy as np
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 512
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64)
B = a[2 * nl * 16:2 * nl * 16 + 8 * 1280 * 2].reshape(-1, 8)
B = B[B[:, 0] != 0]
t0, t1, hw, nch = B[:, 0].astype(np.int64), B[:, 1].astype(np.int64), B[:, 2], B[:, 3].astype(np.int64)
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
# every XCD has its own counter: times are taken from the XCD's first block start
for x in np.unique(xcc):
    m = xcc == x
    base = t0[m].min()
    t0[m] -= base; t1[m] -= base
T0 = 0
print("blocks", len(B), "span", t1.max() - T0, "ticks; start spread", t0.max() - T0)
print("distinct CUs", len(np.unique(cuid)), "blocks per CU min/max", np.bincount(np.unique(cuid, return_inverse=True)[1]).min(), np.bincount(np.unique(cuid, return_inverse=True)[1]).max())
dur = t1 - t0
for lo, hi in ((1, 8), (8, 16), (16, 32), (32, 48), (48, 64)):
    m = (nch >= lo) & (nch < hi)
    if m.any():
        print("chunks %2d-%2d: %4d blocks, duration mean %7d ticks, per chunk %6.1f, end mean %7d" % (lo, hi, m.sum(), dur[m].mean(), (dur[m] / nch[m]).mean(), (t1[m] - T0).mean()))
