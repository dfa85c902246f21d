"""Block-level timeline of the persistent kernel's last generation from gpurun_out/stamps.bin (instrumented build)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64)
nl = a.size // 64
MG = a[3 * nl * 16:].reshape(-1, 16).astype(np.int64)
B = MG.reshape(-1, 16, 16)
names = ["gen start", "end propose0", "after barrier", "end mfma0", "after barrier", "end propose1", "after barrier", "end mfma1", "after barrier", "end metropolis"]
t0 = B[:, :, 0].min(1)
for i in range(10):
    print("%-16s first %7d  last %7d" % (names[i], (B[:, :, i].min(1) - t0).mean(), (B[:, :, i].max(1) - t0).mean()))
print("per-wave generation mean", (MG[:, 9] - MG[:, 0]).mean())
p0 = (MG[:, 1] - MG[:, 0]).reshape(-1, 16)
print("propose0 duration by wave:", " ".join("%d" % x for x in p0.mean(0)))
