// dz_experiments.h -- cycle-stamp instrumentation for timing experiments (tools/variants.sh, tools/stamps.py).
// Included only by builds made with -DDZ_EXPERIMENTS; the product library does not contain any of this.
#pragma once
#define DZ_STAMP(p_, phase_, c_, i_) do { if ((p_).dbg && (threadIdx.x & 63) == 0) (p_).dbg[((size_t)(phase_) * (p_).nl + (c_)) * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
#define DZ_LSTAMP(p_, w_, i_) do { if ((p_).dbg && (threadIdx.x & 63) == 0) (p_).dbg[((size_t)2 * (p_).nl + (w_)) * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
// inside k_generations: last generation of the launch, the chain's first wave
#define DZ_MSTAMP(i_) do { if (p.dbg && gi == ngen - 1 && lane == 0 && sub == 0) p.dbg[((size_t)3 * p.nl + blockIdx.x * CH + cl) * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
// k_generations_w4: the chain's LAST wave (two pre-tries) stamps into the region of k_propose's phase 0 (unused on the persistent path)
// (both waves' stamps of that kernel are taken in the launch's second-to-last generation: the last one makes nothing ahead)
#define DZ_W0STAMP(i_) do { if (p.dbg && gi == ngen - 2 && lane == 0 && sub == 0) p.dbg[((size_t)3 * p.nl + blockIdx.x * CH + cl) * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
#define DZ_WSTAMP(i_) do { if (p.dbg && gi == ngen - 2 && lane == 0 && sub == 3) p.dbg[((size_t)0 * p.nl + blockIdx.x * CH + cl) * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
