// dz_mega_tu.hip -- the instantiations of k_generations (dz_megakernel.h) for ONE row-tile count, -DDZ_TU_NRT=1..8.
#define DZ_TEMPLATES_ONLY
#include "dz_megakernel.h"
#include "dz_megakernel_w4.h"
#include "dz_megakernel_d2.h"
#include "dz_mega_launch.h"
#include <hip/hip_ext.h>

#ifndef DZ_TU_NRT
#error "compile with -DDZ_TU_NRT=<1..16>"
#endif

namespace dz {

#define DZ_CAT_(a, b) a##b
#define DZ_CAT(a, b) DZ_CAT_(a, b)
#define DZ_STR_(x) #x
#define DZ_STR(x) DZ_STR_(x)

#if DZ_TU_NRT > 8
// 128 < ld <= 256: k_generations_d2 (dz_megakernel_d2.h) -- the matrix read from L2, two 128-dimension chunks per lane
const char* DZ_CAT(mega_launch_nrt, DZ_TU_NRT)(const MegaLaunch& a)
{
#define DZ_D2(K1_, NAME_)                                                                                                                 \
    do {                                                                                                                                   \
        if (a.pb) {                                                                                                                        \
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 16, K1_, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, \
                                  a.slot0, a.zappend, a.seg0, *a.publish);                                                                         \
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full" NAME_ ">";                                                 \
        }                                                                                                                                  \
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 16, K1_>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, \
                              a.slot0, a.zappend, a.seg0, *a.publish);                                                                             \
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean" NAME_ ">";                                                     \
    } while (0)
    if (a.k1) DZ_D2(true, ",k1");          // (the host only sends the triangular factor at 16 chains per block here: mega_d2_chains)
    if (a.sp && a.ch == 16) {              // 16 chains per block, the proposal set in two passes over k - 1 point tiles (round 6)
        if (a.pb) {
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 16, false, true, 1, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                                  a.slot0, a.zappend, a.seg0, *a.publish);
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full,two-pass>";
        }
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 16, false, false, 1, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                              a.slot0, a.zappend, a.seg0, *a.publish);
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean,two-pass>";
    }
    if (a.ch == 4) {                       // ... 4 chains x 4 waves where not even 8 chains' tiles fit (256 dimensions at 8 tries)
        if (a.pb) {
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 4, false, true, 4>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                                  a.slot0, a.zappend, a.seg0, *a.publish);
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full>";
        }
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 4, false, false, 4>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                              a.slot0, a.zappend, a.seg0, *a.publish);
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean>";
    }
    if (a.ch == 8 && a.sp) {               // 8 chains x 2 waves with the two-pass set (256 dimensions at 8 tries)
        if (a.pb) {
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 8, false, true, 2, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                                  a.slot0, a.zappend, a.seg0, *a.publish);
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full,two-pass>";
        }
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 8, false, false, 2, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                              a.slot0, a.zappend, a.seg0, *a.publish);
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean,two-pass>";
    }
    if (a.ch == 8) {                       // ... or, where the point tiles of 16 chains do not fit LDS (d > ~228 at 5 tries), at 8 chains x 2 waves (round 6)
        if (a.pb) {
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 8, false, true, 2>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                                  a.slot0, a.zappend, a.seg0, *a.publish);
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full>";
        }
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 8, false, false, 2>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M,
                              a.slot0, a.zappend, a.seg0, *a.publish);
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean>";
    }
    DZ_D2(false, "");
#undef DZ_D2
}
#else
template <bool TRI, bool X, int CH, int WPC, bool PB, bool K1>
static const char* launch_one(const MegaLaunch& a)
{
    if constexpr (PB && !K1) {
        if (a.redo) {
            hipExtLaunchKernelGGL((k_generations<DZ_TU_NRT, TRI, X, CH, WPC, PB, K1, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0,
                                  a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
            return TRI ? "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,full,redo>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,full,redo>";
        }
    }
    if constexpr (X && CH == 16 && WPC == 1 && !K1) {
        if (a.multi) {      // adapt_lag >= 1: several burn-in generations per launch
            hipExtLaunchKernelGGL((k_generations<DZ_TU_NRT, TRI, X, CH, WPC, PB, K1, false, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0,
                                  a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
            return TRI ? (PB ? "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,full,multi>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,lean,multi>")
                       : (PB ? "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,full,multi>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,lean,multi>");
        }
    }
    hipExtLaunchKernelGGL((k_generations<DZ_TU_NRT, TRI, X, CH, WPC, PB, K1>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0,
                          a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
    // (NRT, matrix, chain states, chains per block, waves per chain, proposal code)
    return TRI ? (X ? (PB ? (K1 ? "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,full,k1>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,full>")
                          : (K1 ? "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,lean,k1>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,lean>"))
                      : (K1 ? "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean,k1>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean>"))
               : (X ? (PB ? (K1 ? "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,full,k1>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,full>")
                          : (K1 ? "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,lean,k1>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,lean>"))
                      : (K1 ? "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xhbm,%d,%d,lean,k1>" : "k_generations<" DZ_STR(DZ_TU_NRT) ",dense,xhbm,%d,%d,lean>"));
}

template <bool TRI, bool X, bool PB>
static const char* launch_ch(const MegaLaunch& a)
{
    if constexpr (!PB && X) {
        if (a.ahead && a.ch == 4 && !a.k1) {      // small populations: four waves per chain, the tries' base-independent halves made ahead (dz_megakernel_w4.h)
            hipExtLaunchKernelGGL((k_generations_w4<DZ_TU_NRT, TRI>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
            return TRI ? "k_generations_w4<" DZ_STR(DZ_TU_NRT) ",tri,xlds,%d,%d,lean,ahead>" : "k_generations_w4<" DZ_STR(DZ_TU_NRT) ",dense,xlds,%d,%d,lean,ahead>";
        }
    }
#ifdef DZ_TU_W4ONLY    // (compile-time experiments on k_generations_w4 alone)
    return nullptr;
#else
#ifdef DZ_TU_FAST      // experiment builds (tools/fastbuild.sh): multi-try only, 16 chains per block or 4 x 4 waves -- a third of the instantiations
    if (a.ch == 4) return launch_one<TRI, X, 4, 4, PB, false>(a);
    return launch_one<TRI, X, 16, 1, PB, false>(a);
#endif
    if (a.k1) {
        if (a.ch == 16) return launch_one<TRI, X, 16, 1, PB, true>(a);
        if (a.ch == 8) return launch_one<TRI, X, 8, 1, PB, true>(a);
        return launch_one<TRI, X, 4, 1, PB, true>(a);
    }
    if (a.ch == 16) return launch_one<TRI, X, 16, 1, PB, false>(a);
    if (a.ch == 12) return launch_one<TRI, X, 12, 1, PB, false>(a);      // (round 6: 2049 .. 3072 chains on 256 CUs -- multi-try only)
    if (a.ch == 8) return launch_one<TRI, X, 8, 1, PB, false>(a);
    return launch_one<TRI, X, 4, 4, PB, false>(a);
#endif
}

// (priors / boundaries / several pairs: only with the chain states in LDS -- mega_eligible -- which keeps the number of kernels down)
const char* DZ_CAT(mega_launch_nrt, DZ_TU_NRT)(const MegaLaunch& a)
{
    if (a.pb) return a.tri ? launch_ch<true, true, true>(a) : launch_ch<false, true, true>(a);
    if (a.tri) return a.xlds ? launch_ch<true, true, false>(a) : launch_ch<true, false, false>(a);
    return a.xlds ? launch_ch<false, true, false>(a) : launch_ch<false, false, false>(a);
}
// d <= 128 with the point tiles of 16 chains NOT fitting next to the matrix in LDS (113..128 dimensions at 5 tries, 100 dimensions at 8 or more), and
// (round 6) more than 15 tries at any d <= 128: k_generations_d2 with one chunk per lane -- 16 chains per block, or 8 chains x 2 waves where the tiles of 16
// do not fit even without the matrix (100 dimensions at 16..20 tries)
const char* DZ_CAT(mega_launch_d2_nrt, DZ_TU_NRT)(const MegaLaunch& a)
{
    if (a.ch == 4) {                       // 4 chains x 4 waves: 24..32 tries at 100 dimensions, 20..32 at 128
        if (a.pb) {
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 4, false, true, 4>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full>";
        }
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 4, false, false, 4>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean>";
    }
    if (a.ch == 8) {
        if (a.pb) {
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 8, false, true, 2>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full>";
        }
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 8, false, false, 2>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish);
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean>";
    }
#define DZ_D2(K1_, NAME_)                                                                                                                 \
    do {                                                                                                                                   \
        if (a.pb) {                                                                                                                        \
            hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 16, K1_, true>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish); \
            return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,full" NAME_ ">";                                                 \
        }                                                                                                                                  \
        hipExtLaunchKernelGGL((k_generations_d2<DZ_TU_NRT, true, 16, K1_>), a.grid, a.block, a.lds, a.st, a.ka, a.kb, 0, a.pp, a.g, a.n, a.M, a.slot0, a.zappend, a.seg0, *a.publish); \
        return "k_generations_d2<" DZ_STR(DZ_TU_NRT) ",tri,xhbm,%d,%d,lean" NAME_ ">";                                                     \
    } while (0)
    DZ_D2(false, "");          // (multitry on only: the multitry-off kernels' 16-chain layout always fits)
#undef DZ_D2
}
#endif      // DZ_TU_NRT <= 8

}  // namespace dz
