// dz_mega_launch.h -- the launch interface between dz_engine.hip and the translation units that hold the instantiations of the
// persistent generation kernel (dz_mega_tu.hip, compiled once per row-tile count NRT = ld / 16 so that the ~300 instantiations
// build in parallel: pydream_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dz {

struct Params;
struct Publish;

struct MegaLaunch {
    bool tri, xlds, pb, k1, redo;     // (redo: redraw rounds inside the kernel, with pb and multitry on)
    bool ahead = false;               // 4 chains x 4 waves, lean, multitry 3..6: k_generations_w4 (the tries' base-independent halves made ahead)
    bool sp = false;                  // k_generations_d2, 16 chains per block whose proposal set goes through the point tiles in two passes (229..256 dimensions at 5 tries)
    bool multi = false;               // adapt_lag >= 1: several burn-in generations per launch (16 chains per block, one wave each, states in LDS: the MG instantiations)
    // triangular factor / chain states in LDS / full proposal code (priors, bounds, DEpairs > 1) / multitry off
    int ch, wpc;                // chains per block (16, 8, 4), waves per chain (1; 4 at 4 chains per block with multitry on)
    dim3 grid, block; size_t lds; hipStream_t st;
    hipEvent_t ka, kb;          // the launch's own start / stop events (profiling pass) or null
    const Params* pp; uint32_t g; int n; uint32_t M; int64_t slot0; int64_t zappend; int seg0 = 0; const Publish* publish;      // seg0: generations up to and including the launch's first history append
};

// returns a static string naming the instantiation that was launched ("k_generations<7,tri,xlds,16,1,lean>")
#define DZ_MEGA_DECL(N_) const char* mega_launch_nrt##N_(const MegaLaunch& a);
DZ_MEGA_DECL(1) DZ_MEGA_DECL(2) DZ_MEGA_DECL(3) DZ_MEGA_DECL(4) DZ_MEGA_DECL(5) DZ_MEGA_DECL(6) DZ_MEGA_DECL(7) DZ_MEGA_DECL(8)
DZ_MEGA_DECL(9) DZ_MEGA_DECL(10) DZ_MEGA_DECL(11) DZ_MEGA_DECL(12) DZ_MEGA_DECL(13) DZ_MEGA_DECL(14) DZ_MEGA_DECL(15) DZ_MEGA_DECL(16)      // (k_generations_d2)
const char* mega_launch_d2_nrt1(const MegaLaunch& a); const char* mega_launch_d2_nrt2(const MegaLaunch& a);      // (k_generations_d2 at ld <= 128: where 16 chains'
const char* mega_launch_d2_nrt3(const MegaLaunch& a); const char* mega_launch_d2_nrt4(const MegaLaunch& a);      //  point tiles do not fit next to the matrix in LDS,
const char* mega_launch_d2_nrt5(const MegaLaunch& a); const char* mega_launch_d2_nrt6(const MegaLaunch& a);      //  and for more than 15 tries)
const char* mega_launch_d2_nrt7(const MegaLaunch& a); const char* mega_launch_d2_nrt8(const MegaLaunch& a);
#undef DZ_MEGA_DECL

}  // namespace dz
