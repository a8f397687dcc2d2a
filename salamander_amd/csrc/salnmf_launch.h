// Launchers of the heavy template kernels.  The instantiations live in translation units of their own
// (salnmf_fused_inst.hip, compiled once per geometry set, and salnmf_forward_inst.hip) so that the library builds in
// parallel: one translation unit with all ~170 instantiations of the fused pass took three minutes.  Every
// instantiation exists in exactly one translation unit; salnmf.hip only calls these functions.
#pragma once
#include <hip/hip_runtime.h>

#include "salnmf_kernels.h"
#include "salnmf_kernels_f32.h"

namespace salnmf {

constexpr int FUSED_GEOM_SETS = 6;

struct FusedSel {
    int KS, KTM, KR;               // geometry (salnmf.hip: pick_ks / salnmf_create)
    bool G, U, STATS, WTS;         // template switches of fused_kernel
    bool PERSIST;                  // the persistent multi-step variant (only in builds with SALNMF_WITH_PERSISTENT)
    bool BLOCKED;                  // one 96-feature block of a wider problem: the update_H pass that accumulates U over blocks
    bool RGIVEN = false;           // one chunk of <= 64 signatures of a wider problem: p.X holds the ratio X / (H W) over all of them
    bool MVJ = false;              // MvNMF: update_H and the numerator pass behind it in one pass (with G, U, STATS)
};

// Return 0 when the kernel was launched (HIP launch errors are left for hipGetLastError), 1 when this build has no
// such instantiation.  ev_start / ev_stop (may be null) are bound to the dispatch itself (hipExtLaunchKernelGGL).
int launch_fused_inst(const FusedSel& s, const FusedParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);
// (mode + FWD_PIN: the instantiation whose P starts from p.pin -- modes 0, 1, 2, 4)
constexpr int FWD_PIN = 16;
int launch_forward_inst(int KS, int mode, const FwdParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);

// Small cohorts (salnmf_small.hip): n_steps joint KL steps of an unweighted problem with at most SMALL_MAX_TILES tiles and
// 16 signatures in ONE one-workgroup launch; same bits as the per-step path.  0 = launched, 1 = shape not covered.
constexpr int SMALL_MAX_TILES = 64;
struct SmallParams {
    const double* __restrict__ X;  // [Np][VMAX]
    double* __restrict__ H;        // [Np][16] in / out
    const double* W;               // [K][V] in
    double* Wout;                  // [K][V] out (normally == W)
    double* G;                     // [K][V] the last step's reduced numerator (what the W tail leaves behind)
    int V, K, ntiles, nsteps, n_given, clip_mode;
};
int launch_small_kl_steps(int KS, const SmallParams& p, hipStream_t stream);
int launch_fused_f32_inst(int KS, const Fused32Params& p, int grid, hipStream_t stream);
bool built_with_persistent();

// one function per geometry set (salnmf_fused_inst.hip compiled with -DSALNMF_GEOM_SET=i)
#define SALNMF_DECLARE_SET(i) \
    int launch_fused_set##i(const FusedSel& s, const FusedParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);
SALNMF_DECLARE_SET(0)
SALNMF_DECLARE_SET(1)
SALNMF_DECLARE_SET(2)
SALNMF_DECLARE_SET(3)
SALNMF_DECLARE_SET(4)
SALNMF_DECLARE_SET(5)
#undef SALNMF_DECLARE_SET

}  // namespace salnmf
