// Instantiations of the forward / objective pass (salnmf_forward_kernel.h: forward_kernel) and of the fp32 fast mode's
// fused pass (salnmf_kernels_f32.h), plus the dispatcher over the geometry sets of salnmf_fused_inst.hip.
#define SALNMF_TEMPLATES_ONLY 1
#include "salnmf_launch.h"

#include <hip/hip_ext.h>

namespace salnmf {

bool built_with_persistent() {
#ifdef SALNMF_WITH_PERSISTENT
    return true;
#else
    return false;
#endif
}

int launch_fused_inst(const FusedSel& s, const FusedParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    typedef int (*set_fn)(const FusedSel&, const FusedParams&, int, hipStream_t, hipEvent_t, hipEvent_t);
    static const set_fn sets[FUSED_GEOM_SETS] = {launch_fused_set0, launch_fused_set1, launch_fused_set2,
                                                 launch_fused_set3, launch_fused_set4, launch_fused_set5};
    for (set_fn f : sets)
        if (f(s, p, grid, stream, ev_start, ev_stop) == 0) return 0;
    return 1;
}

template <int KS, int MODE, bool PIN = false>
static void launch_fwd_one(const FwdParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const dim3 g(grid), b(BLOCK);
    if (ev_stop)
        hipExtLaunchKernelGGL((forward_kernel<KS, MODE, PIN>), g, b, 0, stream, ev_start, ev_stop, 0, p);
    else
        hipLaunchKernelGGL((forward_kernel<KS, MODE, PIN>), g, b, 0, stream, p);
}

template <int KS>
static int launch_fwd_mode(int mode, const FwdParams& p, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    switch (mode) {
        case 0: launch_fwd_one<KS, 0>(p, grid, st, e0, e1); return 0;
        case 1: launch_fwd_one<KS, 1>(p, grid, st, e0, e1); return 0;
        case 2: launch_fwd_one<KS, 2>(p, grid, st, e0, e1); return 0;
        case 3: launch_fwd_one<KS, 3>(p, grid, st, e0, e1); return 0;
        default: break;
    }
    if constexpr (KS >= 13) {  // (signature chunks use the geometries whose H layout is 64 columns wide)
        switch (mode) {
            case FWD_PIN + 0: launch_fwd_one<KS, 0, true>(p, grid, st, e0, e1); return 0;
            case FWD_PIN + 1: launch_fwd_one<KS, 1, true>(p, grid, st, e0, e1); return 0;
            case FWD_PIN + 2: launch_fwd_one<KS, 2, true>(p, grid, st, e0, e1); return 0;
            case FWD_PIN + 4: launch_fwd_one<KS, 4, true>(p, grid, st, e0, e1); return 0;
            default: break;
        }
    }
    return 1;
}

#define SALNMF_KS_LIST(X) X(1) X(2) X(4) X(8) X(10) X(13) X(16)

int launch_forward_inst(int KS, int mode, const FwdParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
#define SALNMF_CASE(ks) \
    if (KS == ks) return launch_fwd_mode<ks>(mode, p, grid, stream, ev_start, ev_stop);
    SALNMF_KS_LIST(SALNMF_CASE)
#undef SALNMF_CASE
    return 1;
}

int launch_fused_f32_inst(int KS, const Fused32Params& p, int grid, hipStream_t stream) {
#define SALNMF_CASE(ks)                                                                            \
    if (KS == ks) {                                                                                \
        hipLaunchKernelGGL((fused_f32_kernel<ks>), dim3(grid), dim3(BLOCK), 0, stream, p);         \
        return 0;                                                                                  \
    }
    SALNMF_KS_LIST(SALNMF_CASE)
#undef SALNMF_CASE
    return 1;
}

}  // namespace salnmf
