// Instantiations of the fused update pass (salnmf_fused_kernel.h: fused_kernel) for one set of geometries.
// Compiled FUSED_GEOM_SETS times with -DSALNMF_GEOM_SET=0..5 (__graft_entry__.py: build) so that the sets build in
// parallel; each (geometry, variant) pair is instantiated in exactly one translation unit.
#define SALNMF_TEMPLATES_ONLY 1
#include "salnmf_launch.h"

#include <hip/hip_ext.h>

#ifndef SALNMF_GEOM_SET
#error "compile with -DSALNMF_GEOM_SET=<0..5>"
#endif

// (KS, KTM, KR): contraction depth in k-steps of 4, 16-wide MFMA tiles and VALU remainder columns on the output side
#if SALNMF_GEOM_SET == 0
#define SET_GEOMETRIES(X) X(13, 3, 0) X(13, 3, 1) X(8, 1, 1)
#define SET_FN launch_fused_set0
#elif SALNMF_GEOM_SET == 1
#define SET_GEOMETRIES(X) X(13, 3, 2) X(13, 3, 3) X(8, 1, 2)
#define SET_FN launch_fused_set1
#elif SALNMF_GEOM_SET == 2
#define SET_GEOMETRIES(X) X(13, 3, 4) X(16, 4, 0) X(8, 1, 3)
#define SET_FN launch_fused_set2
#elif SALNMF_GEOM_SET == 3
#define SET_GEOMETRIES(X) X(10, 3, 0) X(10, 2, 1) X(10, 2, 2) X(8, 1, 4)
#define SET_FN launch_fused_set3
#elif SALNMF_GEOM_SET == 4
#define SET_GEOMETRIES(X) X(10, 2, 3) X(10, 2, 4) X(4, 1, 0)
#define SET_FN launch_fused_set4
#elif SALNMF_GEOM_SET == 5
#define SET_GEOMETRIES(X) X(8, 2, 0) X(1, 1, 0) X(2, 1, 0)
#define SET_FN launch_fused_set5
#else
#error "unknown geometry set"
#endif

namespace salnmf {

template <int KS, int KTM, int KR, bool G, bool U, bool STATS, bool WTS, bool PERSIST = false, bool BLOCKED = false, bool RGIVEN = false, bool MVJ = false>
static void launch_one(const FusedParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const dim3 g(grid), b(BLOCK);
    if (ev_stop)
        hipExtLaunchKernelGGL((fused_kernel<KS, KTM, KR, G, U, STATS, WTS, PERSIST, BLOCKED, RGIVEN, MVJ>), g, b, 0, stream, ev_start, ev_stop, 0, p);
    else
        hipLaunchKernelGGL((fused_kernel<KS, KTM, KR, G, U, STATS, WTS, PERSIST, BLOCKED, RGIVEN, MVJ>), g, b, 0, stream, p);
}

template <int KS, int KTM, int KR>
static int launch_geometry(const FusedSel& s, const FusedParams& p, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    if (s.PERSIST) {
#ifdef SALNMF_WITH_PERSISTENT
        if (!(s.G && s.U && !s.STATS && !s.WTS)) return 1;
        launch_one<KS, KTM, KR, true, true, false, false, true>(p, grid, st, e0, e1);
        return 0;
#else
        return 1;
#endif
    }
    if (s.RGIVEN) {  // n_signatures > 64: one chunk's passes on the given ratio (weights honoured at run time)
        // (chunks use the geometries whose H layout is 64 columns wide: salnmf.hip, salnmf_create)
        if constexpr (KS >= 13) {
            if (s.STATS) return 1;
            if (s.BLOCKED) {  // > 96 features as well: the passes of one (chunk, feature block) pair on that block's ratio
                if (!s.U) return 1;
                if (s.G) launch_one<KS, KTM, KR, true, true, false, true, false, true, true>(p, grid, st, e0, e1);
                else launch_one<KS, KTM, KR, false, true, false, true, false, true, true>(p, grid, st, e0, e1);
                return 0;
            }
            if (s.G && s.U) launch_one<KS, KTM, KR, true, true, false, true, false, false, true>(p, grid, st, e0, e1);
            else if (s.U) launch_one<KS, KTM, KR, false, true, false, true, false, false, true>(p, grid, st, e0, e1);
            else if (s.G) launch_one<KS, KTM, KR, true, false, false, true, false, false, true>(p, grid, st, e0, e1);
            else return 1;
            return 0;
        } else {
            return 1;
        }
    }
    if (s.MVJ) {  // MvNMF: update_H + the numerator pass behind it in one pass
        if (!(s.G && s.U && s.STATS && !s.WTS)) return 1;
        launch_one<KS, KTM, KR, true, true, true, false, false, false, false, true>(p, grid, st, e0, e1);
        return 0;
    }
    if (s.BLOCKED) {  // n_features > 96: the update_H pass over one feature block, alone or with the block's numerator (weights honoured at run time)
        if (!(s.U && !s.STATS)) return 1;
        if (s.G) launch_one<KS, KTM, KR, true, true, false, true, false, true>(p, grid, st, e0, e1);
        else launch_one<KS, KTM, KR, false, true, false, true, false, true>(p, grid, st, e0, e1);
        return 0;
    }
    // the variants the engine uses: joint step / update_H / update_W, each weighted or not; the two MvNMF passes
    // with statistics (unweighted); the joint step that evaluates the objective of the state it starts from
#define SALNMF_VARIANT(g_, u_, s_, w_)                                            \
    if (s.G == g_ && s.U == u_ && s.STATS == s_ && s.WTS == w_) {                 \
        launch_one<KS, KTM, KR, g_, u_, s_, w_>(p, grid, st, e0, e1);             \
        return 0;                                                                 \
    }
    SALNMF_VARIANT(true, true, false, false)
    SALNMF_VARIANT(true, true, false, true)
    SALNMF_VARIANT(false, true, false, false)
    SALNMF_VARIANT(false, true, false, true)
    SALNMF_VARIANT(true, false, false, false)
    SALNMF_VARIANT(true, false, false, true)
    SALNMF_VARIANT(true, false, true, false)
    SALNMF_VARIANT(false, true, true, false)
    SALNMF_VARIANT(true, true, true, false)
#undef SALNMF_VARIANT
    return 1;
}

int SET_FN(const FusedSel& s, const FusedParams& p, int grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
#define SALNMF_CASE(ks, ktm, kr) \
    if (s.KS == ks && s.KTM == ktm && s.KR == kr) return launch_geometry<ks, ktm, kr>(s, p, grid, stream, ev_start, ev_stop);
    SET_GEOMETRIES(SALNMF_CASE)
#undef SALNMF_CASE
    return 1;
}

}  // namespace salnmf
