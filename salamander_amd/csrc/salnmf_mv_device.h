// Device functions of the small dense (K x K, K <= 64) algebra of the MvNMF W step: one workgroup, fp64, all in LDS.
// Reference arithmetic: src/salamander/models/mvnmf.py:19-24 (volume_logdet), :37-66 (update_W_unconstrained).
// S = W W^T + delta I is symmetric positive definite (delta > 0), so the reference's LU-based inv/det are replaced by a
// Gauss-Jordan elimination without pivoting (same values to rounding).  Kernels: salnmf_mv_kernels.h; the spare workgroup of the MvNMF
// update_H pass (salnmf_fused_kernel.h: fused_kernel) runs mv_prepare_W_body too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace salnmf {

constexpr int MV_BLOCK = 1024;  // one workgroup, four waves per SIMD: the parallel phases are fp64-issue bound
constexpr int MV_KMAX = 64;
constexpr int MV_LD = MV_KMAX + 1;  // padded leading dimension in LDS
constexpr int MV_VMAX = 96;
constexpr int MV_WS = 97;  // LDS row stride of W: odd, so different signature rows fall into different banks

typedef double mv_d4 __attribute__((ext_vector_type(4)));

// S (LDS, [K][MV_LD]) <- Wl Wl^T + delta I, with Wl (LDS, [K][MV_WS]) rows = signatures.  NT = threads of the workgroup.
// On the matrix cores: one 16 x 16 output tile per wave at a time (v_mfma_f64_16x16x4: A = W[16 ti + c16][4 s + q],
// B = W[16 tj + c16][4 s + q], D register r = row q + 4 r, column c16), the feature axis in k-steps of 4 in order -- so an
// entry's value does not depend on NT or on which wave computes it, and S[a][b] == S[b][a] bit for bit (the same products
// in the same order).  Rows beyond K are clamped reads whose results are not stored.  (The VALU form -- one dot product per
// thread, two LDS reads per FMA -- took 6.5 us of the side workgroup's 40 at K = 30.)
template <int NT = MV_BLOCK>
__device__ inline void mv_gram(const double* Wl, double* S, int K, int V, double delta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
    const int KT = (K + 15) / 16, vsteps = (V + 3) / 4;
    for (int t = wave; t < KT * KT; t += NT / 64) {  // (uniform per wave)
        const int ti = t / KT, tj = t - ti * KT;
        const double* ra = Wl + min(16 * ti + c16, K - 1) * MV_WS;
        const double* rb = Wl + min(16 * tj + c16, K - 1) * MV_WS;
        mv_d4 acc = (mv_d4){0, 0, 0, 0};
        for (int s = 0; s < vsteps; ++s) {
            const int v = 4 * s + q;
            const double a = v < V ? ra[v] : 0.0, b = v < V ? rb[v] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + q + 4 * r, j = 16 * tj + c16;
            if (i < K && j < K) S[i * MV_LD + j] = i == j ? acc[r] + delta : acc[r];
        }
    }
    __syncthreads();
}

// Gauss-Jordan elimination without pivoting on the SPD matrix in `src` (LDS, [K][MV_LD]), the whole workgroup on every
// step, ping-pong between `src` and `dst` (one barrier per step):
//   step k:  d = A[k][k] (the k-th pivot -> piv[k]),  row k <- row k / d with 1/d on the diagonal,
//            row i != k <- row i - A[i][k] * (row k / d) with -A[i][k]/d in column k
// After K steps the matrix is the inverse.  FULL = false eliminates the trailing submatrix only: the same pivots, bit for
// bit (an element with i, j > k is computed by the same expression), which is all the log det needs.
// S is SPD with eigenvalues >= delta, so no pivoting is needed (the pivots are the squared diagonal of the Cholesky
// factor).  Every element is a fixed expression of the previous step's matrix: the result does not depend on NT.
// K steps of ~0.2 us replace the single-wave left-looking Cholesky (a chain of K dependent dot products, 20 us at
// K = 30) and the column-by-column triangular inverse behind it (14 us).  Returns the buffer that holds the result.
// Work split: NT / K threads share a row, each with a run of SEG = ceil(K / (NT / K)) consecutive columns, so a step
// costs a thread ONE read of its row's pivot-column entry, and per element the pivot row's entry (a broadcast among the
// threads of the other rows) and the element itself.  All LDS reads of a step are issued together: a step pays one LDS
// latency (the first form of this loop -- element by element with early exits -- paid one per element: 0.45 us per step at
// K = 30 with 256 threads, 13 of the side workgroup's 40 us).  EA = a power of two >= SEG.
template <int NT, bool FULL, int EA>
__device__ __forceinline__ double* mv_eliminate_n(double* src, double* dst, double* piv, int K) {
    const int tpr = NT / K > 0 ? NT / K : 1;  // threads per row (K <= NT)
    const int seg = (K + tpr - 1) / tpr;
    const int i = threadIdx.x / tpr, j0 = (threadIdx.x - i * tpr) * seg;
    const bool row_live = i < K;
    const int irow = row_live ? i : 0;
    bool have[EA];
    int col[EA];  // column of element e (clamped: reads stay in bounds)
#pragma unroll
    for (int e = 0; e < EA; ++e) {
        have[e] = row_live && e < seg && j0 + e < K;
        col[e] = have[e] ? j0 + e : 0;
    }
    for (int k = 0; k < K; ++k) {
        const double d = src[k * MV_LD + k];
        const double f = src[irow * MV_LD + k];
        double pr[EA], own[EA];
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            pr[e] = src[k * MV_LD + col[e]];
            own[e] = src[irow * MV_LD + col[e]];
        }
        const double rd = 1.0 / d;
        if (threadIdx.x == 0) piv[k] = d;
        const double nfrd = -f * rd;
#pragma unroll
        for (int e = 0; e < EA; ++e) {  // (selects, no branches: the EA chains advance side by side)
            const int j = col[e];
            const double r = pr[e] * rd;
            const double in_row_k = (j == k) ? rd : r;
            const double elsewhere = (j == k) ? nfrd : __builtin_fma(-f, r, own[e]);
            const double v = (i == k) ? in_row_k : elsewhere;
            if (have[e] && (FULL || (i > k && j > k))) dst[i * MV_LD + j] = v;
        }
        // (Keeping every thread's elements in registers and publishing only row / column k + 1 through LDS was measured
        // slower -- 61 -> 75 us for the side workgroup at K = 50: the step is bound by the issue of the per-element mask
        // and select logic, not by LDS traffic.)
        __syncthreads();
        double* t = src;
        src = dst;
        dst = t;
    }
    return src;
}
template <int NT, bool FULL>
__device__ inline double* mv_eliminate(double* src, double* dst, double* piv, int K) {
    constexpr int E = NT >= 1024 ? 4 : 16;  // SEG at K = MV_KMAX
    static_assert(NT == 256 || NT == 1024, "the side workgroup of the fused pass or a stand-alone kernel");
    const int tpr = NT / K > 0 ? NT / K : 1;
    const int seg = (K + tpr - 1) / tpr;  // (uniform)
    if (E >= 16 && seg > 8) return mv_eliminate_n<NT, FULL, (E >= 16 ? 16 : E)>(src, dst, piv, K);
    if (E >= 8 && seg > 4) return mv_eliminate_n<NT, FULL, (E >= 8 ? 8 : E)>(src, dst, piv, K);
    if (seg > 2) return mv_eliminate_n<NT, FULL, 4>(src, dst, piv, K);
    if (seg > 1) return mv_eliminate_n<NT, FULL, 2>(src, dst, piv, K);
    return mv_eliminate_n<NT, FULL, 1>(src, dst, piv, K);
}

// log det = sum_k log(pivot_k), fixed order; piv (LDS, [K]) is overwritten, `slot` = one LDS double of the caller's
__device__ inline double mv_logdet_from_pivots(double* piv, int K, double* slot) {
    if ((int)threadIdx.x < K) piv[threadIdx.x] = log(piv[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += piv[k];
        *slot = t;
    }
    __syncthreads();
    return *slot;
}

// compact W[K][V] (global) -> Wl[K][MV_WS] (LDS); all loads of a thread are issued before any use
template <int NT = MV_BLOCK>
__device__ inline void mv_load_W(const double* __restrict__ W, double* Wl, int K, int V) {
    constexpr int PT = (MV_KMAX * MV_VMAX + NT - 1) / NT;
    double w[PT];
    const int total = K * V;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int idx = threadIdx.x + NT * j;
        w[j] = idx < total ? W[idx] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int idx = threadIdx.x + NT * j;
        if (idx < total) {
            int k = idx / V, v = idx - k * V;
            Wl[k * MV_WS + v] = w[j];
        }
    }
    __syncthreads();
}

// ---- MvNMF W step, split so that everything that depends on W alone can run on a second stream while the
// passes over the samples run (salnmf_host_mv.h: mv_update_W_impl)

// W-only half of update_W_unconstrained: A = W @ Y_minus, B = W @ |Y| with Y = (W W^T + delta I)^-1
// (mvnmf.py:48-54, in the K x V layout), and log det(W W^T + delta I) (mvnmf.py:19-24).
// One workgroup of NT threads; LDS scratch: Wl [K][MV_WS], S [K][MV_LD], T [K][MV_LD], piv [K + 1].  The body is shared by
// the stand-alone kernel (1024 threads, side stream) and by the spare workgroup of the MvNMF update_H pass
// (salnmf_fused_kernel.h: fused_kernel, 256 threads); its results do not depend on NT, so both produce the same bits.
template <int NT>
__device__ __forceinline__ void mv_prepare_W_body(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ Aout,
                                                  double* __restrict__ Bout, double* __restrict__ logdet_out, double* Wl, double* S, double* T,
                                                  double* piv) {
    mv_load_W<NT>(W, Wl, K, V);
    mv_gram<NT>(Wl, S, K, V, delta);
    S = mv_eliminate<NT, true>(S, T, piv, K);  // Y = (W W^T + delta I)^-1
    const double ld = mv_logdet_from_pivots(piv, K, piv + K);
    if (threadIdx.x == 0) *logdet_out = ld;
    // A = Y_minus W and B = |Y| W on the matrix cores: one 16 x 16 output tile (signatures x features) per wave at a time,
    // both products side by side, the contraction over the K signatures in k-steps of 4 in order (entries beyond K are
    // zero operands); operand A is the entry Y[m][k] of the inverse, transformed on the fly (mvnmf.py:50-51)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
        const int KT = (K + 15) / 16, VT6 = (V + 15) / 16, ksteps = (K + 3) / 4;
        for (int t = wave; t < KT * VT6; t += NT / 64) {  // (uniform per wave)
            const int ti = t / VT6, tj = t - ti * VT6;
            const int krow = min(16 * ti + c16, K - 1), vcol = min(16 * tj + c16, V - 1);
            mv_d4 accA = (mv_d4){0, 0, 0, 0}, accB = (mv_d4){0, 0, 0, 0};
            for (int s2 = 0; s2 < ksteps; ++s2) {
                const int m = 4 * s2 + q;
                const bool in = m < K;
                const double y = in ? S[m * MV_LD + krow] : 0.0;
                const double wv = in ? Wl[m * MV_WS + vcol] : 0.0;
                accA = __builtin_amdgcn_mfma_f64_16x16x4f64(fmax(0.0, -y), wv, accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f64_16x16x4f64(fabs(y), wv, accB, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * ti + q + 4 * r, v = 16 * tj + c16;
                if (k < K && v < V) {
                    Aout[k * V + v] = accA[r];
                    Bout[k * V + v] = accB[r];
                }
            }
        }
    }
}

}  // namespace salnmf
