// MvNMF with more than 96 features (feature blocks): the W-only algebra and the line-search trials for a signature matrix
// of any width (reference arithmetic: mvnmf.py:19-24 volume_logdet, :37-66 update_W_unconstrained, :69-92 line_search;
// the reference has no limit on the number of features).  The K x K part -- Gram matrix W W^T + delta I, elimination,
// log det -- is salnmf_mv_device.h's, with the Gram matrix and the products A = Y_minus W, B = |Y| W reading W from
// global memory (K x V does not fit LDS beyond 96 features) in the same MFMA tiling: an entry's value is the chain over
// the feature k-steps in order, as in the 96-feature kernels.  Per-row kernels (one workgroup per signature) evaluate the
// closed-form root / the blend, the row sum in a fixed order (thread t: features t, t + 256, ...; then a binary tree),
// normalise and clip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "salnmf_kernels.h"
#include "salnmf_mv_device.h"
#include "salnmf_mv_kernels.h"

namespace salnmf {

// S (LDS, [K][MV_LD]) <- W W^T + delta I with W [K][V] in global memory
__device__ inline void mv_gram_wide(const double* __restrict__ W, double* S, int K, int V, double delta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
    const int KT = (K + 15) / 16, vsteps = (V + 3) / 4;
    for (int t = wave; t < KT * KT; t += MV_BLOCK / 64) {  // (uniform per wave)
        const int ti = t / KT, tj = t - ti * KT;
        const double* ra = W + (int64_t)min(16 * ti + c16, K - 1) * V;
        const double* rb = W + (int64_t)min(16 * tj + c16, K - 1) * V;
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

__global__ void __launch_bounds__(MV_BLOCK) mv_logdet_wide_kernel(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ out) {
    __shared__ double S[MV_KMAX * MV_LD];
    __shared__ double T[MV_KMAX * MV_LD];
    __shared__ double piv[MV_KMAX + 1];
    mv_gram_wide(W, S, K, V, delta);
    mv_eliminate<MV_BLOCK, false>(S, T, piv, K);
    const double ld = mv_logdet_from_pivots(piv, K, piv + K);
    if (threadIdx.x == 0) *out = ld;
}

// A = Y_minus W, B = |Y| W with Y = (W W^T + delta I)^-1, and the log det (mv_prepare_W_body for any V)
__global__ void __launch_bounds__(MV_BLOCK) mv_prepare_W_wide_kernel(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ Aout,
                                                                     double* __restrict__ Bout, double* __restrict__ logdet_out) {
    __shared__ double Sa[MV_KMAX * MV_LD];
    __shared__ double Ta[MV_KMAX * MV_LD];
    __shared__ double piv[MV_KMAX + 1];
    mv_gram_wide(W, Sa, K, V, delta);
    const double* S = mv_eliminate<MV_BLOCK, true>(Sa, Ta, piv, K);
    const double ld = mv_logdet_from_pivots(piv, K, piv + K);
    if (threadIdx.x == 0) *logdet_out = ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
    const int KT = (K + 15) / 16, VT6 = (V + 15) / 16, ksteps = (K + 3) / 4;
    for (int t = wave; t < KT * VT6; t += MV_BLOCK / 64) {  // (uniform per wave)
        const int ti = t / VT6, tj = t - ti * VT6;
        const int krow = min(16 * ti + c16, K - 1), vcol = min(16 * tj + c16, V - 1);
        mv_d4 accA = (mv_d4){0, 0, 0, 0}, accB = (mv_d4){0, 0, 0, 0};
        for (int s2 = 0; s2 < ksteps; ++s2) {
            const int m = 4 * s2 + q;
            const bool in = m < K;
            const double y = in ? S[m * MV_LD + krow] : 0.0;
            const double wv = in ? W[(int64_t)m * V + vcol] : 0.0;
            accA = __builtin_amdgcn_mfma_f64_16x16x4f64(fmax(0.0, -y), wv, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f64_16x16x4f64(fabs(y), wv, accB, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * ti + q + 4 * r, v = 16 * tj + c16;
            if (k < K && v < V) {
                Aout[(int64_t)k * V + v] = accA[r];
                Bout[(int64_t)k * V + v] = accB[r];
            }
        }
    }
}

// One signature row of a line-search trial (mvnmf.py:80-81, 85-88), one workgroup per row:
//   ROOT: W_unconstrained[k][:] from the closed-form root (mv_root_entry; G in the feature-block layout of the blocked
//         numerator passes, compact [K][vb] per block at stride K * 96), stored to Wunc;
//   else: the blend (1 - gamma) W + gamma W_unconstrained (blend != 0) or W_unconstrained itself;
// then the row sum -> cs[k] (the factor H is rescaled by), and the normalised, clipped row -> Wtrial.
template <bool ROOT>
__global__ void __launch_bounds__(256) mv_trial_row_wide_kernel(const double* __restrict__ W, double* __restrict__ Wunc, double gamma, int blend, int K, int V,
                                                                double* __restrict__ Wtrial, double* __restrict__ cs, const double* __restrict__ A,
                                                                const double* __restrict__ B, const double* __restrict__ Gblk,
                                                                const double* __restrict__ hsum, double lam, int n_given) {
    __shared__ double red[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t row = (int64_t)k * V;
    double part = 0.0;
    for (int v = tid; v < V; v += 256) {
        double wt;
        if (ROOT) {
            const int b = v / VMAX, vv = v - b * VMAX;
            const int vb = V - VMAX * b < VMAX ? V - VMAX * b : VMAX;
            const double g = Gblk[(int64_t)b * K * VMAX + k * vb + vv];
            wt = mv_root_entry(W[row + v], A[row + v], B[row + v], g, hsum[k], lam, k < n_given);
            Wunc[row + v] = wt;
        } else {
            wt = Wunc[row + v];
            if (blend) wt = (1 - gamma) * W[row + v] + gamma * wt;
        }
        Wtrial[row + v] = wt;  // (unnormalised for the moment)
        part += wt;
    }
    red[tid] = part;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    const double rs = red[0];
    if (tid == 0) cs[k] = rs;
    for (int v = tid; v < V; v += 256) Wtrial[row + v] = clip_lo(Wtrial[row + v] / rs, kEps);
}

}  // namespace salnmf

// ---- MvNMF with more than 64 signatures (signature chunks, round 5): the K x K part no longer fits one workgroup's LDS
// (K = 512: 2 MB), so it runs in global memory -- Gram matrix W W^T + delta I, Gauss-Jordan elimination without pivoting
// (symmetric positive definite for delta > 0, as in salnmf_mv_device.h) on the matrix augmented with the identity, log det
// from the pivots, then A = Y_minus W and B = |Y| W.  Plain kernels: the K^3 work is 2.7e8 multiply-adds at K = 512, small
// beside the passes over the samples of an engine that large.  Reference arithmetic: mvnmf.py:19-24, :48-54; the reference
// has no limit on n_signatures (mvnmf.py:116-126).
namespace salnmf {

constexpr int MVM_BLOCK = 1024;
constexpr int MVM_KMAX = 512;

// S[i][0..K) = (W W^T + delta I)[i][:], S[i][K..2K) = identity row (AUG) -- one workgroup per row, the dot products in
// the order v = 0 .. V-1
template <bool AUG>
__global__ void __launch_bounds__(256) mv_many_gram_kernel(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ S) {
    const int i = blockIdx.x, ld = AUG ? 2 * K : K;
    for (int j = threadIdx.x; j < K; j += 256) {
        double s = 0.0;
        for (int v = 0; v < V; ++v) s = __builtin_fma(W[(int64_t)i * V + v], W[(int64_t)j * V + v], s);
        S[(int64_t)i * ld + j] = i == j ? s + delta : s;
        if (AUG) S[(int64_t)i * ld + K + j] = i == j ? 1.0 : 0.0;
    }
}

// One workgroup.  AUG: Gauss-Jordan on [K][2K] -- afterwards the right half is the inverse; else forward elimination of the
// trailing block only (the pivots are all a log det needs).  logdet_out = sum_p log(pivot_p), p in order.
// At pivot p only the columns p .. K + p of the augmented matrix can change: left of the pivot row p is zero already, and right
// of column K + p the identity part has not been touched (row p is zero there), so the update x - f * 0 leaves those entries
// as they are -- they are skipped (half the work; the touched entries get the same fused multiply-adds as before).  Threads
// are laid out (row group, column): a row of the update is read and written in whole 128-byte lines, a thread's rows are
// independent loads the compiler keeps in flight, and there is no integer division in the loop.
template <bool AUG>
__global__ void __launch_bounds__(MVM_BLOCK) mv_many_eliminate_kernel(double* __restrict__ S, int K, double* __restrict__ logdet_out) {
    __shared__ double f[MVM_KMAX];          // column p of the matrix (the row multipliers)
    __shared__ double prow[MVM_KMAX + 1];   // row p divided by the pivot, columns c0 .. c0 + ncol - 1
    __shared__ double lp[MVM_KMAX];         // log of the pivots
    const int tid = threadIdx.x, ld = AUG ? 2 * K : K;
    constexpr int TX = 128, TY = MVM_BLOCK / TX;  // columns per pass x rows per pass
    const int tx = tid % TX, ty = tid / TX;
    for (int p = 0; p < K; ++p) {
        const double pivot = S[(int64_t)p * ld + p];
        const int c0 = p, r0 = AUG ? 0 : p + 1;
        const int ncol = AUG ? K + 1 : K - p;
        for (int c = tid; c < ncol; c += MVM_BLOCK) prow[c] = S[(int64_t)p * ld + c0 + c] / pivot;
        for (int r = r0 + tid; r < K; r += MVM_BLOCK) f[r] = S[(int64_t)r * ld + p];
        if (tid == 0) lp[p] = log(pivot);
        __syncthreads();
        for (int cb = 0; cb < ncol; cb += TX) {
            const int c = cb + tx;
            if (c < ncol) {
                const double pc = prow[c];
                for (int r = r0 + ty; r < K; r += TY) {
                    double* dst = S + (int64_t)r * ld + c0 + c;
                    if (r == p) *dst = pc;
                    else *dst = __builtin_fma(-f[r], pc, *dst);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += lp[k];
        *logdet_out = t;
    }
}

// A[k][v] = sum_m max(0, -Y[m][k]) W[m][v], B[k][v] = sum_m |Y[m][k]| W[m][v] (mvnmf.py:50-53), Y = the right half of the
// eliminated augmented matrix; one workgroup per signature row, m in order
__global__ void __launch_bounds__(128) mv_many_AB_kernel(const double* __restrict__ S, const double* __restrict__ W, int K, int V, double* __restrict__ A,
                                                         double* __restrict__ B) {
    const int k = blockIdx.x;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {  // (more than 96 features as well)
        double a = 0.0, b = 0.0;
        for (int m = 0; m < K; ++m) {
            const double y = S[(int64_t)m * 2 * K + K + k], w = W[(int64_t)m * V + v];
            a = __builtin_fma(fmax(0.0, -y), w, a);
            b = __builtin_fma(fabs(y), w, b);
        }
        A[(int64_t)k * V + v] = a;
        B[(int64_t)k * V + v] = b;
    }
}

}  // namespace salnmf
