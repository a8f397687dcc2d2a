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
