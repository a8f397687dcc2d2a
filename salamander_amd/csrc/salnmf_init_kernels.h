// Device kernels of the initialisation step: what the reference computes on the
// host before the first update (src/salamander/initialization/methods.py:58-86, initialize.py:116-118).
//
// NNDSVD needs the rank-K truncated SVD of X (n_samples x n_features, n_features <= 96).  With so few
// features the right singular vectors are the eigenvectors of the 96 x 96 Gram matrix X^T X: ONE pass over X on
// the fp64 MFMA units (gram_kernel; all-reduced when the samples are sharded), a tiny eigendecomposition on the
// host, and the left factor as the projection X V (init_project_kernel), whose columns the NNDSVD sign split
// needs only through the norms of their positive and negative parts (same kernel).  init_finish_kernel then writes
// the exposures in the engine's padded layout: sign part, sklearn's zero threshold, the "nndsvda" fill, the scaling
// of normalize_WH and the EPSILON clip, all per element.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "salnmf_kernels.h"

namespace salnmf {

constexpr int GRAM_TILES = VT * (VT + 1) / 2;       // upper-triangular 16 x 16 tiles of the 96 x 96 matrix
constexpr int GRAM_PART = GRAM_TILES * 4 * 64;      // doubles per wave partial: [tile][register][lane]

// part[(blockIdx.x * WAVES + wave)][tile][r][lane]: this wave's partial of the upper triangle of X^T X, tile (vt, vt')
// with vt <= vt' holding G[16 vt + (lane >> 4) + 4 r][16 vt' + (lane & 15)];  xsum_part[...] = the wave's sum of X.
// X is the engine's padded [Np][VMAX] layout (pads are zero, so they contribute nothing).
__global__ void __launch_bounds__(BLOCK) gram_kernel(const double* __restrict__ X, int64_t ntiles, double* __restrict__ part,
                                                      double* __restrict__ xsum_part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    d4 acc[GRAM_TILES];
#pragma unroll
    for (int i = 0; i < GRAM_TILES; ++i) acc[i] = (d4){0, 0, 0, 0};
    double xs = 0.0;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += stride) {
        const double* src = X + (tile * 16 + q) * VMAX + c16;
        double xv[4][VT];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) xv[s][vt] = src[4 * s * VMAX + 16 * vt];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // A[i = feature c16 of tile vt][k = sample q of this k-step] and B[k = sample q][j = feature c16 of tile vt']
            // are the SAME register: lane (c16, q) holds X[n0 + 4 s + q][16 vt + c16]
            int idx = 0;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                xs += xv[s][vt];
#pragma unroll
                for (int wt = vt; wt < VT; ++wt, ++idx) acc[idx] = mfma(xv[s][vt], xv[s][wt], acc[idx]);
            }
        }
    }
    double* out = part + ((int64_t)blockIdx.x * WAVES + wave) * GRAM_PART + lane;
#pragma unroll
    for (int i = 0; i < GRAM_TILES; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(i * 4 + r) * 64] = acc[i][r];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) xs += __shfl_xor(xs, m, 64);
    if (lane == 0) xsum_part[(int64_t)blockIdx.x * WAVES + wave] = xs;
}

// More than 96 features: the off-diagonal blocks of X^T X.  Xa, Xb are two 96-feature blocks of X (padded layout each);
// this launch forms the tiles (vt, wt), vt in [3 HALF, 3 HALF + 3), wt in [0, 6) of Xa^T Xb -- half of the block's 36
// tiles, which is what the accumulators of one wave hold (the diagonal blocks go through gram_kernel).
//   part[(blockIdx.x * WAVES + wave)][tile = 6 (vt - 3 HALF) + wt][r][lane] = G[16 vt + (lane >> 4) + 4 r][16 wt + (lane & 15)]
constexpr int GRAMX_TILES = 3 * VT;
constexpr int GRAMX_PART = GRAMX_TILES * 4 * 64;
template <int HALF>
__global__ void __launch_bounds__(BLOCK) gram_cross_kernel(const double* __restrict__ Xa, const double* __restrict__ Xb, int64_t ntiles,
                                                            double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    d4 acc[GRAMX_TILES];
#pragma unroll
    for (int i = 0; i < GRAMX_TILES; ++i) acc[i] = (d4){0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += stride) {
        const double* sa = Xa + (tile * 16 + q) * VMAX + c16 + 48 * HALF;
        const double* sb = Xb + (tile * 16 + q) * VMAX + c16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double xa[3], xb[VT];
#pragma unroll
            for (int vt = 0; vt < 3; ++vt) xa[vt] = sa[4 * s * VMAX + 16 * vt];
#pragma unroll
            for (int wt = 0; wt < VT; ++wt) xb[wt] = sb[4 * s * VMAX + 16 * wt];
#pragma unroll
            for (int vt = 0; vt < 3; ++vt)
#pragma unroll
                for (int wt = 0; wt < VT; ++wt) acc[VT * vt + wt] = mfma(xa[vt], xb[wt], acc[VT * vt + wt]);
        }
    }
    double* out = part + ((int64_t)blockIdx.x * WAVES + wave) * GRAMX_PART + lane;
#pragma unroll
    for (int i = 0; i < GRAMX_TILES; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(i * 4 + r) * 64] = acc[i][r];
}

// H[n][j] = sum_v X[n][v] B[j][v] into the padded exposure layout [Np][KP] (pad rows 1, pad columns 0, as
// salnmf_upload_H leaves them), and per workgroup the sums of squares of the positive and of the negative parts of
// every column: posneg_part[blockIdx.x][j] and [KP + j].
constexpr int PROJ_LD = VMAX + 1;  // odd row stride in LDS: conflict-free column access
//   feature blocks (n_features > 96): one launch per block; X is that block, B points at the block's first column of the
//   K x n_features operand (row stride ldb), V is the block's width.  `first`: the sum starts here; otherwise it continues
//   from H.  `last`: the total is final -- filler, norms; otherwise the running sum goes back to H.
__global__ void __launch_bounds__(256) init_project_kernel(const double* __restrict__ X, const double* __restrict__ B, double* __restrict__ H,
                                                           int64_t N, int64_t ntiles, int V, int ldb, int K, int KP, double* __restrict__ posneg_part,
                                                           int first, int last) {
    extern __shared__ __attribute__((aligned(16))) double plds[];
    double* Bl = plds;                 // [KP][PROJ_LD], rows >= K zero
    double* Xl = plds + KP * PROJ_LD;  // [16][PROJ_LD]
    double* red = Xl + 16 * PROJ_LD;   // [256]
    const int tid = threadIdx.x, r = tid >> 4, jg = tid & 15;
    for (int i = tid; i < KP * VMAX; i += 256) {
        const int j = i / VMAX, v = i - j * VMAX;
        Bl[j * PROJ_LD + v] = (j < K && v < V) ? B[(int64_t)j * ldb + v] : 0.0;
    }
    const int JJ = KP / 16;
    double pos2[4] = {0, 0, 0, 0}, neg2[4] = {0, 0, 0, 0};
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * 16;
        __syncthreads();  // the previous tile's readers are done (and Bl is complete)
        for (int i = tid; i < 16 * VMAX; i += 256) {
            const int rr = i / VMAX, v = i - rr * VMAX;
            Xl[rr * PROJ_LD + v] = X[(n0 + rr) * VMAX + v];
        }
        __syncthreads();
        const int64_t n = n0 + r;
        for (int jj = 0; jj < JJ; ++jj) {
            const int j = jg + 16 * jj;
            const double* xr = Xl + r * PROJ_LD;
            const double* br = Bl + j * PROJ_LD;
            const bool valid = n < N && j < K;
            double s = (first || !valid) ? 0.0 : H[n * KP + j];  // (blocks after the first continue the sum: block 0 + 1 + ...)
            for (int v = 0; v < VMAX; ++v) s = __builtin_fma(xr[v], br[v], s);
            H[n * KP + j] = valid ? s : (j < K ? 1.0 : 0.0);
            if (valid && last) {
                const double p = s > 0.0 ? s : 0.0, m = s < 0.0 ? s : 0.0;
                pos2[jj] = __builtin_fma(p, p, pos2[jj]);
                neg2[jj] = __builtin_fma(m, m, neg2[jj]);
            }
        }
    }
    // fixed-order sums over the 16 row-threads that share a column group
    for (int which = 0; which < 2; ++which)
        for (int jj = 0; jj < JJ; ++jj) {
            __syncthreads();
            red[tid] = which == 0 ? pos2[jj] : neg2[jj];
            __syncthreads();
            if (tid < 16) {
                double s = 0.0;
                for (int rr = 0; rr < 16; ++rr) s += red[rr * 16 + tid];
                posneg_part[(int64_t)blockIdx.x * 2 * KP + which * KP + 16 * jj + tid] = s;
            }
        }
}

// sklearn's _initialize_nmf after the SVD (_nmf.py: the NNDSVD loop, `W[W < eps] = 0`, the "nndsvda" fill), then
// the reference's post-processing (initialize.py:116-118): exposures * column sums of the signatures, clip.
//   j == 0:  |x| * scale[0]                       otherwise: scale[j] * (take_neg[j] ? max(-x, 0) : max(x, 0))
struct InitFinishParams {
    double* __restrict__ H;              // [Np][KP] in / out
    const double* __restrict__ scale;    // [K]
    const int* __restrict__ take_neg;    // [K]
    const double* __restrict__ post;     // [K] column sums of the raw signatures (normalize_WH)
    double zero_below;                   // sklearn's eps (1e-6): smaller entries become 0
    double fill;                         // "nndsvda": value of the zeros (X.mean()); 0 = leave them
    int64_t N, Np;
    int K, KP;
    int first_component;                 // column 0 of this H is the first component of the SVD (signature chunks: chunk 0 only)
};
__global__ void init_finish_kernel(InitFinishParams p) {
    const int64_t total = p.Np * p.KP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / p.KP;
        const int j = (int)(i - n * p.KP);
        if (n >= p.N || j >= p.K) continue;  // pads keep the filler init_project_kernel wrote
        const double x = p.H[i];
        double v;
        if (j == 0 && p.first_component) v = fabs(x) * p.scale[0];
        else v = p.scale[j] * (p.take_neg[j] ? (x < 0.0 ? -x : 0.0) : (x > 0.0 ? x : 0.0));
        if (v < p.zero_below) v = 0.0;
        if (p.fill != 0.0 && v == 0.0) v = p.fill;
        p.H[i] = clip_lo(v * p.post[j], kEps);
    }
}

// init_flat (methods.py:58-66) + post-processing: exposure (n, j) = clip(rowsum(X_n) / K * post[j])
//   nb feature blocks of X at stride Np * VMAX (pads are zero): the row sum runs over all of them, block after block
//   K: columns of this H (one chunk of the signatures), Ktot: n_signatures
__global__ void init_flat_kernel(const double* __restrict__ X, double* __restrict__ H, int64_t N, int64_t Np, int K, int KP,
                                 const double* __restrict__ post, int nb, int Ktot) {
    const int lane = threadIdx.x & 15;
    const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t rstride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t n = row0; n < Np; n += rstride) {  // 16 lanes per row (Np is a multiple of 16: whole waves stay in step)
        double s = 0.0;
        for (int b = 0; b < nb; ++b)
            for (int v = lane; v < VMAX; v += 16) s += X[((int64_t)b * Np + n) * VMAX + v];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
        const double e = s / Ktot;
        for (int j = lane; j < KP; j += 16) H[n * KP + j] = (j < K) ? (n < N ? clip_lo(e * post[j], kEps) : 1.0) : 0.0;
    }
}

// ---- separableNMF (successive projection, Gillis & Vavasis 2013): the signature side of
// src/salamander/initialization/methods.py:112-135.  R = X^T with every sample normalised to sum 1; K rounds of
//   j = argmax_n |R_n|^2 (lowest index on ties, as np.argmax),  u = R_j,  R_n <- R_n - (u <u, R_n>) / |u|^2
// The reference forms the V x V projector's action through a V x N temporary per round (seconds at c2, in front of a
// 39 ms loop); here a round is ONE pass over the resident R (deflation by the previous round's u fused with the next
// round's norms and per-workgroup argmax) and a one-workgroup selection kernel.  16 lanes per sample row: coalesced
// 128-byte row segments, reductions over the 16 lanes by shuffles.
constexpr int SEP_BLOCK = 256;  // 16 samples x 16 lanes
constexpr int SEP_STATE = 128;  // doubles: u[96] | |u|^2 at [96]

template <bool INIT>
__global__ void __launch_bounds__(SEP_BLOCK) sep_pass_kernel(const double* __restrict__ X, double* __restrict__ R, int64_t N, int V,
                                                              const double* __restrict__ state, double* __restrict__ pval,
                                                              long long* __restrict__ pidx) {
    __shared__ double bval[16];
    __shared__ long long bidx[16];
    const int slot = threadIdx.x >> 4, c = threadIdx.x & 15;
    double u[VT], un = 1.0;
    if (!INIT) {
#pragma unroll
        for (int t = 0; t < VT; ++t) u[t] = state[c + 16 * t];
        un = state[VMAX];
    }
    double best = -1.0;
    long long besti = 0x7fffffffffffffffll;
    for (int64_t n = (int64_t)blockIdx.x * 16 + slot; n < N; n += (int64_t)gridDim.x * 16) {
        double r[VT];
        const double* src = (INIT ? X : R) + n * VMAX + c;
#pragma unroll
        for (int t = 0; t < VT; ++t) r[t] = (c + 16 * t < V) ? src[16 * t] : 0.0;
        double acc = 0.0;
        if (INIT) {
#pragma unroll
            for (int t = 0; t < VT; ++t) acc += r[t];
        } else {
#pragma unroll
            for (int t = 0; t < VT; ++t) acc = __builtin_fma(u[t], r[t], acc);
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 16);
        if (INIT) {
#pragma unroll
            for (int t = 0; t < VT; ++t) r[t] = r[t] / acc;               // data_mat.T / data_mat.T.sum(axis=0)
        } else {
#pragma unroll
            for (int t = 0; t < VT; ++t) r[t] = r[t] - (u[t] * acc) / un;  // R - np.outer(u, u @ R) / norms[j]
        }
        double nrm = 0.0;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            R[n * VMAX + c + 16 * t] = r[t];
            nrm = __builtin_fma(r[t], r[t], nrm);
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) nrm += __shfl_xor(nrm, m, 16);
        if (nrm > best) {  // (n increases: a later equal norm does not replace an earlier one)
            best = nrm;
            besti = n;
        }
    }
    if (c == 0) {
        bval[slot] = best;
        bidx[slot] = besti;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = bval[0];
        long long bi = bidx[0];
        for (int i = 1; i < 16; ++i)
            if (bval[i] > b || (bval[i] == b && bidx[i] < bi)) {
                b = bval[i];
                bi = bidx[i];
            }
        pval[blockIdx.x] = b;
        pidx[blockIdx.x] = bi;
    }
}

// the same pass for more than 96 features: a sample's row of R is ldr = 96 * (number of feature blocks) doubles, X is read
// block by block ([b][Np][96]); the row is walked twice (dot product or sum, then deflation + norm) instead of held in
// registers.  state = u[ldr] | |u|^2
template <bool INIT>
__global__ void __launch_bounds__(SEP_BLOCK) sep_pass_wide_kernel(const double* __restrict__ X, double* __restrict__ R, int64_t N, int64_t Np, int V,
                                                                   int ldr, const double* __restrict__ state, double* __restrict__ pval,
                                                                   long long* __restrict__ pidx) {
    __shared__ double bval[16];
    __shared__ long long bidx[16];
    const int slot = threadIdx.x >> 4, c = threadIdx.x & 15;
    const double un = INIT ? 1.0 : state[ldr];
    double best = -1.0;
    long long besti = 0x7fffffffffffffffll;
    for (int64_t n = (int64_t)blockIdx.x * 16 + slot; n < N; n += (int64_t)gridDim.x * 16) {
        double* row = R + n * ldr;
        double acc = 0.0;
        for (int v = c; v < V; v += 16) {
            if (INIT) acc += X[((int64_t)(v / VMAX) * Np + n) * VMAX + v % VMAX];
            else acc = __builtin_fma(state[v], row[v], acc);
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 16);
        double nrm = 0.0;
        for (int v = c; v < ldr; v += 16) {
            double r = 0.0;
            if (v < V) r = INIT ? X[((int64_t)(v / VMAX) * Np + n) * VMAX + v % VMAX] / acc : row[v] - (state[v] * acc) / un;
            row[v] = r;
            nrm = __builtin_fma(r, r, nrm);
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) nrm += __shfl_xor(nrm, m, 16);
        if (nrm > best) {
            best = nrm;
            besti = n;
        }
    }
    if (c == 0) {
        bval[slot] = best;
        bidx[slot] = besti;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = bval[0];
        long long bi = bidx[0];
        for (int i = 1; i < 16; ++i)
            if (bval[i] > b || (bval[i] == b && bidx[i] < bi)) {
                b = bval[i];
                bi = bidx[i];
            }
        pval[blockIdx.x] = b;
        pidx[blockIdx.x] = bi;
    }
}

// one workgroup: the global argmax of the per-workgroup candidates (lowest index on ties), u <- R_j, |u|^2, chosen[round] <- j
__global__ void __launch_bounds__(256) sep_select_kernel(const double* __restrict__ R, const double* __restrict__ pval,
                                                         const long long* __restrict__ pidx, int nparts, double* __restrict__ state,
                                                         long long* __restrict__ chosen, double* __restrict__ norms, int round, int ldr) {
    __shared__ double bval[256];
    __shared__ long long bidx[256];
    double b = -1.0;
    long long bi = 0x7fffffffffffffffll;
    for (int i = threadIdx.x; i < nparts; i += 256)
        if (pval[i] > b || (pval[i] == b && pidx[i] < bi)) {
            b = pval[i];
            bi = pidx[i];
        }
    bval[threadIdx.x] = b;
    bidx[threadIdx.x] = bi;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            const double o = bval[threadIdx.x + h];
            const long long oi = bidx[threadIdx.x + h];
            if (o > bval[threadIdx.x] || (o == bval[threadIdx.x] && oi < bidx[threadIdx.x])) {
                bval[threadIdx.x] = o;
                bidx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const long long j = bidx[0];
    for (int t = threadIdx.x; t < ldr; t += 256) state[t] = R[j * ldr + t];
    if (threadIdx.x == 0) {
        state[ldr] = bval[0];
        chosen[round] = j;
        norms[round] = bval[0];  // the winning squared norm: the caller checks that it is not at rounding level
    }
}

}  // namespace salnmf
