// The W tail (tail_kernel) and the small plain kernels of the engine: reductions, padding / unpadding of the host layouts,
// the l-half penalty, the W update of engines with feature blocks.  Helpers: salnmf_kernels.h (which includes this file).
#pragma once
#include "salnmf_kernels.h"

namespace salnmf {

// ----------------------------------------------------------------------------------------------
// W tail (_utils_klnmf.py:338-341 / :208-215): one workgroup per signature k.
//   stage 1 (nslabs > 0): G[k][:] = sum over the per-workgroup slabs, fixed order
//   stage 2 (do_tail)   : W' = W*G ; W' /= sum_v W' ; keep given rows ; clip
// One entry of update_W_unconstrained (mvnmf.py:55-65): the closed-form root from A = W Y_minus, B = W |Y|, the numerator
// G and rowsums_H, with given rows kept and the others clipped.  One definition for the two kernels that evaluate it
// (tail_kernel's root, mv_trial_light_kernel<true>), so that both produce the same bits.
// No fused multiply-adds here: which products hipcc contracts depends on the code around the inlined body, and the
// reference (NumPy) rounds every product and sum.
__device__ __forceinline__ double mv_root_entry(double w, double wa, double wb, double wg, double hsum, double lam, bool given) {
#pragma clang fp contract(off)
    const double bb = hsum - 4.0 * lam * wa;
    const double root = sqrt(bb * bb + 8.0 * lam * wb * wg);
    const double wu = w * (root - bb) / (4.0 * lam * wb);
    return given ? w : clip_lo(wu, kEps);
}

struct TailParams {
    const double* __restrict__ Gpart;  // [nslabs][K][VMAX]
    double* __restrict__ G;            // [K][V]
    const double* W;                   // [K][V] in
    double* Wout;                      // [K][V] out (normally == W)
    int nslabs;
    int V;
    int K;
    int n_given;
    int clip_mode;
    int do_tail;
    // optional (MvNMF): the same launch also reduces the per-workgroup row sums of H and KL partials, in the
    // summation order of sum_partials_kernel
    const double* __restrict__ hsum_part;  // [nparts][K] or null
    double* __restrict__ hsum_out;         // [K]
    const double* __restrict__ kl_part;    // [nparts] or null: partials of the KL divergence (tile_kl: x-only constants included)
    double* __restrict__ kl_out;           // [1]
    // optional (MvNMF, steps queued ahead of the host: salnmf_host_mv.h, mv_steps_queued): the line-search decision of the
    // PREVIOUS step on the device.  mv_flag: device word, non-zero = a trial was rejected, everything queued behind it
    // returns at once.  dec_f0 != null: this launch first decides the previous step's first trial -- f1 = dec_kl +
    // dec_lam * dec_logdet against dec_f0 (mvnmf.py:84), the host's expression operation for operation -- and on rejection
    // stores dec_code to the flag and returns; the host resolves that step on the classic path.
    unsigned* mv_flag;
    const double* dec_f0;
    const double* dec_kl;
    const double* dec_logdet;
    double dec_lam;
    unsigned dec_code;
    int kl_extra;  // the grid has one workgroup more than rows: it only reduces kl_part into kl_out -- the
                   // objective folded into a joint step (fused_kernel<.., true, true, true>); hsum_part is null then
    int nparts;    // KL partials (workgroups of the numerator pass)
    int nparts_h;  // row-sum partials (workgroups of the preceding update_H pass)
    // optional (MvNMF inside mv_step, unsharded; with hsum_part / kl_part): the first line-search trial in the same launch.
    // Workgroup k holds everything row k of update_W_unconstrained needs once its sums are reduced (mvnmf.py:55-65:
    // closed-form root from A, B, G[k], rowsums_H[k]; :80-81: normalise, clip; column sum for H), so the separate
    // one-workgroup kernel (mv_trial_light_kernel<true>, 7.4 us + a boundary per step) is not launched.
    const double* rootA;      // [K][V] W Y_minus; null = no root here
    const double* rootB;      // [K][V] W |Y|
    const double* rootLogdet; // [1] log det(W W^T + delta I) of the current W
    double* rootF0;           // [1] f0 = KL + lam * log det  (mvnmf.py:79)
    double* rootWunc;         // [K][V] W_unconstrained
    double* rootWtrial;       // [K][V] normalised, clipped trial
    double* rootCs;           // [KP] column sums of W_unconstrained (the factor H is rescaled by)
    double rootLam;
};

#ifndef SALNMF_TEMPLATES_ONLY  // the plain kernels below are compiled by salnmf.hip only (salnmf_launch.h)
__global__ void __launch_bounds__(TAIL_BLOCK) tail_kernel(TailParams p) {
    __shared__ TailScratch S;
    __shared__ double mvsh[2];  // (MvNMF) this row's reduced rowsums_H entry, and the KL divergence (workgroup 0)
    const int k = blockIdx.x;
    const int K = p.K;
    if (p.mv_flag != nullptr) {  // (uniform over the grid)
        __shared__ int mv_exit;
        if (threadIdx.x == 0) {
            int ex = __hip_atomic_load((gsync_t*)p.mv_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (!ex && p.dec_f0 != nullptr) {
                // every workgroup evaluates the same three doubles: the same decision everywhere
                const double f1 = __dadd_rn(p.dec_kl[0], __dmul_rn(p.dec_lam, p.dec_logdet[0]));
                if (f1 > p.dec_f0[0]) {
                    ex = 1;
                    if (k == 0) __hip_atomic_store((gsync_t*)p.mv_flag, p.dec_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            mv_exit = ex;
        }
        __syncthreads();
        if (mv_exit) return;
    }
    if (p.kl_extra && k == K) {  // (uniform over the workgroup) summation order of sum_partials_kernel
        __shared__ double kred[256];
        double s = 0.0;
        if (threadIdx.x < 256) {
            for (int i = threadIdx.x; i < p.nparts; i += 256) s += p.kl_part[i];
            kred[threadIdx.x] = s;
        }
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) kred[threadIdx.x] += kred[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) p.kl_out[0] = kred[0];
        return;
    }
    if (p.hsum_part) {  // uniform over the grid
        __shared__ double hred[256];
        for (int which = 0; which < ((k == 0 && p.kl_part) ? 2 : 1); ++which) {
            const double* part = which == 0 ? p.hsum_part + k : p.kl_part;
            const int stride = which == 0 ? K : 1;
            const int nparts = which == 0 ? p.nparts_h : p.nparts;
            double s = 0.0;
            if (threadIdx.x < 256)
                for (int i = threadIdx.x; i < nparts; i += 256) s += part[(int64_t)i * stride];
            if (threadIdx.x < 256) hred[threadIdx.x] = s;
            __syncthreads();
            for (int h = 128; h > 0; h >>= 1) {
                if ((int)threadIdx.x < h) hred[threadIdx.x] += hred[threadIdx.x + h];
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                const double value = hred[0];
                (which == 0 ? p.hsum_out[k] : p.kl_out[0]) = value;
                mvsh[which] = value;  // (for the root below: through LDS, not back through global memory)
            }
            __syncthreads();
        }
    }
    tail_row<TAIL_BLOCK, false>(S, threadIdx.x, k, p.Gpart, p.nslabs, p.G, p.W, p.Wout, p.V, K, p.n_given, p.clip_mode, p.do_tail != 0);
    if (p.rootA) {  // (uniform over the grid; requires hsum_part, kl_part and nslabs > 0)
        // tail_row left G[k][:] in S.red[0] behind a barrier; hsum_out[k] (and kl_out by workgroup 0) were stored by
        // thread 0 of this workgroup above
        const int tid = threadIdx.x, V = p.V;
        __syncthreads();
        const double hs = mvsh[0];
        if (k == 0 && tid == 0) p.rootF0[0] = mvsh[1] + p.rootLam * p.rootLogdet[0];
        double a = 0.0;
        if (tid < V) {
            a = mv_root_entry(p.W[k * V + tid], p.rootA[k * V + tid], p.rootB[k * V + tid], S.red[0][tid], hs, p.rootLam, k < p.n_given);
            p.rootWunc[k * V + tid] = a;
        }
        if (tid < VMAX) S.wn[tid] = a;  // (0 beyond V)
        __syncthreads();
        // row sum in tail_row's fixed two-level order
        if (tid < VMAX / 8) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += S.wn[8 * tid + i];
            S.red[1][tid] = t;
        }
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < VMAX / 8; ++i) t += S.red[1][i];
            S.rowsum = t;
            p.rootCs[k] = t;
        }
        __syncthreads();
        if (tid < V) p.rootWtrial[k * V + tid] = clip_lo(a / S.rowsum, kEps);
    }
}

// out[j] = sum_i part[i*stride + j], j < width: one workgroup per output, fixed summation order
// (thread t adds rows t, t+256, ... in order; then a fixed binary tree over the 256 threads)
__global__ void __launch_bounds__(256) sum_partials_kernel(const double* __restrict__ part, int n, int stride, int width,
                                                           double* __restrict__ out, const double* __restrict__ addend = nullptr) {
    __shared__ double red[256];
    const int j = blockIdx.x;
    if (j >= width) return;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[(int64_t)i * stride + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = addend ? red[0] + addend[0] : red[0];
}

// one double, device -> (pinned) host, as a kernel: its completion signal carries the event the reader waits for
__global__ void copy_scalar_kernel(double* __restrict__ dst, const double* __restrict__ src) { *dst = *src; }
// out = a + lam * b (the penalised objective from its two parts, mvnmf.py:27-34)
__global__ void combine_scalar_kernel(double* __restrict__ out, const double* __restrict__ a, double lam, const double* __restrict__ b) {
    *out = *a + lam * *b;
}

// c[n][l] = sum over the features v = l mod 16 of sample n of (x log x - x) (0 where x == 0): the x-only part of the KL
// terms that lane column l of the accumulator layout holds (tile_kl), once per upload of X.  Library log: any x the
// reference accepts.  X is [Np][ldx], pad rows are 0.
__global__ void __launch_bounds__(256) xlogx_lane_kernel(const double* __restrict__ X, int64_t Np, int V, int ldx, double* __restrict__ c) {
    const int64_t n = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    if (n >= Np) return;
    double s = 0.0;
    for (int v = l; v < V; v += 16) s += kl_term_x(X[n * ldx + v]);
    c[n * 16 + l] = s;
}

// out[k] = sum over rows n < N of H[n][k] (padded layout, leading dimension ldh): one workgroup per
// column, fixed order (thread t adds rows t, t+256, ...; then a fixed binary tree)
__global__ void __launch_bounds__(256) colsum_kernel(const double* __restrict__ H, int64_t N, int ldh, double* __restrict__ out) {
    __shared__ double red[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += 256) s += H[n * ldh + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = red[0];
}

// H <- clip(H * scale[k]) on the padded layout (normalize_WH + clip of an accepted MvNMF trial,
// mvnmf.py:80-81); scale has ldh entries, filler 1
__global__ void scale_H_kernel(double* __restrict__ H, const double* __restrict__ scale, int64_t total, int ldh) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) H[i] = clip_lo(H[i] * scale[i % ldh], kEps);
}

// compact [rows][cols] -> padded [prows][ld] (clip_lo > 0 clips the copied entries from below)
__global__ void pad_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t rows, int cols,
                           int64_t prows, int ld, double fill_cols, double fill_rows, double clip_lo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < prows * ld; i += stride) {
        int64_t r = i / ld;
        int c = (int)(i - r * ld);
        double v;
        if (r >= rows) v = fill_rows;
        else if (c >= cols) v = fill_cols;
        else {
            v = src[r * cols + c];
            if (clip_lo > 0.0) v = v < clip_lo ? clip_lo : v;
        }
        dst[i] = v;
    }
}

// A block of `rows` compact rows of element type T (row length cols) -> rows [0, rows) of a padded double matrix with
// leading dimension ld: converts, clips from below (clip_lo > 0) and fills the pad columns.  The ingest pipeline
// runs it per staged chunk (salnmf.hip: upload_rows_staged), so integer count matrices are converted on the device.
template <typename T>
__global__ void pad_rows_kernel(double* __restrict__ dst, const T* __restrict__ src, int64_t rows, int cols, int ld, double fill_cols,
                                double clip_lo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < rows * ld; i += stride) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        double v = fill_cols;
        if (c < cols) {
            v = (double)src[r * cols + c];
            if (clip_lo > 0.0) v = v < clip_lo ? clip_lo : v;
        }
        dst[i] = v;
    }
}
// The same for a matrix wider than one block -- X with n_features > 96 (blocks of bw = 96 features), H with n_signatures >
// 64 (chunks of bw = 64 signatures): the rows are scattered into nb blocks of bw columns each, dst[b][r][c] =
// src[r][bw b + c] for c < bw (0 beyond cols and in the pad columns bw <= c < ldb of a block); ldb = row stride inside a block,
// block_stride = doubles between consecutive blocks.
template <typename T>
__global__ void pad_rows_blocked_kernel(double* __restrict__ dst, const T* __restrict__ src, int64_t rows, int cols, int nb,
                                        int64_t block_stride, double clip_lo, int bw, int ldb) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t per_row = (int64_t)nb * ldb;
    for (; i < rows * per_row; i += stride) {
        const int64_t r = i / per_row;
        const int rc = (int)(i - r * per_row), b = rc / ldb, c = rc - b * ldb;
        const int col = bw * b + c;
        double v = 0.0;
        if (c < bw && col < cols) {
            v = (double)src[r * cols + col];
            if (clip_lo > 0.0) v = v < clip_lo ? clip_lo : v;
        }
        dst[(int64_t)b * block_stride + r * ldb + c] = v;
    }
}
// blocks of bw columns at row stride ldb, [nb][.][ldb] -> compact [rows][cols]
__global__ void unpad_blocked_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t rows, int cols, int bw, int ldb,
                                     int64_t block_stride) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < rows * cols; i += stride) {
        const int64_t r = i / cols;
        const int col = (int)(i - r * cols), b = col / bw;
        dst[i] = src[(int64_t)b * block_stride + r * ldb + (col - b * bw)];
    }
}
// l-half penalty of one signature chunk (klnmf.py:75-79): part[workgroup] = sum_n w_n sum_{k < K} sqrt(H[n][k]), fixed order
__global__ void __launch_bounds__(256) lhalf_penalty_kernel(const double* __restrict__ H, const double* __restrict__ wlh, int64_t N, int K, int ld,
                                                            double* __restrict__ part) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * ld; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / ld;
        const int k = (int)(i - n * ld);
        if (k < K) s += wlh[n] * sqrt(H[i]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// W tail of a problem with more than 96 features (_utils_klnmf.py:338-341 / :208-215): one workgroup per signature row.
// Gblk holds the reduced numerator of every feature block, compact [K][vb] per block at stride K * 96; the row's
// products W * G are summed in a fixed order (thread t: features t, t + 256, ...; then a binary tree), then normalise,
// keep given rows, clip.  G (full [K][V]) is left behind as the engine's reduced numerator.
__global__ void __launch_bounds__(256) w_finish_blocked_kernel(const double* __restrict__ Gblk, double* __restrict__ G, const double* __restrict__ W,
                                                               double* __restrict__ Wout, int V, int K, int n_given, int clip_mode) {
    __shared__ double red[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    double part = 0.0;
    for (int v = tid; v < V; v += 256) {
        const int b = v / VMAX, vv = v - b * VMAX;
        const int vb = V - VMAX * b < VMAX ? V - VMAX * b : VMAX;
        const double g = Gblk[(int64_t)b * K * VMAX + k * vb + vv];
        G[(int64_t)k * V + v] = g;
        part += W[(int64_t)k * V + v] * g;
    }
    red[tid] = part;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    const double rowsum = red[0];
    for (int v = tid; v < V; v += 256) {
        const double wold = W[(int64_t)k * V + v];
        double w = (wold * G[(int64_t)k * V + v]) / rowsum;
        if (k < n_given) {
            w = wold;
            if (clip_mode == 0) w = clip_lo(w, kEps);
        } else {
            w = clip_lo(w, kEps);
        }
        Wout[(int64_t)k * V + v] = w;
    }
}

// rows [r0, r1) of a padded matrix <- fill
__global__ void fill_rows_kernel(double* __restrict__ dst, int64_t r0, int64_t r1, int ld, int cols, double fill_rows, double fill_cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + r0 * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < r1 * ld; i += stride) dst[i] = ((int)(i % ld) < cols) ? fill_rows : fill_cols;
}

// padded [.][ld] -> compact [rows][cols]
__global__ void unpad_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t rows, int cols, int ld) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < rows * cols; i += stride) {
        int64_t r = i / cols;
        int c = (int)(i - r * cols);
        dst[i] = src[r * ld + c];
    }
}

#endif  // SALNMF_TEMPLATES_ONLY

}  // namespace salnmf
