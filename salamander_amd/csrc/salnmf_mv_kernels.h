// Small dense (K x K, K <= 64) algebra of the MvNMF W step, one workgroup, fp64, all in LDS.
// Reference arithmetic: src/salamander/models/mvnmf.py:19-24 (volume_logdet), :37-66
// (update_W_unconstrained), :80-81,86-88 (normalize + clip of a line-search trial).
// S = W W^T + delta I is symmetric positive definite (delta > 0), so the reference's
// LU-based inv/det are replaced by a Cholesky factorisation (same values to rounding).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "salnmf_kernels.h"

namespace salnmf {

constexpr int MV_BLOCK = 1024;  // one workgroup, four waves per SIMD: the parallel phases are fp64-issue bound
constexpr int MV_KMAX = 64;
constexpr int MV_LD = MV_KMAX + 1;  // padded leading dimension in LDS
constexpr int MV_VMAX = 96;
constexpr int MV_WS = 97;  // LDS row stride of W: odd, so different signature rows fall into different banks

// sum_{m<n} a[m*sa] * b[m*sb] over LDS operands, reads issued in independent batches of 8 so that one
// LDS latency is paid per batch instead of per element; fixed summation order
__device__ __forceinline__ double mv_dot(const double* a, int sa, const double* b, int sb, int n) {
    double s = 0.0;
    int m = 0;
    for (; m + 8 <= n; m += 8) {
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = a[(m + u) * sa]; y[u] = b[(m + u) * sb]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u] * y[u];
    }
    if (m < n) {  // last partial batch: clamped (in-bounds) reads, contributions masked by selects
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int mm = (m + u < n) ? m + u : n - 1;
            x[u] = a[mm * sa];
            y[u] = b[mm * sb];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (m + u < n) ? x[u] * y[u] : 0.0;
    }
    return s;
}

// S (LDS, [K][MV_LD]) <- Wl Wl^T + delta I, with Wl (LDS, [K][MV_WS]) rows = signatures
__device__ inline void mv_gram(const double* Wl, double* S, int K, int V, double delta) {
    // symmetric: each pair a <= b is computed once and mirrored (the phase is LDS-bandwidth bound)
    for (int idx = threadIdx.x; idx < K * K; idx += MV_BLOCK) {
        int a = idx / K, b = idx - a * K;
        if (a <= b) {
            double s = mv_dot(Wl + a * MV_WS, 1, Wl + b * MV_WS, 1, V);
            if (a == b) s += delta;
            S[a * MV_LD + b] = s;
            S[b * MV_LD + a] = s;
        }
    }
    __syncthreads();
}

// in-place lower Cholesky of S (K <= 64); returns log det = 2 sum log L_ii (same value in every
// thread).  Left-looking, done by ONE wave in lockstep (lane i owns row i): at column j every lane
// forms S[i][j] - sum_{m<j} L[i][m] L[j][m] from finished columns, lane j's value gives the pivot.
// No workgroup barriers inside the column loop; LDS operations of one wave execute in order.
__device__ inline double mv_cholesky_logdet(double* S, int K) {
    __shared__ double ld_shared;
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        const int row = (i < K ? i : K - 1) * MV_LD;  // idle lanes shadow the last row (results discarded)
        for (int j = 0; j < K; ++j) {
            const double dot = mv_dot(S + row, 1, S + j * MV_LD, 1, j);
            const double v = S[row + j] - dot;
            const double piv = sqrt(__shfl(v, j, 64));
            if (i == j) S[row + j] = piv;
            else if (i > j && i < K) S[row + j] = v / piv;
            __builtin_amdgcn_wave_barrier();
        }
        double l = (i < K) ? log(S[i * MV_LD + i]) : 0.0;
        // fixed-order sum over the lanes
        double tot = 0.0;
        for (int j = 0; j < K; ++j) tot += __shfl(l, j, 64);
        if (i == 0) ld_shared = 2.0 * tot;
    }
    __syncthreads();
    return ld_shared;
}

// compact W[K][V] (global) -> Wl[K][MV_WS] (LDS); all loads of a thread are issued before any use
__device__ inline void mv_load_W(const double* __restrict__ W, double* Wl, int K, int V) {
    constexpr int PT = (MV_KMAX * MV_VMAX + MV_BLOCK - 1) / MV_BLOCK;
    double w[PT];
    const int total = K * V;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int idx = threadIdx.x + MV_BLOCK * j;
        w[j] = idx < total ? W[idx] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int idx = threadIdx.x + MV_BLOCK * j;
        if (idx < total) {
            int k = idx / V, v = idx - k * V;
            Wl[k * MV_WS + v] = w[j];
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(MV_BLOCK) mv_logdet_kernel(const double* __restrict__ W, int K, int V, double delta,
                                                             double* __restrict__ out) {
    __shared__ double Wl[MV_KMAX * MV_WS];
    __shared__ double S[MV_KMAX * MV_LD];
    mv_load_W(W, Wl, K, V);
    mv_gram(Wl, S, K, V, delta);
    double ld = mv_cholesky_logdet(S, K);
    if (threadIdx.x == 0) *out = ld;
}

// ---- MvNMF W step, split so that everything that depends on W alone can run on a second stream while the
// passes over the samples run (salnmf.hip: mv_update_W_impl)

// W-only half of update_W_unconstrained: A = W @ Y_minus, B = W @ |Y| with Y = (W W^T + delta I)^-1
// (mvnmf.py:48-54, in the K x V layout), and log det(W W^T + delta I) (mvnmf.py:19-24)
__global__ void __launch_bounds__(MV_BLOCK)
    mv_prepare_W_kernel(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ Aout,
                        double* __restrict__ Bout, double* __restrict__ logdet_out) {
    __shared__ double Wl[MV_KMAX * MV_WS];
    __shared__ double S[MV_KMAX * MV_LD];   // Gram -> Cholesky factor L -> Y = S^-1
    __shared__ double Li[MV_KMAX * MV_LD];  // L^-1
    mv_load_W(W, Wl, K, V);
    mv_gram(Wl, S, K, V, delta);
    double ld = mv_cholesky_logdet(S, K);
    if (threadIdx.x == 0) *logdet_out = ld;
    for (int c = threadIdx.x; c < K; c += MV_BLOCK) {
        for (int i = 0; i < c; ++i) Li[i * MV_LD + c] = 0.0;
        Li[c * MV_LD + c] = 1.0 / S[c * MV_LD + c];
        for (int i = c + 1; i < K; ++i) {
            const double s = mv_dot(S + i * MV_LD + c, 1, Li + c * MV_LD + c, MV_LD, i - c);
            Li[i * MV_LD + c] = -s / S[i * MV_LD + i];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < K * K; idx += MV_BLOCK) {
        int a = idx / K, b = idx - a * K;
        int m0 = a > b ? a : b;
        S[a * MV_LD + b] = mv_dot(Li + m0 * MV_LD + a, MV_LD, Li + m0 * MV_LD + b, MV_LD, K - m0);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < K * V; idx += MV_BLOCK) {
        int k = idx / V, v = idx - k * V;
        double A = 0.0, B = 0.0;
        int m = 0;
        for (; m + 8 <= K; m += 8) {
            double y[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                y[u] = S[(m + u) * MV_LD + k];
                wv[u] = Wl[(m + u) * MV_WS + v];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                A += wv[u] * fmax(0.0, -y[u]);
                B += wv[u] * fabs(y[u]);
            }
        }
        for (; m < K; ++m) {
            const double y = S[m * MV_LD + k], wv = Wl[m * MV_WS + v];
            A += wv * fmax(0.0, -y);
            B += wv * fabs(y);
        }
        Aout[idx] = A;
        Bout[idx] = B;
    }
}

// closed-form root per entry (mvnmf.py:55-65) from A, B, the reduced G, the row sums of H and W; f0 (mvnmf.py:79)
struct MvRootParams {
    const double *A, *B, *G, *hsum, *kl, *logdet;
    double* f0_out;
    double lam;
    int n_given;
};

// the part of a line-search trial the forward pass needs (mvnmf.py:80-81, 85-88): blend, row sums, normalise, clip.
// ROOT: the first trial of an update, which also evaluates the closed-form root from MvRootParams and
// stores it as Wunc -- one kernel boundary less on the serial path of every MvNMF step.
template <bool ROOT>
__global__ void __launch_bounds__(MV_BLOCK)
    mv_trial_light_kernel(const double* __restrict__ W, double* __restrict__ Wunc, double gamma, int blend, int K, int V,
                          double* __restrict__ Wtrial, double* __restrict__ cs, MvRootParams r) {
    __shared__ double Wl[MV_KMAX * MV_WS];
    __shared__ double rs[MV_KMAX];
    constexpr int PT = (MV_KMAX * MV_VMAX + MV_BLOCK - 1) / MV_BLOCK;
    double a[PT], b[PT];
    const int total = K * V;
    if constexpr (ROOT) {
        if (threadIdx.x == 0) *r.f0_out = *r.kl + r.lam * *r.logdet;
        double wa[PT], wb[PT], wg[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            int idx = threadIdx.x + MV_BLOCK * j;
            const bool in = idx < total;
            b[j] = in ? W[idx] : 1.0;
            wa[j] = in ? r.A[idx] : 0.0;
            wb[j] = in ? r.B[idx] : 1.0;
            wg[j] = in ? r.G[idx] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            int idx = threadIdx.x + MV_BLOCK * j;
            if (idx < total) {
                const int k = idx / V;
                const double bb = r.hsum[k] - 4.0 * r.lam * wa[j];
                const double root = sqrt(bb * bb + 8.0 * r.lam * wb[j] * wg[j]);
                const double wu = b[j] * (root - bb) / (4.0 * r.lam * wb[j]);
                a[j] = (k < r.n_given) ? b[j] : clip_lo(wu, kEps);
                Wunc[idx] = a[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            int idx = threadIdx.x + MV_BLOCK * j;
            a[j] = idx < total ? Wunc[idx] : 0.0;
            b[j] = (blend && idx < total) ? W[idx] : 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int idx = threadIdx.x + MV_BLOCK * j;
        if (idx < total) {
            int k = idx / V, v = idx - k * V;
            double wt = a[j];
            if (!ROOT && blend) wt = (1 - gamma) * b[j] + gamma * wt;
            Wl[k * MV_WS + v] = wt;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += MV_BLOCK) {
        double s = 0.0;
#pragma unroll 8
        for (int v = 0; v < V; ++v) s += Wl[k * MV_WS + v];
        rs[k] = s;
        cs[k] = s;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < K * V; idx += MV_BLOCK) {
        int k = idx / V, v = idx - k * V;
        Wtrial[idx] = clip_lo(Wl[k * MV_WS + v] / rs[k], kEps);
    }
}

}  // namespace salnmf
