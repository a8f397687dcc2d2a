// Device functions of the small dense (K x K, K <= 64) algebra of the MvNMF W step: one workgroup, fp64, all in LDS.
// Reference arithmetic: src/salamander/models/mvnmf.py:19-24 (volume_logdet), :37-66 (update_W_unconstrained).
// S = W W^T + delta I is symmetric positive definite (delta > 0), so the reference's LU-based inv/det are replaced by a
// Gauss-Jordan elimination without pivoting (same values to rounding).  Kernels: salnmf_mv_kernels.h; the spare workgroup of the MvNMF
// update_H pass (salnmf_kernels.h: fused_kernel) runs mv_prepare_W_body too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// S (LDS, [K][MV_LD]) <- Wl Wl^T + delta I, with Wl (LDS, [K][MV_WS]) rows = signatures.  NT = threads of the workgroup.
template <int NT = MV_BLOCK>
__device__ inline void mv_gram(const double* Wl, double* S, int K, int V, double delta) {
    // symmetric: each pair a <= b is computed once and mirrored (the phase is LDS-bandwidth bound)
    for (int idx = threadIdx.x; idx < K * K; idx += NT) {
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
template <int NT, bool FULL>
__device__ inline double* mv_eliminate(double* src, double* dst, double* piv, int K) {
    constexpr int E = (MV_KMAX * MV_KMAX + NT - 1) / NT;  // elements per thread, at most
    int off[E];      // i * MV_LD + j of this thread's elements, -1 = none
    short ei[E], ej[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = threadIdx.x + NT * e;
        const int i = idx / K, j = idx - i * K;
        ei[e] = (short)i;
        ej[e] = (short)j;
        off[e] = idx < K * K ? i * MV_LD + j : -1;
    }
    for (int k = 0; k < K; ++k) {
        const double d = src[k * MV_LD + k];
        const double rd = 1.0 / d;
        if (threadIdx.x == 0) piv[k] = d;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (off[e] < 0) continue;
            const int i = ei[e], j = ej[e];
            if (!FULL && (i <= k || j <= k)) continue;
            const double r = src[k * MV_LD + j] * rd;
            double v;
            if (i == k) {
                v = (j == k) ? rd : r;
            } else {
                const double f = src[i * MV_LD + k];
                v = (j == k) ? -f * rd : __builtin_fma(-f, r, src[off[e]]);
            }
            dst[off[e]] = v;
        }
        __syncthreads();
        double* t = src;
        src = dst;
        dst = t;
    }
    return src;
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
// passes over the samples run (salnmf.hip: mv_update_W_impl)

// W-only half of update_W_unconstrained: A = W @ Y_minus, B = W @ |Y| with Y = (W W^T + delta I)^-1
// (mvnmf.py:48-54, in the K x V layout), and log det(W W^T + delta I) (mvnmf.py:19-24).
// One workgroup of NT threads; LDS scratch: Wl [K][MV_WS], S [K][MV_LD], T [K][MV_LD], piv [K + 1].  The body is shared by
// the stand-alone kernel (1024 threads, side stream) and by the spare workgroup of the MvNMF update_H pass
// (salnmf_kernels.h: fused_kernel, 256 threads); its results do not depend on NT, so both produce the same bits.
template <int NT>
__device__ __forceinline__ void mv_prepare_W_body(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ Aout,
                                                  double* __restrict__ Bout, double* __restrict__ logdet_out, double* Wl, double* S, double* T,
                                                  double* piv) {
    mv_load_W<NT>(W, Wl, K, V);
    mv_gram<NT>(Wl, S, K, V, delta);
    S = mv_eliminate<NT, true>(S, T, piv, K);  // Y = (W W^T + delta I)^-1
    const double ld = mv_logdet_from_pivots(piv, K, piv + K);
    if (threadIdx.x == 0) *logdet_out = ld;
    for (int idx = threadIdx.x; idx < K * V; idx += NT) {
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

}  // namespace salnmf
