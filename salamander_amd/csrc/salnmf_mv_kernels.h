// Small dense (K x K, K <= 64) algebra of the MvNMF W step, one workgroup, fp64, all in LDS.
// Reference arithmetic: src/salamander/models/mvnmf.py:19-24 (volume_logdet), :37-66
// (update_W_unconstrained), :80-81,86-88 (normalize + clip of a line-search trial).
// S = W W^T + delta I is symmetric positive definite (delta > 0), so the reference's
// LU-based inv/det are replaced by an elimination without pivoting (salnmf_mv_device.h; same values to rounding).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "salnmf_kernels.h"
#include "salnmf_mv_device.h"

namespace salnmf {

__global__ void __launch_bounds__(MV_BLOCK) mv_logdet_kernel(const double* __restrict__ W, int K, int V, double delta,
                                                             double* __restrict__ out) {
    __shared__ double Wl[MV_KMAX * MV_WS];
    __shared__ double S[MV_KMAX * MV_LD];
    __shared__ double T[MV_KMAX * MV_LD];
    __shared__ double piv[MV_KMAX + 1];
    mv_load_W(W, Wl, K, V);
    mv_gram(Wl, S, K, V, delta);
    mv_eliminate<MV_BLOCK, false>(S, T, piv, K);  // (the pivots of mv_prepare_W_body's full elimination, bit for bit)
    const double ld = mv_logdet_from_pivots(piv, K, piv + K);
    if (threadIdx.x == 0) *out = ld;
}

__global__ void __launch_bounds__(MV_BLOCK)
    mv_prepare_W_kernel(const double* __restrict__ W, int K, int V, double delta, double* __restrict__ Aout,
                        double* __restrict__ Bout, double* __restrict__ logdet_out) {
    __shared__ double Wl[MV_KMAX * MV_WS];
    __shared__ double S[MV_KMAX * MV_LD];  // Gram matrix / elimination ping
    __shared__ double T[MV_KMAX * MV_LD];  // elimination pong
    __shared__ double piv[MV_KMAX + 1];
    mv_prepare_W_body<MV_BLOCK>(W, K, V, delta, Aout, Bout, logdet_out, Wl, S, T, piv);
}

// closed-form root per entry (mvnmf.py:55-65) from A, B, the reduced G, the row sums of H and W; f0 (mvnmf.py:79)
struct MvRootParams {
    const double *A, *B, *G, *hsum, *kl, *logdet;
    double* f0_out;
    double lam;
    int n_given;
    // optional (steps queued ahead of the host on a SAMPLE-SHARDED engine: salnmf_host_mv.h, mv_steps_queued): the device-side
    // line-search decision that tail_kernel takes on an unsharded engine (TailParams::mv_flag, dec_*), here in the kernel
    // that follows the all-reduce of [G | rowsums_H | KL | the previous trial's KL] -- the operands are the all-reduced
    // sums, the same bits on every rank, so every rank takes the same decision.
    unsigned* mv_flag;
    const double* dec_f0;
    const double* dec_kl;
    const double* dec_logdet;
    double dec_lam;
    unsigned dec_code;
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
    if (ROOT && r.mv_flag != nullptr) {  // (tail_kernel's prologue, expression for expression)
        __shared__ int mv_exit;
        if (threadIdx.x == 0) {
            int ex = __hip_atomic_load((gsync_t*)r.mv_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (!ex && r.dec_f0 != nullptr) {
                const double f1 = __dadd_rn(r.dec_kl[0], __dmul_rn(r.dec_lam, r.dec_logdet[0]));
                if (f1 > r.dec_f0[0]) {
                    ex = 1;
                    __hip_atomic_store((gsync_t*)r.mv_flag, r.dec_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            mv_exit = ex;
        }
        __syncthreads();
        if (mv_exit) return;
    }
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
                a[j] = mv_root_entry(b[j], wa[j], wb[j], wg[j], r.hsum[k], r.lam, k < r.n_given);
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
        // the row sum in the W tail's fixed two-level order (12 groups of 8 consecutive features, then the groups): the
        // root evaluated inside tail_kernel and this kernel give the same bits
        double s = 0.0;
        for (int g0 = 0; g0 < MV_VMAX; g0 += 8) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += (g0 + i < V) ? Wl[k * MV_WS + g0 + i] : 0.0;
            s += t;
        }
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
