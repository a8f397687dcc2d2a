// The forward pass of the KL-NMF path (forward_kernel): W @ H with the objective / per-sample divergences / reconstruction /
// Poisson log-likelihood / ratio epilogues.  Layout and helpers: salnmf_kernels.h (which includes this file).
#pragma once
#include "salnmf_kernels.h"

namespace salnmf {

// ----------------------------------------------------------------------------------------------
// Forward pass + objective: the "W@H step" of the north star.  P = Ht . W on MFMA, then
//   mode 0: weighted KL partial per workgroup (+ l-half penalty)        klnmf.py:64-80
//   mode 1: per-sample KL, zeros replaced by EPS in X and WH            _utils_klnmf.py:58-97
//   mode 2: the reconstruction H @ W                                    signature_nmf.py:221-224
//   mode 3: Poisson log-likelihood partial (CorrNMF ELBO), no factorial _utils_klnmf.py:98-133
//   mode 4: the ratio X / (H @ W) of the update rules (the fused passes' division)   _utils_klnmf.py:333
// PIN (n_signatures > 64, one launch per chunk of <= 64 signatures): P = pin + H_c @ W_c, so that a chain of launches
// accumulates the product over the chunks (mode 2) and the last one evaluates what it is needed for (modes 0, 1, 4).
struct FwdParams {
    const double* __restrict__ X;       // [Np][VMAX]
    const double* __restrict__ H;       // [Np][KP]
    const double* __restrict__ W;       // [K][V]
    const double* __restrict__ wkl;     // [Np] or null
    const double* __restrict__ wlh;     // [Np] or null
    const double* __restrict__ hscale;  // [KP] or null: H read as clip(H*hscale)
    const double* __restrict__ xlx;     // [Np][16] mode 0: x-only constants of the KL terms per (sample, lane column) (xlogx_lane_kernel)
    const double* pin;                  // (PIN instantiations) [Np][VMAX] or null: P starts from this instead of 0 -- the product of
                                        // the signature chunks before this one (n_signatures > 64); may be `out` itself
    double* __restrict__ out;           // mode 0: [gridDim.x]; mode 1: [Np]; modes 2, 4: [Np][VMAX]
    int64_t N;
    int V;
    int ldw;                            // row stride of W (= V unless W points at one feature block of a wider matrix)
    int K;
    int64_t ntiles;
    // mode 0, optional: the final sum inside the launch (the workgroup that finishes last adds the partials in the order of
    // sum_partials_kernel -- the same bits -- plus sum_addend[0], and stores the objective): no reduction kernel behind it
    double* sum_out;            // [1] or null (device or pinned host memory)
    const double* sum_addend;   // [1] or null
    unsigned* sum_counter;      // arrival counter, zero between launches
};

// Two workgroups per CU (two waves per SIMD): the objective terms are VALU-heavy and the loads
// are not software-pipelined here, so the second wave hides the first one's memory latency.
// Only the 4*KS rows of W that the contraction touches are staged, which keeps LDS <= 80 KB.
template <int KS>
constexpr int fwd_lds_doubles() { return 4 * KS * WS + WAVES * Geo<KS>::HL + BLOCK + Geo<KS>::KP + LOGTAB_DOUBLES; }

template <int KS, int MODE, bool PIN = false>
__global__ void __launch_bounds__(BLOCK, (fwd_lds_doubles<KS>() * 8 <= 80 * 1024 ? 2 : 1)) forward_kernel(FwdParams p) {
    using G_ = Geo<KS>;
    constexpr int KP = G_::KP, LS = G_::LS, HV = G_::HV;
    constexpr int FROWS = 4 * KS;  // rows of W read by the P product
    __shared__ __attribute__((aligned(16))) double lds[fwd_lds_doubles<KS>()];  // <= 80 KB (two per CU) up to KS = 13

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const int V = p.V, K = p.K;
    const int64_t N = p.N;

    double* Wl = lds;
    double* Hl = lds + FROWS * WS + wave * G_::HL;
    double* red = lds + FROWS * WS + WAVES * G_::HL;
    double* hsl = red + BLOCK;  // [KP] copy of hscale
    if (p.hscale && tid < KP) hsl[tid] = p.hscale[tid];
    // table of log_pos, in LDS.  (Measured and dropped, profiles/r04/log_table.md: the same table through the vector L1
    // -- global loads of the 4 KB __device__ array instead of ds_read_b128 -- 33.9 -> 41.3 us at c2: the lookup's latency
    // then sits in front of every batch of logarithms; a conflict-free replicated table does not fit: two workgroups of
    // this kernel leave 1.4 KB of a CU's 160 KB.)
    double* ltab = hsl + KP;
    if (MODE == 0 || MODE == 3) stage_logtab(ltab, tid);

    stage_W<FROWS>(Wl, p.W, K, V, p.ldw, tid);
    __syncthreads();

    int hrow[HV], hcol[HV];
#pragma unroll
    for (int j = 0; j < HV; ++j) {
        int e = 2 * lane + 128 * j;
        hrow[j] = e / KP;
        hcol[j] = e - hrow[j] * KP;
    }

    const int64_t tstride = (int64_t)gridDim.x * WAVES;
    double total = 0.0;

    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < p.ntiles; tile += tstride) {
        const int64_t n0 = tile * 16;
        const d2* hsrc = reinterpret_cast<const d2*>(p.H + n0 * KP) + lane;
        d2 hv[HV];
#pragma unroll
        for (int j = 0; j < HV; ++j) hv[j] = hsrc[64 * j];
        // (PIN) the product of the chunks before this one: requested with the tile's other loads -- behind the staging of H it was
        // a second memory round trip per tile
        d4 pr[VT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) pr[vt] = (d4){0, 0, 0, 0};
        if (PIN && p.pin) {
            const double* psrc = p.pin + (n0 + q) * VMAX + c16;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[vt][r] = psrc[4 * r * VMAX + 16 * vt];
        }
        double x[VT][4];
        if (MODE != 2) {
            const double* xsrc = p.X + (n0 + q) * VMAX + c16;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[vt][r] = xsrc[4 * r * VMAX + 16 * vt];
        }
        double wv[4] = {1.0, 1.0, 1.0, 1.0}, cv[4] = {0.0, 0.0, 0.0, 0.0};  // mode 0: weight and x-only constants of this lane's rows
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cv[r] = p.xlx[(n0 + q + 4 * r) * 16 + c16];
                if (p.wkl) wv[r] = p.wkl[n0 + q + 4 * r];
            }
        }
        double pen = 0.0;
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            if (p.hscale) {
                hv[j][0] = clip_lo(hv[j][0] * hsl[hcol[j]], kEps);
                hv[j][1] = clip_lo(hv[j][1] * hsl[hcol[j] + 1], kEps);
            }
            if (MODE == 0 && p.wlh) {  // l-half penalty, klnmf.py:75-79
                int64_t n = n0 + hrow[j];
                if (n < N) {
                    double w = p.wlh[n];
                    if (hcol[j] < K) pen += w * sqrt(hv[j][0]);
                    if (hcol[j] + 1 < K) pen += w * sqrt(hv[j][1]);
                }
            }
            *reinterpret_cast<d2*>(Hl + hrow[j] * LS + hcol[j]) = hv[j];
        }
        __builtin_amdgcn_wave_barrier();

        const double* ha = Hl + c16 * LS + q;
        const double* wb = Wl + q * WS + c16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            double a = ha[4 * s];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) pr[vt] = mfma(a, wb[4 * s * WS + 16 * vt], pr[vt]);
        }

        if (MODE == 0) {
            total += pen + tile_kl<true>(x, pr, wv, cv, ltab, n0, N, V, q, c16);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t n = n0 + q + 4 * r;
                double acc = 0.0;
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
                    if (n < N && 16 * vt + c16 < V) {
                        double xv = x[vt][r], pv = pr[vt][r];
                        double xe = (xv == 0.0) ? kEps : xv, pe = (xv == 0.0) ? kEps : pv;
                        double l = (log_operand_ok(xe) && log_operand_ok(pe)) ? log_ratio(xe, pe) : log(xe / pe);
                        acc += xe * l - xv + pv;
                    }
                // reduce over the 16 lanes that share this sample row (same q)
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) acc += __shfl_xor(acc, m, 64);
                if (c16 == 0) p.out[n] = acc;
            }
        } else if (MODE == 3) {
            // Poisson log-likelihood without the factorial term (_utils_klnmf.py:98-133):
            // sum over the valid entries of (P != 0 ? X log P : 0) - P
            bool ok = true;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ok &= !(n0 + q + 4 * r < N && 16 * vt + c16 < V) || log_pos_ok(pr[vt][r]);
            if (__all(ok)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool nvalid = n0 + q + 4 * r < N;
                    double ps[VT], lp[VT];
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) ps[vt] = (nvalid && 16 * vt + c16 < V) ? pr[vt][r] : 1.0;
                    log_pos_n<VT>(ps, ltab, lp);
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
                        if (nvalid && 16 * vt + c16 < V) total += x[vt][r] * lp[vt] - ps[vt];
                }
            } else {
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + q + 4 * r < N && 16 * vt + c16 < V) {
                            const double xv = x[vt][r], pv = pr[vt][r];
                            double t = 0.0;
                            if (pv != 0.0) t = xv * log(pv);
                            total += t - pv;
                        }
            }
        } else {
            double* dst = p.out + (n0 + q) * VMAX + c16;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[4 * r * VMAX + 16 * vt] = MODE == 4 ? div_path(x[vt][r], pr[vt][r]) : pr[vt][r];
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (MODE == 0 || MODE == 3) {
        red[tid] = total;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < BLOCK; ++i) s += red[i];
            if (MODE == 0 && p.sum_out != nullptr) {
                // (cdna_hip_programming.md, guideline 16, counter form: write-through partial, drained, then the ticket)
                __hip_atomic_store((gdouble*)(p.out + blockIdx.x), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned ticket = __hip_atomic_fetch_add((gsync_t*)p.sum_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                red[0] = (ticket == gridDim.x - 1u) ? 1.0 : 0.0;
            } else {
                p.out[blockIdx.x] = s;
            }
        }
        if (MODE == 0 && p.sum_out != nullptr) {
            __syncthreads();
            const bool last = red[0] != 0.0;  // (uniform)
            __syncthreads();
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the loads below the ticket)
                double sum = 0.0;
                for (int i = tid; i < (int)gridDim.x; i += BLOCK) sum += ld_shared<true>(p.out + i);
                red[tid] = sum;
                __syncthreads();
                for (int h = BLOCK / 2; h > 0; h >>= 1) {
                    if (tid < h) red[tid] += red[tid + h];
                    __syncthreads();
                }
                if (tid == 0) {
                    p.sum_out[0] = p.sum_addend ? red[0] + p.sum_addend[0] : red[0];
                    __hip_atomic_store((gsync_t*)p.sum_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

}  // namespace salnmf
