// CorrNMF sample embeddings, batched: SIXTEEN Newton-CG solves per wavefront in lockstep rounds, every round's
// evaluations as two fp64 MFMA products.
//
// The problems (CorrNMFDet.update_sample_embeddings, corrnmf_det.py:115-141; MultimodalCorrNMF.update_sample_embeddings,
// mmcorrnmf.py:398-428; objective / gradient / Hessian _utils_corrnmf.py:182-351, solved by
// scipy.optimize.minimize(method="Newton-CG", maxiter=3), :400-407): one per sample n over u in R^dim,
//   minimise  -[ sum_i aux_i[n] <L_i, u> - sum_i exp(alpha_mod(i)[n] + beta_i + <L_i, u>) - |u|^2 / (2 var) ]
// with the T signatures of all modalities as terms.  Every evaluation the solver asks for is two products with the
// SAME T x dim matrix L -- s = L y, then L^T w with w = exp(..+ s) (objective / gradient) or w = hw * s (Hessian-vector
// product) -- so a per-sample solve (salnmf_corr_kernels.h: one wavefront per sample) is a chain of matrix-VECTOR
// products without reuse: bound by the LDS pipe at ~5 % of the fp64 vector peak (profiles/r03/c5_roofline.md).
// Sixteen samples side by side make it a matrix-MATRIX product:
//   S = L . Y        (T x dim) . (dim x 16)     v_mfma_f64_16x16x4, A = L from LDS, B = the 16 requested vectors
//   W = phi(S)       in the accumulator registers; phi depends on what the column's solve asked for
//   R = L^T . W      (dim x T) . (T x 16)       A = L^T from LDS, B = W *as it stands in the accumulators*
// The f64 MFMA lane maps (salnmf_kernels.h) make this free of data movement: D[row = (lane>>4) + 4 reg][col = lane&15]
// is the B operand of k-step `reg` of a product that contracts over D's rows, so W feeds the second product directly,
// and a vector of the solver lives in the same layout -- lane (q = lane>>4, c = lane&15) holds the components 4 j + q,
// j = 0 .. 4 DT - 1, of sample slot c -- in which it is both the B operand of the first product (k-step j) and the
// D output of the second.  A dot product is 4 DT in-lane FMAs and a sum over the slot's four lanes (xor 16, xor 32).
//
// The solves take different paths (CG iterations, line-search trials), so the solver is the resumable form of
// salnmf_newtoncg.h (salnmf_ncg_machine.h: stops at every evaluation request); a round serves whatever each slot asked
// for, and a slot whose solve has ended stores its result and takes the wave's next sample in the same round
// (a new sample's first round forms sg = L^T aux: the same second product).  The results do not depend on the slot
// or on the neighbours: every column of an MFMA product depends on its own column of B only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "salnmf_corr_params.h"
#include "salnmf_ncg_machine.h"

namespace salnmf {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr double kEpsB = 1.1920928955078125e-07;  // float32 eps, _utils_corrnmf.py:408-409
constexpr int BT_WAVES = 4;                        // one wave per SIMD: the solver state fills most of the register file
constexpr int BT_BLOCK = 64 * BT_WAVES;

__device__ __forceinline__ d4 mfma64(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// v where the lane's mask is all ones, +0.0 where it is zero: phi's per-column choice as bit operations, so that the
// stretch from the first product to the end of the second stays ONE basic block (a ?: chain became exec-mask branches
// around the register reloads of its operands, which cut the scheduling region at every term)
__device__ __forceinline__ double masked(double v, long long mask) { return __longlong_as_double(__double_as_longlong(v) & mask); }
__device__ __forceinline__ double either(double a, double b) { return __longlong_as_double(__double_as_longlong(a) | __double_as_longlong(b)); }
// x / d with r = 1 / d (correctly rounded) at hand: quotient estimate, exact residual, one correction -- the correctly
// rounded quotient for the operands of this path (|x / d| far from the subnormal and overflow ranges) in 3 instructions
// instead of the ~15 of the IEEE sequence; NaN and inf propagate as in a division
__device__ __forceinline__ double div_by(double x, double d, double r) {
    const double q = x * r;
    const double rem = fma(-q, d, x);
    const double q1 = fma(rem, r, q);
    return (q1 == q1) ? q1 : q;  // (inf * 0 in the residual: keep the estimate, which is the division's result)
}

// A vector of one of the 16 problems, NJ elements per lane: lane (q, c) of slot c holds the components 4 j + q.
template <int NJ>
struct LaneVec {
    static constexpr int n = NJ;
    double v[n];
    // sum over the four lanes of a sample slot (lanes c, c + 16, c + 32, c + 48) without the LDS: v_permlane16_swap /
    // v_permlane32_swap of the value with itself leave each lane with its own and its partner's value; the four
    // lanes of a slot add the same pairs in the same order and get the same bits
    static __host__ __device__ __forceinline__ double reduce(double s) {
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
        lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
        a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
#endif
        return s;
    }
};

// LDS of a workgroup: the two operand images of L (every (tile, k-step) block is 64 doubles in lane order: conflict-free
// ds_read_b64 at an immediate offset), the signature scalings and the term -> (modality, signature) map.
// NE = term elements per lane (terms <= 4 NE), NJ = vector elements per lane (dim <= 4 NJ).
template <int NE, int NJ>
struct BatchedLds {
    static constexpr int TT = (NE + 3) / 4, DT = (NJ + 3) / 4;
    static constexpr int IMG1 = TT * NJ * 64;  // [mt][ks][lane]  = L[16 mt + c][4 ks + q]
    static constexpr int IMG2 = DT * NE * 64;  // [ct][ks2][lane] = L[4 ks2 + q][16 ct + c]
    static constexpr int SLOT = (NE + NJ) * 64;  // per wave: aux [e][lane] and sg [j][lane] of the slots' samples
    static constexpr size_t bytes = (size_t)(IMG1 + IMG2 + 4 * NE + 4 * SLOT) * sizeof(double) + 2 * 4 * NE * sizeof(int);
};

// PROF (development builds, SALNMF_DEV_PROFILE): per wave, shader-clock cycles spent in the sections of a round
// [0] refill [2] first product [1] phi and second product [3] - [4] solver [5] rounds [6] sum of occupied slots over the rounds
template <int NE, int NJ, bool PROF>
__global__ void __launch_bounds__(BT_BLOCK, 1) corr_sample_embeddings_batched_kernel(SampleEmbeddingParams p, int64_t per_wave, long long* prof) {
    using V = LaneVec<NJ>;
    using Lds = BatchedLds<NE, NJ>;
    constexpr int TT = Lds::TT, DT = Lds::DT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* img1 = lds;
    double* img2 = img1 + Lds::IMG1;
    double* sol = img2 + Lds::IMG2;
    double* slots = sol + 4 * NE;
    int* tmod = (int*)(slots + 4 * Lds::SLOT);
    int* tk = tmod + 4 * NE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int dim = p.dim;
    int T = 0;
    for (int mo = 0; mo < p.n_mod; ++mo) T += p.K[mo];
    for (int i = tid; i < 4 * NE; i += BT_BLOCK) {
        int mo = 0, k = i;
        while (mo < p.n_mod && k >= p.K[mo]) { k -= p.K[mo]; ++mo; }
        const bool live = mo < p.n_mod;
        tmod[i] = live ? mo : 0;
        tk[i] = live ? k : 0;
        sol[i] = live ? p.beta[mo][k] : 0.0;
    }
    __syncthreads();
    // the two operand images of L: eight unconditional loads (clamped term / component, masked afterwards) in flight per
    // round -- one guarded load per loop iteration is a memory round trip per iteration, 27 of them in front of every launch
    auto fill_image = [&](double* img, int count, auto term_comp) {
        for (int i0 = tid; i0 < count; i0 += 8 * BT_BLOCK) {
            double v[8];
            bool in[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * BT_BLOCK < count ? i0 + u * BT_BLOCK : count - 1;
                int term, comp;
                term_comp(i, term, comp);
                in[u] = term < T && comp < dim;
                const int tt = term < T ? term : T - 1, cc = comp < dim ? comp : dim - 1;
                v[u] = p.L[tmod[tt]][tk[tt] * dim + cc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * BT_BLOCK < count) img[i0 + u * BT_BLOCK] = in[u] ? v[u] : 0.0;
        }
    };
    fill_image(img1, Lds::IMG1, [&](int i, int& term, int& comp) {
        const int l = i & 63, blk = i >> 6, ks = blk % NJ, mt = blk / NJ;
        term = 16 * mt + (l & 15), comp = 4 * ks + (l >> 4);
    });
    fill_image(img2, Lds::IMG2, [&](int i, int& term, int& comp) {
        const int l = i & 63, blk = i >> 6, ks2 = blk % NE, ct = blk / NE;
        term = 4 * ks2 + (l >> 4), comp = 16 * ct + (l & 15);
    });
    __syncthreads();
    // from here on the waves run independently (no workgroup barrier below)
    const int64_t wave_id = (int64_t)blockIdx.x * BT_WAVES + wave;
    const int64_t n_end = std::min<int64_t>(p.N, (wave_id + 1) * per_wave);
    int64_t next = wave_id * per_wave;  // wave-uniform: the next sample of this wave's range without a slot
    const double variance = p.variance, inv_variance = 1.0 / p.variance;
    // first term and number of terms of every modality (uniform), for the refill's per-modality sweeps
    int t_first[CORR_MODS + 1];
    t_first[0] = 0;
#pragma unroll
    for (int mo = 0; mo < CORR_MODS; ++mo) t_first[mo + 1] = t_first[mo] + (mo < p.n_mod ? p.K[mo] : 0);
    const double* img1l = img1 + lane;
    const double* img2l = img2 + lane;

    ncgm::Machine<V> mc;
    mc.phase = ncgm::EMPTY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) mc.xk.v[j] = mc.gv.v[j] = mc.xs.v[j] = mc.ri.v[j] = mc.ps.v[j] = mc.yv.v[j] = 0.0;
    // aux of the lane's terms and sg = sum_i aux_i L[i][.] of the slot's sample live in LDS (wave-private, [element][lane]:
    // conflict-free): they are read once per point evaluation, and the registers they would take are what keeps the
    // largest instantiation out of scratch
    double* const a = slots + wave * Lds::SLOT + lane;
    double* const sg = a + NE * 64;
    double cs[NE];    // (sample scaling + signature scaling) of the lane's terms; -inf beyond T: exp(.) = 0
    double hw[NE];    // exp(cs + s) of the slot's last point evaluation = the Hessian weights at the accepted point
    int64_t n_slot = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        a[e * 64] = 0.0;
        cs[e] = hw[e] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) sg[j * 64] = 0.0;

    long long pc[7] = {0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    for (;;) {
        if (PROF) t0 = __builtin_amdgcn_s_memtime();
        // ---- free slots take the next samples of the wave's range (in slot order).  The loads are consumed after the
        // first product (aux is the B operand of a new slot's second product), so their latency runs beside its MFMAs.
        const bool is_free = mc.phase == ncgm::EMPTY;
        const unsigned free_mask = (unsigned)(__ballot(is_free) & 0xFFFFull);  // lanes 0..15 are the q = 0 lanes of the slots
        if (free_mask && next < n_end) {
            if (is_free) {
                const int64_t n = next + __popc(free_mask & ((1u << c) - 1u));
                if (n < n_end) {
                    n_slot = n;
                    // the lane's terms 4 e + q, modality by modality (uniform loop): one sample scaling and one row base
                    // per modality; a term outside the modality reads the row's first entry and keeps its value
#pragma unroll
                    for (int e = 0; e < NE; ++e) cs[e] = -INFINITY;
                    double av[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) av[e] = 0.0;
#pragma unroll
                    for (int mo = 0; mo < CORR_MODS; ++mo) {
                        if (mo < p.n_mod) {  // uniform
                            const double* row = p.aux[mo] + n * p.KP[mo];
                            const double al = p.alpha[mo][n];
                            const int first = t_first[mo], count = t_first[mo + 1] - t_first[mo];
#pragma unroll
                            for (int e = 0; e < NE; ++e) {
                                const int k = 4 * e + q - first;
                                const bool mine = (unsigned)k < (unsigned)count;
                                const double got = row[mine ? k : 0];
                                av[e] = mine ? got : av[e];
                                cs[e] = mine ? al + sol[4 * e + q] : cs[e];
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < NE; ++e) a[e * 64] = av[e];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) mc.xk.v[j] = (4 * j + q < dim) ? p.U[n * dim + 4 * j + q] : 0.0;
                    mc.phase = ncgm::WAIT_PREP;
                }
            }
            next += __popc(free_mask);  // (beyond n_end once the range is used up: harmless)
        }
        if (!__any(mc.phase != ncgm::EMPTY)) break;

        const int req = mc.request();
        const bool prep = mc.phase == ncgm::WAIT_PREP;
        const bool is_point = req == ncgm::REQ_POINT, is_hp = req == ncgm::REQ_HESSP;
        const long long m_point = is_point ? -1ll : 0ll, m_hp = is_hp ? -1ll : 0ll, m_prep = prep ? -1ll : 0ll, m_keep = ~m_point;
        if (PROF) {
            t1 = __builtin_amdgcn_s_memtime();
            pc[0] += t1 - t0;
            t0 = t1;
            pc[5] += 1;
            pc[6] += __popcll(__ballot(mc.phase != ncgm::EMPTY) & 0xFFFFull);
        }
        // ---- S = L . Y: the B operand of k-step j is element j of the requested vector.  Straight-line code from here
        // to the end of the second product (no branch: the scheduler may place phi's VALU work between the MFMAs);
        // the A operands of a k-step are read from LDS while the previous k-step's MFMAs run.
        double y[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) y[j] = is_hp ? mc.ps.v[j] : is_point ? mc.yv.v[j] : 0.0;
        d4 S[TT];
        {
            double A[2][TT];
#pragma unroll
            for (int mt = 0; mt < TT; ++mt) A[0][mt] = img1l[(mt * NJ) * 64];
#pragma unroll
            for (int ks = 0; ks < NJ; ++ks) {
                if (ks + 1 < NJ) {
#pragma unroll
                    for (int mt = 0; mt < TT; ++mt) A[(ks + 1) & 1][mt] = img1l[(mt * NJ + ks + 1) * 64];
                }
#pragma unroll
                for (int mt = 0; mt < TT; ++mt)
                    S[mt] = ks == 0 ? mfma64(A[0][mt], y[0], d4{0.0, 0.0, 0.0, 0.0}) : mfma64(A[ks & 1][mt], y[ks], S[mt]);
            }
        }
        if (PROF) {
            asm volatile("s_nop 0" ::"v"(S[0][0]), "v"(S[TT - 1][3]));  // (complete before the clock is read)
            t1 = __builtin_amdgcn_s_memtime();
            pc[2] += t1 - t0;
            t0 = t1;
        }
        // ---- W = phi(S) per column: exp(cs + s) for a point, hw * s for a Hessian-vector product, aux for a new sample;
        //      R = L^T . W: k-step e contracts over the terms 4 e + q
        double lin = 0.0, ex = 0.0;
        d4 R[DT];
        {
            // Software pipeline, pinned by sched_barrier fences: the DT MFMAs of term element e each run beside one stage
            // of phi for element e + 1 (~12-15 VALU instructions = one MFMA's 64 cycles).  Left alone -- and under
            // sched_group_barrier hints as well -- the scheduler emitted the MFMAs of two elements back to back (dependent
            // pairs on one accumulator) and phi's VALU chains alone in between: 11.8 k cycles for 7.0 k of MFMA.
            // exp(t) is OCML's algorithm instruction for instruction (same bits as the exp() of the per-sample kernel):
            // k = rint(t / ln 2), f = t - k ln2_hi - k ln2_lo, degree-11 polynomial, ldexp; cut into the stages.
            double A[2][DT];
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) A[0][ct] = img2l[(ct * NE) * 64];
            double st_t, st_k, st_f, st_p;  // the element in flight through the stages
            auto stage_a = [&](int e) {
                st_t = cs[e] + S[e >> 2][e & 3];
                st_k = __builtin_rint(st_t * 0x1.71547652b82fep+0);
                st_f = fma(st_k, -0x1.62e42fefa39efp-1, st_t);
                st_f = fma(st_k, -0x1.abc9e3b39803fp-56, st_f);
                st_p = fma(st_f, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
                st_p = fma(st_f, st_p, 0x1.71dee623fde64p-19);
                st_p = fma(st_f, st_p, 0x1.a01997c89e6b0p-16);
                st_p = fma(st_f, st_p, 0x1.a01a014761f6ep-13);
            };
            auto stage_b = [&]() {
                st_p = fma(st_f, st_p, 0x1.6c16c1852b7b0p-10);
                st_p = fma(st_f, st_p, 0x1.1111111122322p-7);
                st_p = fma(st_f, st_p, 0x1.55555555502a1p-5);
                st_p = fma(st_f, st_p, 0x1.5555555555511p-3);
                st_p = fma(st_f, st_p, 0x1.000000000000bp-1);
                st_p = fma(st_f, st_p, 1.0);
                st_p = fma(st_f, st_p, 1.0);
            };
            auto stage_c = [&](int e) -> double {
                double w = ldexp(st_p, (int)st_k);
                w = (1024.0 < st_t) ? INFINITY : w;
                w = (-1075.0 > st_t) ? 0.0 : w;
                const double s = S[e >> 2][e & 3];
                const double ae = a[e * 64];
                lin = fma(s, ae, lin);  // (used by the point columns only)
                ex += w;
                const double wp = masked(w, m_point);
                const double b2 = either(wp, either(masked(hw[e] * s, m_hp), masked(ae, m_prep)));
                hw[e] = either(wp, masked(hw[e], m_keep));
                return b2;
            };
            stage_a(0);
            stage_b();
            double b2 = stage_c(0), b2_next = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const bool more = e + 1 < NE;
#pragma unroll
                for (int ct = 0; ct < DT; ++ct) {
                    __builtin_amdgcn_sched_barrier(0);
                    R[ct] = e == 0 ? mfma64(A[0][ct], b2, d4{0.0, 0.0, 0.0, 0.0}) : mfma64(A[e & 1][ct], b2, R[ct]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        if (ct == 0) {
#pragma unroll
                            for (int c2 = 0; c2 < DT; ++c2) A[(e + 1) & 1][c2] = img2l[(c2 * NE + e + 1) * 64];
                            stage_a(e + 1);
                        }
                        if (ct == (DT > 2 ? 1 : 0)) stage_b();
                        if (ct == DT - 1) b2_next = stage_c(e + 1);
                    }
                }
                b2 = b2_next;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PROF) {
            asm volatile("s_nop 0" ::"v"(R[0][0]), "v"(R[DT - 1][3]), "v"(lin), "v"(ex));  // (complete before the clock is read)
            t1 = __builtin_amdgcn_s_memtime();
            pc[1] += t1 - t0;
            t0 = t1;
        }
        // ---- hand the results to the solves
        if (prep) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) sg[j * 64] = R[j >> 2][j & 3];
            mc.begin(mc.xk);
        } else if (req != ncgm::REQ_NONE) {
            V res;
            double f = 0.0;
            if (is_point) {
                double yy = 0.0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) yy = fma(y[j], y[j], yy);
                double v = V::reduce(lin);
                v -= V::reduce(ex);
                v -= V::reduce(yy) / (2 * variance);
                f = -v;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double g = -R[j >> 2][j & 3];
                    g += sg[j * 64];
                    g -= div_by(y[j], variance, inv_variance);
                    res.v[j] = -g;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) res.v[j] = R[j >> 2][j & 3] + div_by(y[j], variance, inv_variance);
            }
            mc.advance(f, res, dim, p.maxiter);
            if (mc.phase == ncgm::DONE) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double x = mc.xk.v[j];
                    if (x > 0.0 && x < kEpsB) x = kEpsB;
                    if (x < 0.0 && x > -kEpsB) x = -kEpsB;
                    if (4 * j + q < dim) p.U[n_slot * dim + 4 * j + q] = x;
                }
                if (p.status && q == 0) p.status[n_slot] = mc.status;
                mc.phase = ncgm::EMPTY;
            }
        }
        if (PROF) {
            t1 = __builtin_amdgcn_s_memtime();
            pc[4] += t1 - t0;
        }
    }
    if (PROF && lane == 0)
        for (int i = 0; i < 7; ++i) prof[wave_id * 7 + i] = pc[i];
}

template <int NE, int NJ>
bool launch(const SampleEmbeddingParams& p, hipStream_t stream) {
    using Lds = BatchedLds<NE, NJ>;
#ifdef SALNMF_DEV_PROFILE
    constexpr bool kProf = true;
#else
    constexpr bool kProf = false;
#endif
    auto kern = corr_sample_embeddings_batched_kernel<NE, NJ, kProf>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    // more than 64 KB of dynamic LDS needs the attribute, per function AND per device (a process may hold engines on several)
    // Set per launch: the call is a host-side table update (no device work), and a flag that remembers it would have to be
    // guarded -- two host threads driving engines could otherwise see the flag before the attribute is in place.
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lds::bytes) != hipSuccess) return false;
    // one workgroup per CU; every wave gets a contiguous range of at least 16 samples (one per slot) where N allows
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t max_waves = (int64_t)cus * BT_WAVES;
    const int64_t waves = std::max<int64_t>(1, std::min<int64_t>(max_waves, (p.N + 15) / 16));
    const int64_t per_wave = (p.N + waves - 1) / waves;
    const int grid = (int)((waves + BT_WAVES - 1) / BT_WAVES);
    long long* prof = nullptr;
#ifdef SALNMF_DEV_PROFILE
    const size_t n_prof = (size_t)grid * BT_WAVES * 7;
    if (hipMalloc(&prof, n_prof * sizeof(long long)) != hipSuccess) return false;
    (void)hipMemsetAsync(prof, 0, n_prof * sizeof(long long), stream);
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BT_BLOCK), Lds::bytes, stream, p, per_wave, prof);
#ifdef SALNMF_DEV_PROFILE
    std::vector<long long> h(n_prof);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h.data(), prof, n_prof * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipFree(prof);
    double sum[7] = {0, 0, 0, 0, 0, 0, 0}, max_total = 0, max_rounds = 0;
    for (size_t w = 0; w < n_prof / 7; ++w) {
        double tot = 0;
        for (int i = 0; i < 7; ++i) sum[i] += (double)h[w * 7 + i];
        for (int i = 0; i < 5; ++i) tot += (double)h[w * 7 + i];
        max_total = std::max(max_total, tot);
        max_rounds = std::max(max_rounds, (double)h[w * 7 + 5]);
    }
    const double nw = (double)(n_prof / 7), rounds = sum[5] / nw;
    fprintf(stderr,
            "[batched<%d,%d> N=%lld] per wave: %.0f rounds (max %.0f), %.1f of 16 slots occupied; clock ticks per round: refill %.0f, "
            "L.Y %.0f, phi + L^T.W %.0f, solver %.0f; total ticks per wave mean %.0f max %.0f\n",
            NE, NJ, (long long)p.N, rounds, max_rounds, sum[6] / sum[5], sum[0] / sum[5], sum[2] / sum[5], sum[1] / sum[5], sum[4] / sum[5],
            (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / nw, max_total);
#endif
    return true;
}

}  // namespace

bool launch_sample_embeddings_batched(const SampleEmbeddingParams& p, int terms, hipStream_t stream) {
    // instantiations by (term elements, vector elements) per lane: the smallest pair that covers (terms, dim)
    const int ne = (terms + 3) / 4, nj = (p.dim + 3) / 4;
#define SALNMF_BATCHED_CASE(NE_, NJ_) \
    if (ne <= NE_ && nj <= NJ_) return launch<NE_, NJ_>(p, stream);
    SALNMF_BATCHED_CASE(2, 2)
    SALNMF_BATCHED_CASE(4, 4)
    SALNMF_BATCHED_CASE(8, 4)
    SALNMF_BATCHED_CASE(8, 8)
    SALNMF_BATCHED_CASE(16, 8)
    SALNMF_BATCHED_CASE(10, 10)
    SALNMF_BATCHED_CASE(20, 10)
    SALNMF_BATCHED_CASE(12, 12)
    SALNMF_BATCHED_CASE(20, 12)
#undef SALNMF_BATCHED_CASE
    return false;  // larger problems: the one-wavefront-per-sample kernel
}

}  // namespace salnmf
