// CDNA4 (gfx950) kernels of the KL-NMF update path.  fp64 throughout.
//
// Device layout (owned by the engine; the C ABI converts from/to AnnData's compact layout):
//   X  [Np][96]   counts, sample-major, feature columns >= V are 0, rows >= N are 0
//   H  [Np][KP]   exposures, KP = 16*KT >= K; columns >= K and rows >= N hold finite filler
//   W  [K][V]     signatures, compact (38 KB; each workgroup stages it into LDS once)
//   w_kl, w_lhalf [Np]  per-sample weights (filler 1 / 0)
// Np = 16 * ceil(N/16).  One *tile* = 16 consecutive samples; with this padding every tile is
// full, every row is 128-byte aligned, and the hot loop needs no masks:
//   * pad rows have X = 0 and H > 0, so P > 0 and R = X/P = 0 exactly: nothing reaches G or U;
//   * pad feature columns have W_lds = 1 (P > 0, R = 0);
//   * pad signature rows of W_lds are 0, so pad columns of H never reach P, and the pad
//     columns of U / pad rows of G are simply never read.
//
// One wave64 owns a tile end to end; the four waves of a workgroup (one per SIMD) run
// independent tile streams and share only the LDS copy of W.  Per tile (reference arithmetic:
// _utils_klnmf.py:328-347):
//   P = Ht . W          16 x 96     v_mfma_f64_16x16x4, contraction over K
//   R = X / P           in the accumulator registers (never stored to HBM)
//   G += Ht^T . R       K x 96      contraction over the tile's samples; the accumulator tile R
//                                   is consumed directly as the B operand
//   U = R . W^T         16 x K      contraction over the features; R goes through a
//                                   wave-private LDS transpose to become the A operand
//   H <- clip(H * U)    written back in place
// f64 MFMA lane maps (checked on hardware by tools/mfma_f64_probe.hip):
//   A[i = lane&15][k = lane>>4]   B[k = lane>>4][j = lane&15]
//   D[row = (lane>>4) + 4*reg][col = lane&15],  reg = 0..3
// so register `reg` of a D tile is the B operand of k-step `reg` of a product that contracts
// over D's row index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "salnmf_logtab.h"
#include "salnmf_mv_device.h"

namespace salnmf {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr double kEps = 1.1920928955078125e-07;  // float32 eps, _utils_klnmf.py:7
constexpr int VT = 6;                 // feature tiles of 16  -> V <= 96
constexpr int VMAX = 16 * VT;         // leading dimension of X on the device
constexpr int VSTEPS = VMAX / 4;      // feature k-steps of 4
constexpr int WS = 98;                // LDS row stride (doubles) of W: 2*odd -> conflict-free column-slab reads
constexpr int RS = 98;                // LDS row stride (doubles) of the ratio tile
constexpr int WAVES = 4;              // one wave per SIMD
constexpr int BLOCK = 64 * WAVES;

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// Accumulate step on an AGPR-resident accumulator.  The s_nop covers the VALU-write ->
// MFMA-operand-read wait states, which hipcc does not insert inside an asm statement.
__device__ __forceinline__ void mfma_agpr(d4& c, double a, double b) {
    asm("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// Lower clip with numpy's semantics (ndarray.clip, _utils_klnmf.py:341,347): a NaN stays a NaN, so that a
// poisoned fit shows up in the objective instead of being masked (fmax would return the bound).
__device__ __forceinline__ double clip_lo(double x, double lo) { return x < lo ? lo : x; }

// x / p for the path's operands: reciprocal (v_rcp_f64: |r p - 1| < 4.7e-8 measured), ONE Newton step on it
// (-> ~2^-48), quotient, residual correction of the quotient (-> ~2^-96 before the final rounding).  hipcc's IEEE
// fp64 divide runs a second Newton step and range-scaling / fix-up instructions that only act on operands outside
// ~[1e-280, 1e280].  Bit-identical to `x / p` on 16.7 M probes covering counts, clipped zeros, 1e-30..1e30 ratios
// and x = 0 (tools/div_probe.hip, profiles/r02/div_probe.txt); the parity tests pin it.
__device__ __forceinline__ double div_path(double x, double p) {
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = x * r;
    double rem = __builtin_fma(-p, q, x);
    return __builtin_fma(rem, r, q);
}

// Exchange with lane (c16 ^ M) inside each row of 16 lanes, M in {8, 4, 2, 1}: DPP moves only
// (VALU, no LDS round trip).  xor 8 = row_ror:8, xor 4 = row_half_mirror then quad_perm[3,2,1,0],
// xor 2 = quad_perm[2,3,0,1], xor 1 = quad_perm[1,0,3,2].
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane of these permutations reads a live lane, so the "old" value is never used: an empty asm hands the
    // compiler an arbitrary register for it instead of a v_mov 0 per move
    int olo, ohi;
    asm volatile("" : "=v"(olo));
    asm volatile("" : "=v"(ohi));
    lo = __builtin_amdgcn_update_dpp(olo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(ohi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int M>
__device__ __forceinline__ double xor16(double v) {
    if (M == 8) return dpp_f64<0x128>(v);
    if (M == 4) return dpp_f64<0x1B>(dpp_f64<0x141>(v));
    if (M == 2) return dpp_f64<0x4E>(v);
    return dpp_f64<0xB1>(v);
}

// Sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the four q groups of an accumulator column), the same bits in all four,
// by v_permlane16_swap / v_permlane32_swap of the value with itself: VALU moves, no LDS round trip -- as two __shfl_xor steps
// (two ds_bpermute each on doubles, dependent) a sum costs a few hundred cycles.  The same pairs in the same order as
// t += shfl_xor(t, 16); t += shfl_xor(t, 32): a + b and b + a are the same bits.
__device__ __forceinline__ double rows_sum(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

template <int KS>
struct Geo {
    static constexpr int KT = (KS + 3) / 4;        // signature tiles of 16
    static constexpr int KP = 16 * KT;             // leading dimension of H on the device
    static constexpr int LS = KP + 2;              // LDS row stride of the H tile (2*odd: conflict-free A reads)
    static constexpr int WROWS = KP;               // rows of the LDS copy of W (zero padded)
    static constexpr int HL = 16 * LS;             // doubles per wave for the H tile
    static constexpr int RL = 16 * RS;             // doubles per wave for the ratio tile
    static constexpr int HV = KP / 8;              // 16-byte loads per lane that fetch one H tile
    static constexpr int LDS_DOUBLES = WROWS * WS + WAVES * (HL + RL);
};

struct FusedParams {
    const double* __restrict__ X;    // [Np][VMAX]
    double* __restrict__ H;          // [Np][KP]  read; DO_U writes the update to Hout (normally == H)
    double* Hout;                    // [Np][KP]  destination of the H update (CorrNMF aux: a separate buffer)
    double hfloor;                   // lower clip of the update: EPSILON, or 0 for aux = H * (R W^T) unclipped
    const double* __restrict__ W;    // [K][V]
    const double* __restrict__ wkl;  // [Np] or null
    const double* __restrict__ wlh;  // [Np] or null
    // WTS launches: always-valid arrays the tile loop loads its weights from -- wkl / wlh themselves, or the engine's
    // filler arrays (ones / zeros) where one of them is null.  wkl / wlh keep deciding, as flags, which arithmetic runs;
    // unconditional loads issued with the tile's other loads let hipcc count vmcnt instead of draining it at the loop's
    // back edge (the weighted step was 29 % slower than the unweighted one, profiles/r03/bench_final1.json).
    const double* __restrict__ wkl_eff;
    const double* __restrict__ wlh_eff;
    const double* __restrict__ hscale;  // [KP] or null: H is read as clip(H*hscale) (MvNMF trial)
    double* __restrict__ Gpart;      // [gridDim.x][K][VMAX]     (DO_G) per-workgroup partial numerators
    double* __restrict__ Hsumpart;   // [gridDim.x][K]           (DO_STATS) row sums of H
    double* __restrict__ KLpart;     // [gridDim.x]              (DO_STATS) unweighted partial of the KL divergence (tile_kl) (DO_U: optional, null = skip)
    const double* __restrict__ xlx;  // [Np][16]                 (DO_STATS) x-only constants of the KL terms per (sample, lane column): xlogx_lane_kernel
    double* __restrict__ KLpartB;    // [gridDim.x]              (MVJ) the numerator half's KL partial; KLpart then is the update_H half's (the trial's)
    const unsigned* skip_flag;       // (DO_STATS, optional) device word: non-zero = return at once (a queued MvNMF trial was rejected)
    int64_t N;
    int V;                  // features of this pass (<= 96: one feature block)
    int ldw;                // row stride of W (= V unless W points at one block of a wider matrix)
    int K;
    int64_t ntiles;
    int wdma;               // W -> LDS by LDS-DMA where the layout allows it (stage_W_dma_issue); 0: through registers (stage_W)
    // development builds (SALNMF_DEV_PROFILE): [waves of the grid][FK_NSEC + 1] shader-clock cycles per section of the pass
    // summed over the launches (+ the wave's number of tiles), or null -- printed by salnmf_destroy (profiles/r05/mvj_sections.md)
    unsigned long long* prof;
    // feature blocks (n_features > 96; fused_kernel<..., BLOCKED>, update_H pass only): U = R W^T is a sum over the
    // 96-feature blocks.  ublock = 1: first block, store U into Uacc; 2: add Uacc, store; 3: last block, add Uacc and
    // update H with the total.  Uacc is [Np][KP] like H.
    int ublock;
    double* Uacc;
    // MvNMF update_H pass only (fused_kernel<!DO_G, DO_U, DO_STATS>), both optional:
    //  * sideW != null: the LAST workgroup of the grid processes no tiles; it runs the W-only algebra of the W step
    //    (mv_prepare_W_body: A = W Y_minus, B = W |Y|, log det(W W^T + delta I), mvnmf.py:19-24,48-54) on sideW beside the
    //    pass -- the latency chain that used to need a second stream and an event wait on this one;
    //  * kl_out != null (with KLpart): the workgroup that finishes last sums the KL partials in the order of
    //    sum_partials_kernel and stores kl_out[0] -- one kernel and one boundary less per MvNMF step.
    //    kl_counter: arrival counter, zero between launches (the last arriver resets it).
    const double* sideW;
    double sideDelta;
    double* sideA;
    double* sideB;
    double* sideLogdet;
    double* kl_out;
    unsigned* kl_counter;
    // persistent multi-step mode (PERSIST instantiation only): the joint update_WH step nsteps times in ONE launch
    int nsteps;
    int n_given;
    double* Wmut;           // = W, written in place by the row owners
    double* G;              // [K][V] reduced numerator of the last step (what the W tail leaves behind)
    unsigned* sync;         // device words, zeroed before every launch: [0] slabs published, [32] W rows published, [64] abort
    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up
};

// log(x / p) for positive normal x, p with ONE division and no library call (the objective
// kernels are bound by fp64 VALU work, which shares the pipe with the MFMAs).
//   k  = round(log2(x/p)) estimated from the exponent/mantissa bits (integer ops only)
//   p' = p * 2^k (exponent-field add), so x/p' lies in about [0.67, 1.50]
//   s  = (x - p') / (x + p')      x - p' is exact (Sterbenz)
//   log(x/p) = k ln2 + log((1+s)/(1-s)) = k ln2 + 2s + s R(s^2), R fitted for this range
// Measured against a long-double reference (tools/log_probe.hip): abs error <= 2e-14 over
// |log| <= 460, relative error <= 1e-15 away from ratio = 1 and better than log(fl(x/p)) near it.
// Callers guarantee the operands are in range with log_operand_ok().  Used by the per-sample KL (forward mode 1); the
// objectives use log_pos below.
__device__ __forceinline__ bool log_operand_ok(double v) {
    // positive, normal, and far enough from the ends of the exponent range for the p * 2^k trick
    return (unsigned)(__double2hiint(v) - 0x03D00000) < (unsigned)(0x7C200000 - 0x03D00000);
}
__device__ __forceinline__ double log_ratio(double x, double p) {
    const int hx = __double2hiint(x), hp = __double2hiint(p);
    const int k = (hx - hp + 0x80000) >> 20;
    const double ps = __hiloint2double(hp + (k << 20), __double2loint(p));  // p * 2^k
    const double s = div_path(x - ps, x + ps);
    // R(z)/z: 8-coefficient Chebyshev fit of sum 2/(2i+3) z^i on z in [0, 0.041] (|s| <= 0.2025, the
    // range the integer estimate of k leaves); max error 3.1e-17, i.e. < 3e-19 on the logarithm
    const double z = s * s, w = z * z;
    const double t1 = __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 0.1365426141372305, 0.15389174135906675), 0.22222223148322984), 0.40000000000009306);
    const double t2 = __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 0.1320375159044889, 0.18181729869745253), 0.2857142856666864), 0.6666666666666666);
    const double R = z * __builtin_fma(z, t1, t2);
    const double kd = (double)k;
    double t = __builtin_fma(kd, 1.90821492927058770002e-10, s * R);  // kd*ln2_hi + (2s + (s*R + kd*ln2_lo))
    t = __builtin_fma(2.0, s, t);
    return __builtin_fma(kd, 6.93147180369123816490e-01, t);
}

// ---- log(p) for the objective terms, table driven (no division): p = 2^e m, m in [0.5, 1) (v_frexp_exp_i32_f64,
// v_frexp_mant_f64); entry i = top 8 mantissa bits holds inv_i ~ 1 / c_i (c_i the centre of the i-th mantissa interval)
// and lc_i = -log(inv_i);
//   r = m inv_i - 1 (one fma, |r| <= 2^-9),  log p = e ln2 + lc_i + (r - r^2/2 + r^3/3 - r^4/4 + r^5/5)
// (the next term is < 1e-17).  3 integer + 12 fp64-rate instructions and one 16-byte LDS read per logarithm against 26 fp64
// instructions (one of them a division chain) for log_ratio: the objective terms are fp64 VALU work on the pipe the
// MFMAs use.  Error <= 2.5e-16 max(|log p|, 0.5) (tools/gen_logtab.py on the host, tools/log_probe.hip on the device).
// The KL divergence is evaluated as  sum_d w_d sum_c [ c_dc + sum_{v = c mod 16} (p - x log p) ],  c_dc = sum_{v = c mod 16} (x log x - x)
// (0 where x = 0): the x-only part is computed once per upload of X (xlogx_lane_kernel, library log), so an objective costs
// ONE logarithm per entry, of p alone.  The constants are kept PER (sample, lane column) -- the six entries of a sample
// that one lane of the accumulator layout holds -- and are added in that lane, before any sum over lanes, samples or
// workgroups: x log x and x log p are each ~|x log x| and cancel to the entry's KL term, and a cancellation that happens
// only in the final scalar (one constant per sample or per matrix) costs digits in proportion to sum |x log x| / KL,
// 3-7 of them for counts of 1e5-1e6 or near-perfect fits.  Local, the error is eps |x log x| per lane and sample,
// independent of the problem size and of the launch geometry.  _utils_klnmf.py:41-53: same value to rounding (entries
// with x = 0 contribute p).
constexpr int LOGTAB_DOUBLES = 2 * LOGTAB_N;
__device__ __forceinline__ void stage_logtab(double* tab, int tid) {
    static_assert(LOGTAB_N == BLOCK, "one table entry per thread");
    reinterpret_cast<d2*>(tab)[tid] = reinterpret_cast<const d2*>(kLogTab)[tid];
}
// positive, normal, finite
__device__ __forceinline__ bool log_pos_ok(double v) { return (unsigned)(__double2hiint(v) - 0x00100000) < 0x7FE00000u; }

// M independent logarithms, stage by stage (the M dependent chains stand next to each other, as in log_ratio_n)
template <int M>
__device__ __forceinline__ void log_pos_n(const double (&p)[M], const double* __restrict__ tab, double (&out)[M]) {
    double kd[M], m[M], r[M], r2[M], h[M];
    d2 te[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const int hi = __double2hiint(p[i]);
        te[i] = *reinterpret_cast<const d2*>(tab + 2 * ((hi >> 12) & 0xFF));
        kd[i] = (double)__builtin_amdgcn_frexp_exp(p[i]);
        m[i] = __builtin_amdgcn_frexp_mant(p[i]);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) r[i] = __builtin_fma(m[i], te[i][0], -1.0);
#pragma unroll
    for (int i = 0; i < M; ++i) r2[i] = r[i] * r[i];
#pragma unroll
    for (int i = 0; i < M; ++i) h[i] = __builtin_fma(r[i], 0.2, -0.25);
#pragma unroll
    for (int i = 0; i < M; ++i) h[i] = __builtin_fma(r[i], h[i], 0.33333333333333331);
#pragma unroll
    for (int i = 0; i < M; ++i) h[i] = __builtin_fma(r[i], h[i], -0.5);
#pragma unroll
    for (int i = 0; i < M; ++i) h[i] = __builtin_fma(r2[i], h[i], r[i]);                          // log1p(r)
#pragma unroll
    for (int i = 0; i < M; ++i) r[i] = __builtin_fma(kd[i], 6.93147180369123816490e-01, te[i][1]);  // e ln2_hi + lc (product exact)
#pragma unroll
    for (int i = 0; i < M; ++i) h[i] = __builtin_fma(kd[i], 1.90821492927058770002e-10, h[i]);
#pragma unroll
    for (int i = 0; i < M; ++i) out[i] = r[i] + h[i];
}
__device__ __forceinline__ double log_pos(double p, const double* __restrict__ tab) {
    const double pp[1] = {p};
    double o[1];
    log_pos_n<1>(pp, tab, o);
    return o[0];
}

// x-free part of one entry of the generalised KL divergence for any operands (library log): p - x log p, and p alone
// where x == 0 (_utils_klnmf.py:47-50: such entries contribute only WH)
__device__ __forceinline__ double kl_term_p(double x, double p) {
    double t = p;
    if (x != 0.0) t -= x * log(p);
    return t;
}
// the x-only part of an entry: x log x - x, 0 where x == 0
__device__ __forceinline__ double kl_term_x(double x) { return x != 0.0 ? x * log(x) - x : 0.0; }

// ---- in-launch synchronisation of the persistent kernel (cdna_hip_programming.md, guideline 16) ----
// Payloads (G slabs, W rows) are stored write-through (sc1), every storing wave drains its stores
// (s_waitcnt vmcnt(0)) and joins a workgroup barrier, then ONE lane bumps a monotonic agent-scope counter.
// Consumers poll that counter with relaxed agent-scope loads from one lane, join a workgroup barrier and then read
// the payload with sc1 loads only (they bypass this CU's L1).  Every wait is bounded: ~0.25 s on the 100 MHz
// clock, or another workgroup having given up; on failure the abort words are set and the workgroup exits.
constexpr int SYNC_SLABS = 0, SYNC_WROWS = 32, SYNC_ABORT = 64, SYNC_WORDS = 128;
typedef __attribute__((address_space(1))) unsigned gsync_t;
__device__ __forceinline__ bool wait_counter(unsigned* sync_, int which, unsigned target, unsigned* abort_host) {
    gsync_t* sync = (gsync_t*)sync_;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 1;; ++spins) {
        if (__hip_atomic_load(sync + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63u) == 0) {
            const bool gave_up = __hip_atomic_load(sync + SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (gave_up || __builtin_amdgcn_s_memrealtime() - t0 > 25000000ull) {
                __hip_atomic_store(sync + SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((gsync_t*)abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
    }
}
__device__ __forceinline__ void bump_counter(unsigned* sync, int which) {
    __hip_atomic_fetch_add((gsync_t*)sync + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// W -> LDS with the padding described at the top of the file.  All loads in flight together.
//   ldw: row stride of W in memory (= V; larger when W points at one 96-feature block of a wider signature matrix)
template <int WROWS, bool SC1 = false>
__device__ __forceinline__ void stage_W(double* Wl, const double* W, int K, int V, int ldw, int tid) {
    constexpr int WPT = (WROWS * VMAX + BLOCK - 1) / BLOCK;
    constexpr int TOTAL = WPT * BLOCK;
    double wreg[WPT];
    // Every workgroup reads the same 38 KB at the same moment: the 32 workgroups that share an XCD's L2 start
    // at 32 different offsets so that they do not all queue on the same L2 channel.  Unconditional loads from
    // clamped addresses (all in flight together), the padding is selected afterwards: per-element branches
    // would serialise the loads.
    const int rot = 0;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        int idx = tid + BLOCK * j + rot;
        idx = idx >= TOTAL ? idx - TOTAL : idx;
        int k = idx / VMAX, v = idx - k * VMAX;
        const double* src = W + (k < K ? k : K - 1) * ldw + (v < V ? v : V - 1);
        wreg[j] = SC1 ? __hip_atomic_load((const __attribute__((address_space(1))) double*)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
    }
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        int idx = tid + BLOCK * j + rot;
        idx = idx >= TOTAL ? idx - TOTAL : idx;
        int k = idx / VMAX, v = idx - k * VMAX;
        const double w = (k < K) ? ((v < V) ? wreg[j] : 1.0) : 0.0;
        if (k < WROWS) Wl[k * WS + v] = w;
    }
}

// The same image by LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes go from its global address straight to LDS at
// base + 16 l, no register in between) for the common case V == ldw == 96, W 16-byte aligned.  A row of W is 48 pieces
// of 16 bytes, a row of the image 49 (WS = 98 doubles: the 49th piece is the stride padding, which nothing reads), so piece
// s of the image is row s / 49, piece s % 49, and instruction i of wave w moves pieces 64 (4 i + w) .. + 63.  Only ISSUES
// the loads (2 address registers live, against stage_W's 24 values per lane): the caller waits for them
// (s_waitcnt vmcnt(0)) and joins the workgroup barrier.  The zero rows k >= K are plain stores.
template <int WROWS>
__device__ __forceinline__ void stage_W_dma_issue(double* Wl, const double* W, int K, int tid) {
    static_assert(WS == VMAX + 2, "a row of the image = 48 data pieces + 1 pad piece");
    constexpr int PPR = WS / 2;                                  // pieces per image row
    constexpr int NI = (WROWS * PPR + BLOCK - 1) / BLOCK;        // instructions per wave
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npieces = K * PPR;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int s0 = 64 * (WAVES * i + wave);                  // (wave-uniform) first piece of this instruction
        const int s = s0 + lane;
        const int k = s / PPR, c = s - k * PPR;
        if (s0 < npieces) {                                      // (uniform branch: no instruction for rows beyond K)
            if (s < npieces && c < PPR - 1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + k * VMAX + 2 * c),
                                                 (__attribute__((address_space(3))) void*)(Wl + 2 * s0), 16, 0, 0);
        }
    }
    for (int i = K * WS + tid; i < WROWS * WS; i += BLOCK) Wl[i] = 0.0;
}

// Sum over one tile (accumulator layout: rows n = q+4r, columns v = 16vt+c16) of the per-sample weighted
//   c_dc + sum_{v of this lane} (p - x log p)
// i.e. the tile's share of the KL divergence (see log_pos above).  cv[r] is the x-only constant of (sample n0 + q + 4r,
// lane column c16), loaded by the caller with the tile's other loads; ROWS: wv[r] is the sample's weight, without ROWS the
// sum is unweighted.
// Entries outside [0,N) x [0,V) are skipped.  The objective terms are VALU-issue bound, so the common case -- a full
// tile of a 96-feature problem whose P are all positive normal numbers, always the case inside fit() -- runs without a
// single select: the range check is two min3/max3 chains over the high words.  Partial tiles and V < 96 take the masked
// form (their pad entries have x = 0, hence constants 0); a P that is zero, denormal or not finite (an all-zero row of H
// or W through the function-level API) sends the whole tile to the library path.
//   MB: logarithms evaluated side by side (their intermediates are 14 registers each: the joint step with the objective
//   folded in has room for three at a time)
template <bool ROWS, int MB = VT>
__device__ __forceinline__ double tile_kl(const double (&x)[VT][4], const d4 (&pr)[VT], const double (&wv)[4], const double (&cv)[4],
                                          const double* __restrict__ tab, int64_t n0, int64_t N, int V, int q, int c16) {
    if (n0 + 16 <= N && V == VMAX) {  // (wave-uniform)
        unsigned lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned h = (unsigned)__double2hiint(pr[vt][r]);
                lo = h < lo ? h : lo;
                hi = h > hi ? h : hi;
            }
        if (__all(lo >= 0x00100000u && hi < 0x7FF00000u)) {
            double total = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                static_assert(VT % MB == 0, "batches of equal size");
                double acc = cv[r];
#pragma unroll
                for (int b = 0; b < VT; b += MB) {
                    double ps[MB], lp[MB];
#pragma unroll
                    for (int i = 0; i < MB; ++i) ps[i] = pr[b + i][r];
                    log_pos_n<MB>(ps, tab, lp);
#pragma unroll
                    for (int i = 0; i < MB; ++i) acc += __builtin_fma(-x[b + i][r], lp[i], ps[i]);
                    if (MB != VT) asm volatile("" : "+v"(acc));  // one batch after the other
                }
                if (ROWS) acc *= wv[r];
                total += acc;
            }
            return total;
        }
    }
    bool ok = true;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = (n0 + q + 4 * r < N) && (16 * vt + c16 < V);
            ok &= !valid || log_pos_ok(pr[vt][r]);
        }
    double total = 0.0;
    if (__all(ok)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool nvalid = n0 + q + 4 * r < N;
            double acc = nvalid ? cv[r] : 0.0;
            double ps[VT], lp[VT];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const bool valid = nvalid && (16 * vt + c16 < V);
                ps[vt] = valid ? pr[vt][r] : 1.0;  // pads: a harmless operand
            }
            log_pos_n<VT>(ps, tab, lp);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const bool valid = nvalid && (16 * vt + c16 < V);
                const double t = __builtin_fma(-x[vt][r], lp[vt], ps[vt]);
                acc += valid ? t : 0.0;
            }
            if (ROWS) acc *= wv[r];
            total += acc;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool nvalid = n0 + q + 4 * r < N;
            double acc = nvalid ? cv[r] : 0.0;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
                if (nvalid && 16 * vt + c16 < V) acc += kl_term_p(x[vt][r], pr[vt][r]);
            if (ROWS) acc *= wv[r];
            total += acc;
        }
    }
    return total;
}

// ----------------------------------------------------------------------------------------------
// W tail arithmetic (_utils_klnmf.py:338-341 / :208-215), shared by tail_kernel and the persistent kernel.
constexpr int TAIL_PARTS = 8;
constexpr int TAIL_BLOCK = VMAX * TAIL_PARTS;

// agent-scope relaxed accesses = global_load / global_store ... sc1: they bypass this CU's L1, which is how
// bytes written by another workgroup of the SAME launch are read (persistent kernel); plain otherwise
// (address_space(1) makes them global_ instructions; a flat_ access must not carry a hand-off)
typedef __attribute__((address_space(1))) double gdouble;
template <bool SC1>
__device__ __forceinline__ double ld_shared(const double* ptr) {
    if (SC1) return __hip_atomic_load((const gdouble*)ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *ptr;
}
template <bool SC1>
__device__ __forceinline__ void st_shared(double* ptr, double v) {
    if (SC1) __hip_atomic_store((gdouble*)ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *ptr = v;
}

struct TailScratch {
    double red[TAIL_PARTS][VMAX];
    double wn[VMAX];
    double rowsum;
};

// Row k of the W tail, by NT threads (768 in tail_kernel, 256 inside the persistent kernel): the arithmetic and
// every summation order are the same for both, so the two paths give the same bits.
//   work item c = part * VMAX + v:  partial sum over the slabs part, part + 8, part + 16, ... in ascending order
//   (16 independent loads in flight per item and round), then the 8 parts in order, then W' = W*G, the row sum
//   sequentially over v, normalise, keep given rows, clip.
// The slabs are [nslabs][K][VMAX] (row stride VMAX whatever V is).
//   Wout: where the new row goes (normally W itself; the first step of a kept block writes a second buffer)
//   nslabs: > 0 sum the slabs into G; 0 take the reduced row from G; < 0 it is in S.red[0] already (wold_in = the old
//   row of W, loaded by the caller beside its other loads)
template <int NT, bool SC1>
__device__ __forceinline__ void tail_row(TailScratch& S, int tid, int k, const double* Gpart, int nslabs, double* G, const double* W,
                                         double* Wout, int V, int K, int n_given, int clip_mode, bool do_tail, double wold_in = 0.0) {
    constexpr int NC = TAIL_BLOCK / NT;  // work items per thread
    static_assert(NC * NT == TAIL_BLOCK, "thread count must divide the work items");
    // loads in flight per item and round: with one item per thread all 32 slabs of a 256-workgroup grid at once
    // (the tail is a chain of memory round trips: one for the slabs instead of two), the old row beside them
    constexpr int B = NC == 1 ? 32 : 16;
    double wold = wold_in;
    if (do_tail && nslabs >= 0 && tid < VMAX && tid < V) wold = ld_shared<SC1>(W + k * V + tid);
    if (nslabs > 0) {
        double s[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) s[c] = 0.0;
        const int64_t slab = (int64_t)K * VMAX;
        for (int base = 0; base < nslabs; base += TAIL_PARTS * B) {
            double t[NC][B];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int item = tid + c * NT, part = item / VMAX, v = item - part * VMAX;
                const double* src = Gpart + (int64_t)k * VMAX + v;
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const int sl = base + part + j * TAIL_PARTS;
                    t[c][j] = (v < V && sl < nslabs) ? ld_shared<SC1>(src + sl * slab) : 0.0;
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < B; ++j) s[c] += t[c][j];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int item = tid + c * NT, part = item / VMAX, v = item - part * VMAX;
            S.red[part][v] = s[c];
        }
        __syncthreads();
        if (tid < VMAX && tid < V) {
            double t = 0.0;
            for (int i = 0; i < TAIL_PARTS; ++i) t += S.red[i][tid];
            G[k * V + tid] = t;
            S.red[0][tid] = t;
        }
        __syncthreads();
    } else if (nslabs == 0) {
        if (tid < VMAX && tid < V) S.red[0][tid] = G[k * V + tid];
        __syncthreads();
    }
    if (!do_tail) return;
    const int v = tid;
    if (tid < VMAX) S.wn[v] = (v < V) ? wold * S.red[0][v] : 0.0;
    __syncthreads();
    // row sum in a fixed two-level order: 12 groups of 8 consecutive features (sequential inside a group, one thread
    // each), then the groups in order -- 20 dependent additions on the critical path instead of 96
    if (tid < VMAX / 8) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += S.wn[8 * tid + i];  // (entries v >= V are 0)
        S.red[1][tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < VMAX / 8; ++i) t += S.red[1][i];
        S.rowsum = t;
    }
    __syncthreads();
    if (tid < VMAX && v < V) {
        double w = S.wn[v] / S.rowsum;
        if (k < n_given) {
            w = wold;
            if (clip_mode == 0) w = clip_lo(w, kEps);
        } else {
            w = clip_lo(w, kEps);
        }
        st_shared<SC1>(Wout + k * V + v, w);
    }
}

// End of one step of the persistent kernel (kept out of line: its registers are not the tile loop's; scalar
// arguments, so that the kernel's parameter block is not copied to the stack).
// Returns false when a wait gave up (the whole workgroup then exits).
__device__ __attribute__((noinline)) bool persist_publish_and_tail(unsigned* sync, unsigned* abort_host, const double* Gpart, double* G,
                                                                   double* W, int K, int V, int n_given, double* lds, int step, int tid) {
    // publish the slab: every storing wave drains its write-through stores, then one lane counts the workgroup in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) bump_counter(sync, SYNC_SLABS);
    // rows of the W tail: the workgroups with the highest indices own them (they have the fewest tiles)
    bool waited = false;
    for (int k = (int)gridDim.x - 1 - (int)blockIdx.x; k < K; k += (int)gridDim.x) {
        if (!waited) {
            int* okp = reinterpret_cast<int*>(lds + sizeof(TailScratch) / sizeof(double) + 2);
            if (tid == 0) *okp = wait_counter(sync, SYNC_SLABS, (unsigned)(step + 1) * gridDim.x, abort_host);
            __syncthreads();
            if (*okp == 0) return false;
            waited = true;
        }
        tail_row<BLOCK, true>(*reinterpret_cast<TailScratch*>(lds), tid, k, Gpart, (int)gridDim.x, G, W, W, V, K, n_given, 0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) bump_counter(sync, SYNC_WROWS);
    }
    __syncthreads();  // the LDS scratch of the tail is free again
    return true;
}

// Start of a step > 0 of the persistent kernel: wait until the K rows of the new W are published.
__device__ __attribute__((noinline)) bool persist_wait_W(unsigned* sync, unsigned* abort_host, unsigned target, double* lds, int tid) {
    int* okp = reinterpret_cast<int*>(lds);
    if (tid == 0) *okp = wait_counter(sync, SYNC_WROWS, target, abort_host);
    __syncthreads();
    const bool ok = *okp != 0;
    __syncthreads();
    return ok;
}

// Epilogue geometry of the fused kernel's numerator reduction, shared with the cooperative tile.  The cross-wave sum runs
// in ROUNDS feature ranges of VTR tiles: in a round EVERY wave parks ALL NT = KT * VTR accumulator tiles of the range in LDS
// (area [wave][tile][reg][lane]), one barrier, then wave w sums tiles w, w + 4, ... over the four waves in the order
// 0 + 1 + 2 + 3 and stores them.  The cooperative leftover tile parks its contribution (one per tile, index kt * VT + vt)
// and its remainder rows in the idle waves' LDS; the owners take theirs into registers before the rounds reuse that memory.
template <int KT, int KR, int LDS_DOUBLES>
struct EpiGeo {
    static constexpr int REMD = KR > 0 ? WAVES * KR * VMAX : 0;  // parked remainder rows of the four waves
    static constexpr int COOP_REM = KT * VT * 256;               // offset of the remainder rows inside the cooperative park
    static constexpr int COOP_DOUBLES = COOP_REM + (KR > 0 ? KR * VMAX : 0);
    static constexpr int need(int rounds) { return WAVES * KT * (VT / rounds) * 256 + REMD; }
    static constexpr int ROUNDS = need(1) <= LDS_DOUBLES ? 1 : (need(2) <= LDS_DOUBLES ? 2 : (need(3) <= LDS_DOUBLES ? 3 : 6));
    static_assert(need(ROUNDS) <= LDS_DOUBLES, "the parked accumulator tiles must fit in LDS");
    static constexpr int VTR = VT / ROUNDS, NT = KT * VTR;
    static constexpr int MAXI = (NT + WAVES - 1) / WAVES;        // tiles a wave owns per round, at most
    static constexpr int PARK = WAVES * NT * 256;                // the rounds' area; the four waves' remainder rows follow it
};

// H update with an l-half penalty (_utils_klnmf.py:349-361): I = 4 H (W^T aux) [w_kl^2]; D = w_lh^2 / 4 + I;
// H' = (w_lh / 2 - sqrt(D))^2 / 4 [/ w_kl^2]  (the cooperative tile's form of the statements in process_tile)
__device__ __forceinline__ double lhalf_update(double h, double u, double wl, double wk, bool has_wkl) {
    const double wk2 = wk * wk;
    double inter = 4.0 * h * u;
    if (has_wkl) inter *= wk2;
    const double disc = 0.25 * wl * wl + inter;
    const double t = wl / 2 - sqrt(disc);
    double hn = 0.25 * (t * t);
    if (has_wkl) hn /= wk2;
    return hn;
}

}  // namespace salnmf

// the kernels themselves, one header each
#include "salnmf_fused_kernel.h"    // fused_kernel: the update pass
#include "salnmf_forward_kernel.h"  // forward_kernel: W @ H and the objective terms
#include "salnmf_plain_kernels.h"   // tail_kernel and the plain helper kernels
