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

// ----------------------------------------------------------------------------------------------
// Fused update pass.
//   DO_G     accumulate G = (w_kl * R)^T-contracted numerator for the W update
//   DO_U     update H in place
//   DO_STATS MvNMF by-products: with DO_U the row sums of the *updated* H (rowsums_H of
//            update_W_unconstrained, mvnmf.py:54); with DO_G the unweighted KL(X || WH) partial
//            (the f0 of the line search, mvnmf.py:79)
//
// Register plan (one wave per SIMD, 512 registers): the K x V accumulator G lives in AGPRs for
// the whole kernel (inline-asm MFMA with "a" operands), everything else in <= 256 VGPRs.
// LDS operand reads are software-pipelined one k-step ahead of the MFMAs that consume them;
// sched_barrier(0) pins that order.
//
// Output-side signature columns: KTM full 16-wide tiles go through MFMA; KR (0..4) remainder
// columns k = 16*KTM + j are done on the VALU instead of spending a whole MFMA tile on them
// (K = 50: KTM = 3, KR = 2 -- 222 instead of 270 MFMAs per tile).  KR = 0: KTM = ceil(K/16).
//
// WTS: the instantiation that honours p.wkl / p.wlh.  The unweighted one has no conditional loads in
// the tile loop, so hipcc can count vmcnt exactly: with them it falls back to `s_waitcnt vmcnt(0)` at the
// loop's back edge, which exposes the latency of the H stores of every tile (~1.1 k cycles per tile).
//
// PERSIST: p.nsteps joint update_WH steps in ONE launch (requires every workgroup of the grid to be resident: one
// per CU, grid <= number of CUs).  The workgroups stay on their CUs; per step each publishes its numerator slab,
// the workgroup that owns signature row k (the ones that run out of tiles first) waits for all slabs, runs the
// W tail of that row (tail_row: same arithmetic and summation order as tail_kernel) and publishes the new
// row; everybody waits for the K rows, re-stages W into LDS and goes on.  That replaces two kernel boundaries,
// the tail launch and the launch ramp per step by two counter hand-offs, and the first tile of the next step is
// already in flight while a workgroup waits.
//
// BLOCKED (n_features > 96, update_H pass only): this launch covers ONE 96-feature block of X and W; the product
// U = R W^T is accumulated over the blocks' launches through p.Uacc (p.ublock), the last block updates H.
//
// RGIVEN (n_signatures > 64, one launch per chunk of <= 64 signatures): p.X holds the ratio R = X / (H W) over ALL
// signatures (forward_kernel mode 4 at the end of a chain over the chunks) instead of X; the P phase and the division
// are skipped, everything downstream -- G and U of this chunk's rows / columns, the H update -- is unchanged.
//   MVJ (MvNMF, with DO_G, DO_U, DO_STATS): update_H and the numerator pass behind it in ONE pass over the samples.  A
//   sample's new exposures and its contribution to the next W step's numerator depend on that sample alone, so per tile
//   the update_H half (P = H W, the trial's KL, R, U, H') and the numerator half on H' (P' = H' W, KL = f0, R', G += H'^T R')
//   run back to back: X stays in registers, H' in the wave's LDS tile.  Per entry the arithmetic is the two passes'; the
//   tile -> wave mapping and the slab order are those of an update_H pass with its side workgroup (salnmf.hip: the
//   stand-alone numerator pass of an MvNMF step runs on the same number of tile workgroups), so the bits are the same.
// Section clocks of a development build: s_memtime at the phase boundaries of a tile (each read drains the wave's LDS queue,
// so the software pipelining across a boundary is lost: the table says where the time goes, the sum is a few per cent
// above the production kernel's).  Sections: 0 prologue; per half h (0: the whole tile or the update_H half, 1: MVJ's
// numerator half) 1 + 7 h + {0 stage H, 1 P product, 2 KL terms, 3 divisions, 4 prefetch issue + R transpose, 5 G phase,
// 6 U phase + H update}; 15 epilogue.
constexpr int FK_NSEC = 16;
template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false, bool PERSIST = false, bool BLOCKED = false, bool RGIVEN = false,
          bool MVJ = false>
__global__ void __launch_bounds__(BLOCK, 1) fused_kernel(FusedParams p) {
#ifdef SALNMF_DEV_PROFILE
    unsigned long long tks[FK_NSEC] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk0 = __builtin_amdgcn_s_memtime(), tk1, ntl = 0;
    // (sched_barrier on both sides: the machine scheduler moves nothing across a boundary)
#define FK_TICK(i) do { __builtin_amdgcn_sched_barrier(0); tk1 = __builtin_amdgcn_s_memtime(); tks[i] += tk1 - tk0; tk0 = tk1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FK_TICK(i) do { } while (0)
#endif
    static_assert(!MVJ || (DO_G && DO_U && DO_STATS && !WTS && !PERSIST && !BLOCKED && !RGIVEN), "MVJ: the unweighted MvNMF pass pair");
    static_assert(!RGIVEN || (WTS && !DO_STATS && !PERSIST), "given ratio: the weighted-capable plain passes only (with BLOCKED: the update_H pass of one feature block)");
    static_assert(!PERSIST || (DO_G && DO_U && !DO_STATS && !WTS), "the persistent mode is the plain joint step");
    static_assert(!BLOCKED || (DO_U && !DO_G && !DO_STATS && WTS && !PERSIST), "feature blocks: the weighted-capable update_H pass only");
    using G_ = Geo<KS>;
    constexpr int KT = KTM;  // MFMA tiles on the output side
    constexpr int KP = G_::KP, LS = G_::LS, HV = G_::HV;
    constexpr int KB = 16 * KTM;                                   // first remainder column
    constexpr int NVP = KR == 0 ? 0 : (KR == 1 ? 4 : (KR == 2 ? 8 : 16));  // 4*KR values, padded to a power of two
    static_assert(KR >= 0 && KR <= 4 && KB + KR <= KP, "remainder columns must fit the padded layout");
    // Rows of W any phase reads: the P product 4 KS (its k-steps), the U product and the remainder columns 16 KTM + KR.
    // (Round 5: K = 50 stages 52 rows, not the 64 of the padded layout: 9 KB of LDS that the second H buffer below needs.)
    constexpr int WROWS = 4 * KS > 16 * KTM + KR ? 4 * KS : 16 * KTM + KR;
    static_assert(WROWS <= G_::WROWS, "never more rows than the padded layout");
    // HDMA (round 5, the joint steps): the NEXT tile's H travels straight from global memory into a second LDS
    // tile by LDS-DMA (global_load_lds_dwordx4) instead of through 32 prefetch registers and a staging pass at the top of
    // the tile -- a row of the tile is KP / 2 pieces of 16 bytes, a row of the LDS image one more (LS = KP + 2), as for W.
    // hipcc tracks LDS-DMA writes against vmcnt itself, so the first read of the new tile waits for exactly these loads.
    // Needs the tile's rows as they are in memory: the host applies a pending rescale of H before launching this variant.
    // (every joint step -- plain, weighted, with the objective folded in -- where 160 KB have the room)
    constexpr bool HDMA_FITS = (WROWS * WS + WAVES * (2 * G_::HL + G_::RL) + KP + (DO_STATS ? LOGTAB_DOUBLES : 0) + (WTS ? WAVES * 32 : 0)) * 8 <= 160 * 1024;
    constexpr bool HDMA = DO_G && DO_U && !PERSIST && !BLOCKED && !RGIVEN && !MVJ && HDMA_FITS;
    constexpr int REGION = (HDMA ? 2 : 1) * G_::HL + G_::RL;  // per wave: [H tile | R tile | (HDMA) second H tile]
    constexpr int LDSD = WROWS * WS + WAVES * REGION;
    // The epilogue's cross-wave sum wants 4 x (all accumulator tiles) in ONE round (EpiGeo): where the tile loop's layout is
    // smaller than that the array is simply made as large as the round needs (one workgroup per CU either way), up to the
    // 160 KB a workgroup can have; the small arrays behind it (hscale copy, log table, weights) keep their own space.
    constexpr int LDS_EXTRA = KP + (DO_STATS ? LOGTAB_DOUBLES : 0) + (WTS ? WAVES * 32 : 0);
    using CO_ = EpiGeo<KT, KR, 160 * 1024 / 8 - LDS_EXTRA>;
    constexpr int LDS_MAIN = LDSD > CO_::need(CO_::ROUNDS) ? LDSD : CO_::need(CO_::ROUNDS);
    __shared__ __attribute__((aligned(16))) double lds[LDS_MAIN + LDS_EXTRA];

    // MvNMF update_H pass: the grid's last workgroup may be the one that does the W-only algebra instead of tiles
    constexpr bool MVU = DO_U && DO_STATS && (!DO_G || MVJ);
    // (DO_G && DO_U && DO_STATS: the joint step that also evaluates the KL divergence of the state it starts from -- the
    // objective of a convergence test, folded into the first step of the next block; no row sums of H there)
    // (MvNMF, queued steps: a trial was rejected on the device -- everything queued behind it is a no-op)
    if (DO_STATS && p.skip_flag != nullptr && __hip_atomic_load((gsync_t*)p.skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int nwg = (int)gridDim.x - ((MVU && p.sideW != nullptr) ? 1 : 0);  // workgroups that process tiles
    if (MVU && p.sideW != nullptr && (int)blockIdx.x == nwg) {
        static_assert(!MVU || KP * (MV_WS + 2 * MV_LD + 1) + 1 <= LDSD, "the W-only algebra must fit this geometry's LDS");
        const int K = p.K;  // <= KP: the three matrices are packed by K rows
        double* Wl = lds;
        double* S = Wl + K * MV_WS;
        double* T = S + K * MV_LD;
        mv_prepare_W_body<BLOCK>(p.sideW, K, p.V, p.sideDelta, p.sideA, p.sideB, p.sideLogdet, Wl, S, T, T + K * MV_LD);
        return;
    }
    // (persistent mode) everything a step needs is derived inside the step loop from an opaque copy of the thread
    // index, so that nothing but the step counter is live across the out-of-line synchronisation calls
    const int nsteps = PERSIST ? p.nsteps : 1;
    for (int step = 0; step < nsteps; ++step) {
    int tid = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const int V = p.V, K = p.K;
    const int64_t N = p.N;
    const double* __restrict__ const wkl = WTS ? p.wkl : nullptr;
    const double* __restrict__ const wlh = WTS ? p.wlh : nullptr;

    double* Wl = lds;
    double* Hl = lds + WROWS * WS + wave * REGION;  // (HDMA: the tile being worked on; changes places with Hnx after every tile)
    double* const Rl = lds + WROWS * WS + wave * REGION + G_::HL;
    double* Hnx = Hl + (HDMA ? G_::HL + G_::RL : 0);
    // (HDMA) the same two tiles as wave-uniform addresses for the DMA's M0, and which of them receives the next tile
    double* const Hdma0 = lds + WROWS * WS + __builtin_amdgcn_readfirstlane(wave) * REGION;
    int hsel = 0;

    double* hsl = lds + LDS_MAIN;  // [KP] copy of hscale
    if (p.hscale && tid < KP) hsl[tid] = p.hscale[tid];
    double* ltab = hsl + KP;              // (DO_STATS) table of log_pos
    if (DO_STATS) stage_logtab(ltab, tid);
    // (WTS) this wave's weights of the current tile: [16 rows][w_kl, w_lhalf], staged from the prefetch registers
    double* wgt = ltab + (DO_STATS ? LOGTAB_DOUBLES : 0) + wave * 32;

    d4 g[KT][VT];
    double grem[KR > 0 ? KR : 1][VT];  // remainder rows of G: per-lane partials over this lane's sample rows
    double hsum[KT > 0 ? KT : 1];  // column sums of the updated H over this lane's rows (columns 16kt+c16)
    double hsum_rem = 0.0;         // same for the remainder column this lane owns
    double klacc = 0.0;
    double klacc_b = 0.0;  // (MVJ) the numerator half's KL = f0 of the next step; klacc is the update_H half's (the trial's)
    if (MVU) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) hsum[kt] = 0.0;
    }

    const int64_t tstride = (int64_t)(MVU ? nwg : (int)gridDim.x) * WAVES;
    int64_t tile = 0;
    // Leftover round.  ntiles = R * (waves of the grid) + L: with 0 < L <= workgroups the L leftover tiles would keep
    // L waves busy for a whole tile time while the rest of the chip idles (c2: 106 of 1024 waves, 9.6 of 78 us).  In
    // the plain joint step they are instead worked on by all four waves of workgroup 0 .. L-1 (process_tile_coop below).
    // (Tried for the two MvNMF passes as well, KR == 0: both got slower at c4 -- 38.8 -> 41.7 and 45.2 -> 47.1 us; their
    // tile is half as long as the joint step's while the cooperative tile's fixed cost, three workgroup barriers and
    // loads that nothing hides, stays, and the statistics code in the shared tile costs the main loop registers:
    // profiles/r03/ab_step_variants.txt.)
    // The joint step with the objective folded in (DO_STATS) does the same: its numerator is summed in the plain step's order.
    constexpr bool COOP = DO_G && DO_U && !RGIVEN && !MVJ;  // (with per-sample weights too: process_tile_coop honours them)
    constexpr int CSLAB = WROWS * WS + REGION;  // the cooperative tile's park: the LDS regions of waves 1..3, free meanwhile
    static_assert(!COOP || CO_::COOP_DOUBLES <= (WAVES - 1) * REGION, "the cooperative tile's numerator park must fit the idle waves' LDS");
    const int64_t nleft = p.ntiles % tstride;
    const bool coop = COOP && nleft > 0 && nleft <= (int64_t)gridDim.x && p.hscale == nullptr;
    const int64_t nfull = coop ? p.ntiles - nleft : p.ntiles;  // tiles of the one-wave-per-tile rounds

    // lane's slice of an H tile: element pair e = 2*lane + 128*j of the contiguous [16][KP] block
    int hrow[HV], hcol[HV];
#pragma unroll
    for (int j = 0; j < HV; ++j) {
        int e = 2 * lane + 128 * j;
        hrow[j] = e / KP;
        hcol[j] = e - hrow[j] * KP;
    }

    // prefetch registers: H tile (16-byte pieces of the contiguous block) and X tile (accumulator layout)
    d2 hpre[HV];
    double x[VT][4];
    d2 wpre = (d2){1.0, 0.0};  // (WTS) {w_kl, w_lhalf} of row (lane & 15) of the prefetched tile

    // (HDMA) piece 64 i + lane of the LDS image of an H tile: its offset in the tile's [16][KP] block (doubles), -1 for the pad
    // piece of a row and beyond the sixteen rows
    constexpr int HPR = LS / 2, HNI = (16 * HPR + 63) / 64;
    // (the pad piece of a row receives a copy of the row's last data piece: nothing reads it, and the instruction needs no
    // lane mask -- only the last instruction, whose lanes run past the sixteenth row, has one)
    int hoff[HDMA ? HNI : 1];
    if (HDMA) {
#pragma unroll
        for (int i = 0; i < HNI; ++i) {
            const int sp = 64 * i + lane, row = sp / HPR, c = sp - row * HPR;
            hoff[i] = row < 16 ? row * KP + 2 * (c < HPR - 1 ? c : HPR - 2) : -1;
        }
    }
    auto load_tile_H = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t n0 = t * 16;
        if constexpr (HDMA) {
            // As inline assembly, not through __builtin_amdgcn_global_load_lds: around the builtin hipcc put s_waitcnt vmcnt(0)
            // in front of the X loads of the launch's first tile (one more memory round trip in the prologue) and between the
            // pieces on the register-staged path, yet none in front of the tile's first LDS read -- its bookkeeping of LDS-DMA
            // is of no use here, so the pieces are invisible to it and process_half waits for them by count.  That is sound
            // because vmcnt completes in issue order and no wait hipcc computes spans a DMA group: it waits for X loads (issued
            // right behind the group, volatile + memory clobber keep them there) and for nothing younger before the next group.
            const double* hsrc = p.H + n0 * KP;
            const unsigned m0base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(Hdma0 + hsel * (G_::HL + G_::RL));
#pragma unroll
            for (int i = 0; i < HNI; ++i)
                if (64 * (i + 1) <= 16 * HPR || hoff[i] >= 0)  // (compile-time true for all but the last instruction)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                                 :
                                 : "v"(hsrc + hoff[i]), "s"(__builtin_amdgcn_readfirstlane(m0base + 1024u * i))
                                 : "memory");
            hsel ^= 1;
        } else {
            const d2* hsrc = reinterpret_cast<const d2*>(p.H + n0 * KP) + lane;
#pragma unroll
            for (int j = 0; j < HV; ++j) hpre[j] = hsrc[64 * j];
        }
    };
    auto load_tile_X = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t n0 = t * 16;
        const double* xsrc = p.X + (n0 + q) * VMAX + c16;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[vt][r] = xsrc[4 * r * VMAX + 16 * vt];
        if (WTS) {
            wpre[0] = p.wkl_eff[n0 + c16];
            wpre[1] = p.wlh_eff[n0 + c16];
        }
    };
    // (H first: process_half's counted wait for the DMA pieces relies on the X loads being the younger ones)
    auto load_tile = [&](int64_t t) __attribute__((always_inline)) {
        load_tile_H(t);
        load_tile_X(t);
    };
    bool x_first = false;  // (HDMA) the tile about to be processed was loaded X first (the register-staged prologue only)
    // (HDMA) a workgroup with a cooperative leftover tile fetches that tile's operands while its LAST ordinary tile runs, in
    // the slot where the next tile's prefetch would go: wave 0 sends the H tile by DMA into its free H buffer, every wave
    // loads its own columns of X into the (free) X prefetch registers -- the cooperative tile then starts without a memory
    // round trip of its own (3.1 k cycles of its 12.6 k at c2, profiles/r05/epilogue.md)
    const bool coop_here = COOP && HDMA && coop && (int64_t)blockIdx.x < nleft;  // (uniform)
    bool coop_fetched = false;
    auto coop_prefetch = [&]() __attribute__((always_inline)) {
        const int64_t ct = nfull + blockIdx.x;
        const int wvu = __builtin_amdgcn_readfirstlane(wave);
        if (wvu == 0) load_tile_H(ct);  // -> buffer (tiles per wave) & 1 of wave 0's region
        const int vt0 = wvu < 2 ? 2 * wvu : wvu + 2, nvt = wvu < 2 ? 2 : 1;  // the feature tiles of this wave: {0,1} {2,3} {4} {5}
        const double* xsrc = p.X + (ct * 16 + q) * VMAX + c16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[i][r] = xsrc[4 * r * VMAX + 16 * (vt0 + (i < nvt ? i : 0))];  // (a wave with one tile loads it twice)
        coop_fetched = true;
    };

    // HALF 0: the whole tile as the template switches say; (MVJ) HALF 1: the update_H half, HALF 2: the numerator half on
    // the H' that half 1 left in the wave's LDS tile (no staging, no rescale, X still in registers)
    auto process_half = [&](int64_t tile, auto halftag) __attribute__((always_inline)) {
        constexpr int HALF = decltype(halftag)::value;
        constexpr bool TG = DO_G && HALF != 1, TU = DO_U && HALF != 2;
        const int64_t n0 = tile * 16;
        // (WTS) this tile's weights go to LDS with the H tile: the prefetch registers are reloaded in mid-tile, and
        // holding 8 weights per lane across the tile pushed the weighted joint kernel past its 512 registers
        if (WTS) *reinterpret_cast<d2*>(wgt + 2 * c16) = wpre;  // (the four q groups write the same values)
        // ---- stage the H tile (wave private; LDS ops of one wave are executed in order)
        if (!HDMA && HALF != 2 && p.hscale) {
            // MvNMF: H is read as clip(H * colsum(W_trial)) (a line-search trial, or the rescale of an accepted
            // one that no pass has materialised yet).  Applied here, where the prefetched tile is consumed
            // anyway, from the LDS copy of the scale
#pragma unroll
            for (int j = 0; j < HV; ++j) {
                hpre[j][0] = clip_lo(hpre[j][0] * hsl[hcol[j]], kEps);
                hpre[j][1] = clip_lo(hpre[j][1] * hsl[hcol[j] + 1], kEps);
            }
        }
        if (!HDMA && HALF != 2) {
#pragma unroll
            for (int j = 0; j < HV; ++j) {
                *reinterpret_cast<d2*>(Hl + hrow[j] * LS + hcol[j]) = hpre[j];
            }
        }
        if (HDMA) {
            // This tile's H was sent to LDS by load_tile's DMA, which hipcc does NOT order against the LDS reads below (checked
            // in the ISA: no vmcnt in front of the first ds_read).  vmcnt counts in issue order, and behind this tile's DMA pieces
            // were issued its 24 X loads and, except for the launch's first tile, the previous tile's H stores: "at most 24
            // outstanding" therefore means the DMA pieces (and the first X loads, issued a whole tile ago) have landed, while
            // the stores of the tile just finished stay in flight.
            static_assert(VT * 4 == 24, "the X tile is 24 loads per lane");
            if (x_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (uniform; its DMA pieces are the youngest loads)
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            x_first = false;
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int FKB = 1 + 7 * (HALF == 2 ? 1 : 0);
        FK_TICK(FKB + 0);

        // (DO_STATS) the x-only constants of this lane's KL terms (tile_kl); they land under the P product
        double cv[4] = {0.0, 0.0, 0.0, 0.0};
        if (DO_STATS && (TG || p.KLpart != nullptr)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[r] = p.xlx[(n0 + q + 4 * r) * 16 + c16];
        }
        // ---- P = Ht . W   (A = H[n=c16][k=4s+q], B = W[k=4s+q][v=16vt+c16])
        d4 pr[VT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) pr[vt] = (d4){0, 0, 0, 0};
        if (RGIVEN) {  // the tile of "X" is the ratio already
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[vt][r] = x[vt][r];
        } else {
            const double* ha = Hl + c16 * LS + q;
            const double* wb = Wl + q * WS + c16;
            double a[2], b[2][VT];
            a[0] = ha[0];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) b[0][vt] = wb[16 * vt];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                // the next k-step's operand reads are spread between this k-step's MFMAs (one LDS instruction behind
                // each): issued in a clump they queue up behind the other waves' clumps on the CU's one LDS pipe and
                // hold back the in-order MFMA behind them (tools/phase_probe.hip: 29.7 -> 28.3 ns per MFMA)
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < KS) {
                    a[(s + 1) & 1] = ha[4 * (s + 1)];
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) b[(s + 1) & 1][vt] = wb[4 * (s + 1) * WS + 16 * vt];
                }
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) pr[vt] = mfma(a[s & 1], b[s & 1][vt], pr[vt]);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        FK_TICK(FKB + 1);
        // (joint step with the objective: the KL terms first, before the G operands below take their registers)
        constexpr bool JKL = TG && TU && DO_STATS;  // (never with MVJ: a half has one of the two)
        if (JKL) {
            const double none[4] = {0.0, 0.0, 0.0, 0.0};
            klacc += tile_kl<false, (KR >= 3 ? 2 : 3)>(x, pr, none, cv, ltab, n0, N, V, q, c16);
        }
        // G-phase A operands (H^T): issue the LDS reads now, they land under the divisions
        double ga[4][KT];
        if (TG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double* ha = Hl + (4 * r + q) * LS + c16;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) ga[r][kt] = ha[16 * kt];
            }
        }

        // unweighted KL(X || P) of this tile from P before the division: always with the numerator pass (f0 of the MvNMF
        // line search), with the update_H pass only when asked (KLpart != null: a speculative pass evaluates the trial
        // it starts from, which saves the separate forward pass)
        if (DO_STATS && !JKL && (TG || p.KLpart != nullptr)) {
            const double none[4] = {0.0, 0.0, 0.0, 0.0};
            (HALF == 2 ? klacc_b : klacc) += tile_kl<false>(x, pr, none, cv, ltab, n0, N, V, q, c16);
        }
        FK_TICK(FKB + 2);
        // ---- R = X / P in place (rows n = q + 4r, columns v = 16vt + c16); pads give 0 / P = 0
        // (div_path's sequence, six quotients at a time and stage by stage: independent chains next to each other)
#pragma unroll
        for (int r = 0; r < (RGIVEN ? 0 : 4); ++r) {
            double rc[VT], t0[VT], t1[VT];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) rc[vt] = __builtin_amdgcn_rcp(pr[vt][r]);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) t0[vt] = __builtin_fma(-pr[vt][r], rc[vt], 1.0);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) rc[vt] = __builtin_fma(rc[vt], t0[vt], rc[vt]);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) t1[vt] = x[vt][r] * rc[vt];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) t0[vt] = __builtin_fma(-pr[vt][r], t1[vt], x[vt][r]);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) pr[vt][r] = __builtin_fma(t0[vt], rc[vt], t1[vt]);
        }
        // all 24 divisions before the G phase: left to itself hipcc sinks each division in front of the three MFMAs
        // that consume it, and a single division is a chain of six dependent fp64 instructions whose latencies are
        // then exposed 24 times per tile (the MFMAs share the pipe, so nothing is gained by interleaving them).
        // One empty asm that takes all 24 quotients in and hands them out again pins that.
        asm volatile("" : "+v"(pr[0][0]), "+v"(pr[0][1]), "+v"(pr[0][2]), "+v"(pr[0][3]), "+v"(pr[1][0]), "+v"(pr[1][1]), "+v"(pr[1][2]),
                     "+v"(pr[1][3]), "+v"(pr[2][0]), "+v"(pr[2][1]), "+v"(pr[2][2]), "+v"(pr[2][3]), "+v"(pr[3][0]), "+v"(pr[3][1]));
        asm volatile("" : "+v"(pr[3][2]), "+v"(pr[3][3]), "+v"(pr[4][0]), "+v"(pr[4][1]), "+v"(pr[4][2]), "+v"(pr[4][3]), "+v"(pr[5][0]),
                     "+v"(pr[5][1]), "+v"(pr[5][2]), "+v"(pr[5][3]));

        FK_TICK(FKB + 3);
        // prefetch the next tile: X and the staging registers are free from here on, and the loads
        // get the G and U phases to land
        if (HALF != 1 && tile + tstride < nfull) load_tile(tile + tstride);  // (MVJ: X serves the second half too)
        else if (HALF != 1 && coop_here) coop_prefetch();

        if (TU) {
            // ---- transpose R through LDS: write accumulator layout, read A-operand layout
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Rl[(q + 4 * r) * RS + 16 * vt + c16] = pr[vt][r];
        }

        FK_TICK(FKB + 4);
        // ---- G += (w_kl . Ht)^T . R   (A = H[n=4r+q][k=16kt+c16], B = register r of R)
        if (TG) {
            if (wkl) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double wk = wgt[2 * (q + 4 * r)];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) ga[r][kt] *= wk;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) mfma_agpr(g[kt][vt], ga[r][kt], pr[vt][r]);
            if (KR > 0) {
                // remainder rows: G[KB+j][v] += sum over this lane's rows n = q+4r of H[n][KB+j] * R[n][v]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double wk = wkl ? wgt[2 * (q + 4 * r)] : 1.0;
#pragma unroll
                    for (int j = 0; j < KR; ++j) {
                        double hv = Hl[(4 * r + q) * LS + KB + j];
                        if (wkl) hv *= wk;
#pragma unroll
                        for (int vt = 0; vt < VT; ++vt) grem[j][vt] = __builtin_fma(hv, pr[vt][r], grem[j][vt]);
                    }
                }
            }
        }
        // remainder columns of U: per-lane partial dot products over this lane's 6 feature columns
        double urem[NVP > 0 ? NVP : 1];
        if (TU && KR > 0) {
#pragma unroll
            for (int i = 0; i < NVP; ++i) urem[i] = 0.0;
            // (feature tile outermost: the 4 KR accumulation chains advance side by side; each still adds its six terms
            // in the order vt = 0..5)
            double wj[KR > 0 ? KR : 1][VT];
#pragma unroll
            for (int j = 0; j < KR; ++j)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) wj[j][vt] = Wl[(KB + j) * WS + 16 * vt + c16];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int j = 0; j < KR; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) urem[4 * j + r] = __builtin_fma(pr[vt][r], wj[j][vt], urem[4 * j + r]);
        }
        __builtin_amdgcn_wave_barrier();
        FK_TICK(FKB + 5);
#ifdef SALNMF_DEV_PROFILE
        if (HALF != 1) ++ntl;
#endif

        if (TU) {
            // ---- U = R . W^T   (A = R[n=c16][v=4s+q], B = W[k=16kt+c16][v=4s+q])
            d4 u[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) u[kt] = (d4){0, 0, 0, 0};
            // H of this tile in the U accumulator layout; read now, consumed by the epilogue
            double hcur[4][KT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hcur[r][kt] = Hl[(q + 4 * r) * LS + 16 * kt + c16];
            // Reduce-scatter of the NVP remainder partials over the 16 lanes that share q: every
            // stage halves the live values; afterwards lane c16 holds the complete
            // U[n = q+4r][k = KB+j] for (j, r) = (rs_idx >> 2, rs_idx & 3).
            auto rs_stage = [&](auto mtag, auto livetag) __attribute__((always_inline)) {
                constexpr int M = decltype(mtag)::value, LIVE = decltype(livetag)::value;
                if (LIVE > 1) {
                    constexpr int half = LIVE / 2;
                    const bool upper = (c16 & M) != 0;
#pragma unroll
                    for (int i = 0; i < half; ++i) {
                        double send = upper ? urem[i] : urem[i + half];
                        double keep = upper ? urem[i + half] : urem[i];
                        urem[i] = keep + xor16<M>(send);
                    }
                } else {
                    urem[0] += xor16<M>(urem[0]);
                }
            };
            // value index owned by this lane after the stages: the c16 bits consumed by halving stages
            const int rs_idx = NVP == 16 ? c16 : (NVP == 8 ? (c16 >> 1) : (c16 >> 2));
            const double* ra = Rl + c16 * RS + q;
            const double* wb = Wl + c16 * WS + q;
            double a[2], b[2][KT];
            a[0] = ra[0];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b[0][kt] = wb[16 * kt * WS];
#pragma unroll
            for (int s = 0; s < VSTEPS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < VSTEPS) {
                    a[(s + 1) & 1] = ra[4 * (s + 1)];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) b[(s + 1) & 1][kt] = wb[16 * kt * WS + 4 * (s + 1)];
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) u[kt] = mfma(a[s & 1], b[s & 1][kt], u[kt]);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {  // reads (and the stage's VALU work) spread between the MFMAs, as in the P phase
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KR > 0) { using std::integral_constant; rs_stage(integral_constant<int, 8>{}, integral_constant<int, NVP>{}); rs_stage(integral_constant<int, 4>{}, integral_constant<int, (NVP / 2 > 1 ? NVP / 2 : 1)>{}); rs_stage(integral_constant<int, 2>{}, integral_constant<int, (NVP / 4 > 1 ? NVP / 4 : 1)>{}); rs_stage(integral_constant<int, 1>{}, integral_constant<int, (NVP / 8 > 1 ? NVP / 8 : 1)>{}); }
            if (BLOCKED) {
                // this block's share of U joins the earlier blocks' (order: block 0 + 1 + ...); all but the last block
                // leave the running sum in Uacc and do not touch H
                double* ua = p.Uacc + (n0 + q) * KP + c16;
                if (p.ublock != 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt) u[kt][r] += ua[4 * r * KP + 16 * kt];
                }
                const int jr = rs_idx >> 2, rr = rs_idx & 3;
                const bool rem_owner = KR > 0 && jr < KR && (c16 & (NVP == 16 ? 0 : (NVP == 8 ? 1 : 3))) == 0;
                double* uar = p.Uacc + (n0 + q + 4 * rr) * KP + KB + jr;
                if (KR > 0 && rem_owner && p.ublock != 1) urem[0] += *uar;
                if (p.ublock != 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt) ua[4 * r * KP + 16 * kt] = u[kt][r];
                    if (KR > 0 && rem_owner) *uar = urem[0];
                    FK_TICK(FKB + 6);
                    return;  // (of the tile lambda)
                }
            }
            // ---- H update (_utils_klnmf.py:343-361), rows n = q+4r, columns k = 16kt+c16.
            // Unmasked: pad rows / columns just receive finite filler.  Non-temporal stores: 51 MB of H
            // per launch would otherwise sit dirty in L2 and be flushed at the kernel boundary
            // (-1.5 % on the fused + tail pair, tools/ab_bench.hip).
            double* hdst = p.Hout + (n0 + q) * KP + c16;
            if (wlh == nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const double hn = clip_lo(hcur[r][kt] * u[kt][r], p.hfloor);
                        __builtin_nontemporal_store(hn, &hdst[4 * r * KP + 16 * kt]);
                        if (HALF == 1) Hl[(q + 4 * r) * LS + 16 * kt + c16] = hn;  // (MVJ) the numerator half reads H' from here
                        if (MVU) hsum[kt] += (n0 + q + 4 * r < N) ? hn : 0.0;
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t n = n0 + q + 4 * r;
                    const double wl = wgt[2 * (q + 4 * r) + 1];
                    double wk2 = 1.0;
                    if (wkl) { double w = wgt[2 * (q + 4 * r)]; wk2 = w * w; }
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        double inter = 4.0 * hcur[r][kt] * u[kt][r];
                        if (wkl) inter *= wk2;
                        double disc = 0.25 * wl * wl + inter;
                        double t = wl / 2 - sqrt(disc);
                        double hn = 0.25 * (t * t);
                        if (wkl) hn /= wk2;
                        hn = clip_lo(hn, kEps);
                        __builtin_nontemporal_store(hn, &hdst[4 * r * KP + 16 * kt]);
                        if (MVU) hsum[kt] += (n < N) ? hn : 0.0;
                    }
                }
            }
            if (KR > 0) {
                const int j = rs_idx >> 2, r = rs_idx & 3;
                // one owner lane per (j, r): the lanes whose untouched low bits of c16 are zero
                const int lowmask = NVP == 16 ? 0 : (NVP == 8 ? 1 : 3);
                if (j < KR && (c16 & lowmask) == 0) {
                    const int64_t n = n0 + q + 4 * r;
                    const double h = Hl[(q + 4 * r) * LS + KB + j];
                    double hn;
                    if (wlh == nullptr) {
                        hn = h * urem[0];
                    } else {
                        const double wl = wgt[2 * (q + 4 * r) + 1];
                        double wk2 = 1.0;
                        if (wkl) { double w = wgt[2 * (q + 4 * r)]; wk2 = w * w; }
                        double inter = 4.0 * h * urem[0];
                        if (wkl) inter *= wk2;
                        double disc = 0.25 * wl * wl + inter;
                        double t = wl / 2 - sqrt(disc);
                        hn = 0.25 * (t * t);
                        if (wkl) hn /= wk2;
                    }
                    hn = clip_lo(hn, p.hfloor);
                    p.Hout[n * KP + KB + j] = hn;
                    if (HALF == 1) Hl[(q + 4 * r) * LS + KB + j] = hn;
                    if (MVU) hsum_rem += (n < N) ? hn : 0.0;
                }
            }
        }
        FK_TICK(FKB + 6);
    };

    if (DO_G) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) g[kt][vt] = (d4){0, 0, 0, 0};
        if (KR > 0) {
#pragma unroll
            for (int j = 0; j < KR; ++j)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) grem[j][vt] = 0.0;
        }
    }
    // ---- one tile by the four waves of a workgroup (leftover round, see above).
    //   phase A  wave w takes the feature tiles {0,1} / {2,3} / {4} / {5}: P and R = X / P for those columns, their
    //            share of G (into its own accumulators: G is summed over the waves anyway) and of the remainder rows;
    //            R goes to a tile in LDS that all waves share (wave 0's), as does the staged H tile
    //   phase B  wave kt < KT computes U[:, 16 kt .. 16 kt + 15] = R W^T over all 96 features (the same MFMA chain as in
    //            process_tile: same bits) and updates those columns of H; the next wave takes the KR remainder columns
    //            (one lane per (sample, column), sequential dot product over the features)
    // Per entry the arithmetic is the reference's; what differs from process_tile is only which wave's accumulator a
    // contribution to G lands in and the summation order of the remainder columns of U (rounding level).
    auto process_tile_coop = [&](int64_t ctile) __attribute__((always_inline)) {
        const int64_t n0 = ctile * 16;
        // (indices derived from an opaque copy of the thread index: nothing of this once-per-launch section may be
        // hoisted above the tile loop, where it would cost registers)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
        // wave 0's H tile and R tile serve as the shared ones; a prefetched H tile (coop_prefetch) lies in the buffer wave 0's
        // next tile would have gone to: every wave has done the same number of tiles, so every wave knows which one
        const bool pre = HDMA && coop_fetched;  // (uniform over the workgroup)
        double* Hs = lds + WROWS * WS + ((pre && ((nfull / tstride) & 1)) ? G_::HL + G_::RL : 0);
        double* cslab = lds + CSLAB;  // the LDS regions of waves 1..3 are free meanwhile
        double* Rs = lds + WROWS * WS + G_::HL;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // loads first (they fly while the slower waves of the workgroup arrive): the H tile, 16 bytes per thread and
        // round, and this wave's columns of X
        constexpr int HR = (16 * KP + 2 * BLOCK - 1) / (2 * BLOCK);
        d2 hst[HR];
        const int vt0 = wv < 2 ? 2 * wv : wv + 2, nvt = wv < 2 ? 2 : 1;  // {0,1} {2,3} {4} {5}
        double xx[2][4];
        if (pre) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xx[i][r] = x[i][r];
        } else {
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const int e = 2 * tid + 2 * BLOCK * j;
                hst[j] = (e < 16 * KP) ? *reinterpret_cast<const d2*>(p.H + n0 * KP + e) : (d2){0, 0};
            }
            const double* xsrc = p.X + (n0 + q) * VMAX + c16;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xx[i][r] = (i < nvt) ? xsrc[4 * r * VMAX + 16 * (vt0 + i)] : 0.0;
        }
        // (WTS) the tile's 16 weight pairs, through wave 0's weight slots in LDS
        double* wg0 = ltab + (DO_STATS ? LOGTAB_DOUBLES : 0);
        d2 wco = (d2){1.0, 0.0};
        if (WTS && tid < 16) {
            wco[0] = p.wkl_eff[n0 + tid];
            wco[1] = p.wlh_eff[n0 + tid];
        }
        // every wave has left its own last tile: wave 0's LDS regions are free.  (The DMA pieces of a prefetched H tile are
        // invisible to hipcc: wave 0 waits for them by hand before it joins the barrier.)
        if (pre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (WTS && tid < 16) *reinterpret_cast<d2*>(wg0 + 2 * tid) = wco;
        if (!pre) {
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const int e = 2 * tid + 2 * BLOCK * j;
                if (e < 16 * KP) {
                    const int row = e / KP, col = e - row * KP;
                    *reinterpret_cast<d2*>(Hs + row * LS + col) = hst[j];
                }
            }
        }
        if (WTS || !pre) __syncthreads();
        FK_TICK(9);   // loads + first barrier (the slower waves of the workgroup arrive) + staging + second barrier
        // ---- phase A, one feature tile at a time (compile-time tile index: the accumulators are registers)
        auto phase_a = [&](auto vtag, const double (&xv)[4]) __attribute__((always_inline)) {
            constexpr int VTI = decltype(vtag)::value;
            d4 pp = (d4){0, 0, 0, 0};
            const double* ha = Hs + c16 * LS + q;
            const double* wb = Wl + q * WS + 16 * VTI + c16;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) pp = mfma(ha[4 * s2], wb[4 * s2 * WS], pp);
            if (DO_STATS) {
                // this feature tile's share of the unweighted KL partial (tile_kl's masked form, four entries per lane)
                bool valid[4], ok = true;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    valid[r] = (n0 + q + 4 * r < N) && (16 * VTI + c16 < V);
                    ok &= !valid[r] || log_pos_ok(pp[r]);
                }
                // the lane column's x-only constants cover its six feature tiles, which the cooperative tile spreads over
                // the waves: the wave of feature tile 0 adds them (the cancellation of these at most gridDim.x tiles then
                // happens in the workgroup's sum instead of in the lane)
                if (VTI == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) klacc += p.xlx[(n0 + q + 4 * r) * 16 + c16];
                }
                if (__all(ok)) {
                    double ps[4], lp[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ps[r] = valid[r] ? pp[r] : 1.0;
                    log_pos_n<4>(ps, ltab, lp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) klacc += valid[r] ? __builtin_fma(-xv[r], lp[r], ps[r]) : 0.0;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (valid[r]) klacc += kl_term_p(xv[r], pp[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pp[r] = div_path(xv[r], pp[r]);
                Rs[(q + 4 * r) * RS + 16 * VTI + c16] = pp[r];
            }
            // this tile's contribution to G is parked in LDS (accumulator layout, indexed like the epilogue's tiles); the
            // epilogue's owner waves pick it up before they reuse LDS and add it after the four waves' accumulators.  The
            // accumulators themselves are not touched outside the tile loop (doing so changes hipcc's register assignment
            // inside the loop and costs ~2 % there).
            d4 gc[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) gc[kt] = (d4){0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    double ha = Hs[(4 * r + q) * LS + 16 * kt + c16];
                    if (WTS && wkl) ha *= wg0[2 * (4 * r + q)];
                    gc[kt] = mfma(ha, pp[r], gc[kt]);
                }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) cslab[((kt * VT + VTI) * 4 + r2) * 64 + lane] = gc[kt][r2];
            if (KR > 0) {
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    double t = 0.0;  // this feature column's sum over the 16 samples: 4 rows per lane, then the 4 q groups
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double hv = Hs[(4 * r + q) * LS + KB + j];
                        if (WTS && wkl) hv *= wg0[2 * (4 * r + q)];
                        t = __builtin_fma(hv, pp[r], t);
                    }
                    t = rows_sum(t);
                    if (q == 0) cslab[CO_::COOP_REM + j * VMAX + 16 * VTI + c16] = t;
                }
            }
        };
        {
            using std::integral_constant;
            if (wv == 0) { phase_a(integral_constant<int, 0>{}, xx[0]); phase_a(integral_constant<int, 1>{}, xx[1]); }
            else if (wv == 1) { phase_a(integral_constant<int, 2>{}, xx[0]); phase_a(integral_constant<int, 3>{}, xx[1]); }
            else if (wv == 2) phase_a(integral_constant<int, 4>{}, xx[0]);
            else phase_a(integral_constant<int, 5>{}, xx[0]);
        }
        __syncthreads();  // the ratio tile is complete
        FK_TICK(10);  // phase A + its barrier
        // ---- phase B
        if (wv < KT) {
            const int kt = wv;
            d4 u = (d4){0, 0, 0, 0};
            const double* ra = Rs + c16 * RS + q;
            const double* wb = Wl + (16 * kt + c16) * WS + q;
#pragma unroll
            for (int s2 = 0; s2 < VSTEPS; ++s2) u = mfma(ra[4 * s2], wb[4 * s2], u);
            double* hdst = p.Hout + (n0 + q) * KP + 16 * kt + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double h = Hs[(q + 4 * r) * LS + 16 * kt + c16];
                double hn = h * u[r];
                if (WTS && wlh) hn = lhalf_update(h, u[r], wg0[2 * (q + 4 * r) + 1], wkl ? wg0[2 * (q + 4 * r)] : 1.0, wkl != nullptr);
                __builtin_nontemporal_store(clip_lo(hn, WTS && wlh ? kEps : p.hfloor), &hdst[4 * r * KP]);
            }
        } else if (KR > 0 && wv == KT) {
            const int n = lane & 15, j = lane >> 4;
            if (j < KR) {
                const double* rr = Rs + n * RS;
                const double* wr = Wl + (KB + j) * WS;
                double dot = 0.0;
                for (int v = 0; v < VMAX; ++v) dot = __builtin_fma(rr[v], wr[v], dot);
                const double h = Hs[n * LS + KB + j];
                double hn = h * dot;
                if (WTS && wlh) hn = lhalf_update(h, dot, wg0[2 * n + 1], wkl ? wg0[2 * n] : 1.0, wkl != nullptr);
                p.Hout[(n0 + n) * KP + KB + j] = clip_lo(hn, WTS && wlh ? kEps : p.hfloor);
            }
        }
    };

    tile = (int64_t)blockIdx.x * WAVES + wave;
    if (PERSIST) {
        // the first tile (its H rows were written by this very wave in the previous step) flies during the wait
        if (tile < nfull) load_tile(tile);
        if (step > 0 && !persist_wait_W(p.sync, p.abort_host, (unsigned)step * (unsigned)K, lds, tid)) return;
        stage_W<WROWS, true>(Wl, p.Wmut, K, V, V, tid);  // sc1 loads: rows published by other workgroups
        __syncthreads();
    } else {
        // (the first tile's loads issued ahead of the staging instead: 69.7 -> 71.9 us per step at c2 -- the prologue's
        // registers then overlap the tile's, 234 -> 256 VGPRs + 24 spill copies; profiles/r03/ab_step_variants.txt)
        // Round 5: for the common layout W travels by LDS-DMA (no registers, 13 instructions per wave at K = 50 against 24
        // loads + 24 LDS stores): 69.5 -> 69.0 us per step at c2, 79.9 -> 79.4 at c3's shard (alternating blocks on one
        // engine, profiles/r05/w_dma.md).  The barrier is the bare instruction: __syncthreads() carries a workgroup fence
        // that hipcc lowers to s_waitcnt vmcnt(0) anyway; here the DMA pieces (counted by vmcnt like any load) and the zero
        // rows' ds_writes are waited for by hand, and LDS is coherent within the CU.
        // Measured and dropped in the same A/B: the first tile's loads queued BEHIND the DMA (+2.3 us at c2 whether the wait
        // is for everything or a counted vmcnt(24) that leaves X in flight: the waves' 20 KB of cold loads each then sit in
        // the CU's memory pipeline in front of the other waves' DMA pieces, and the barrier waits for the last of those);
        // only the H tile behind the DMA (+0.5 us at K = 50, -1.0 at K = 30).
        if (p.wdma && V == VMAX && p.ldw == VMAX && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0) {  // (uniform)
            stage_W_dma_issue<WROWS>(Wl, p.W, K, tid);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");  // (no LDS read of W is hoisted above the barrier)
            if (tile < nfull) load_tile(tile);
        } else {
            stage_W<WROWS>(Wl, p.W, K, V, p.ldw, tid);
            __syncthreads();
            if (tile < nfull) {
                if (HDMA) {
                    // X first, then the DMA pieces of H, on this (rare) path: were the two prologues to end in the same
                    // sequence, hipcc would merge their tails and put an s_waitcnt vmcnt(0) between the DMA pieces and the X
                    // loads of BOTH (seen in the ISA): one more memory round trip in front of every launch's first tile
                    load_tile_X(tile);
                    load_tile_H(tile);
                    x_first = true;
                } else {
                    load_tile(tile);
                }
            }
        }
    }
    FK_TICK(0);
    for (; tile < nfull; tile += tstride) {
        using std::integral_constant;
        if constexpr (MVJ) {
            process_half(tile, integral_constant<int, 1>{});
            process_half(tile, integral_constant<int, 2>{});
        } else {
            process_half(tile, integral_constant<int, 0>{});
        }
        if (HDMA) {  // the tile prefetched meanwhile becomes the current one
            double* t = Hl;
            Hl = Hnx;
            Hnx = t;
        }
    }
    FK_TICK(8);  // (non-MVJ kernels: sections 8..12 time the cooperative tile -- 8 is always ~0: the loop's own ticks precede it)
    if (COOP && coop && (int64_t)blockIdx.x < nleft) process_tile_coop(nfull + blockIdx.x);
    FK_TICK(12);

    // ---- workgroup reductions, fixed order (deterministic)
    __syncthreads();  // every wave is done with the LDS copy of W
    FK_TICK(11);  // (non-MVJ: the wait for the workgroup's slowest wave)
    if (DO_G) {
        // the asm MFMAs are opaque to hipcc: drain the matrix pipe before any VALU read of g
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) asm volatile("" : "+a"(g[kt][vt]));  // orders the reads below after the drain
        // Cross-wave sum through LDS (all of it is free now), EpiGeo's scheme.  Round 5: every wave parks ALL tiles of a
        // round -- compile-time register indices and no branch, where the 3/4-parking form before had a scalar branch per
        // tile and wave -- and the owners read all four contributions from LDS.  Same sums in the same order (wave
        // 0 + 1 + 2 + 3, then the cooperative tile's share): same bits.  profiles/r05/epilogue.md.
        constexpr int ROUNDS = CO_::ROUNDS, VTR = CO_::VTR, NT = CO_::NT, MAXI = CO_::MAXI;
        double* remL = lds + CO_::PARK;  // [WAVES][KR][VMAX]
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // a cooperative leftover tile left its numerator contribution in LDS: the owners take theirs into registers
        // before the parking below reuses that memory (clamped addresses, no branch per element)
        const bool coopwg = COOP && coop && (int64_t)blockIdx.x < nleft;
        double cc[ROUNDS][MAXI][4];
        double crem[2] = {0.0, 0.0};
        if (COOP && coopwg) {
            const double* cl = lds + CSLAB;
#pragma unroll
            for (int half = 0; half < ROUNDS; ++half)
#pragma unroll
                for (int i = 0; i < MAXI; ++i) {
                    const int t = wv + WAVES * i < NT ? wv + WAVES * i : NT - 1;  // (a slot beyond the wave's last tile is not used)
                    const int kt = t / VTR, vt = half * VTR + (t - kt * VTR);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cc[half][i][r] = cl[((kt * VT + vt) * 4 + r) * 64 + lane];
                }
            if (KR > 0) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int i = tid + it * BLOCK;
                    crem[it] = cl[CO_::COOP_REM + (i < KR * VMAX ? i : 0)];
                }
            }
            __syncthreads();
        }
        double* out = p.Gpart + (int64_t)blockIdx.x * K * VMAX;
#pragma unroll
        for (int half = 0; half < ROUNDS; ++half) {
            double* mine = lds + (size_t)wv * NT * 256 + lane;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int h = 0; h < VTR; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((kt * VTR + h) * 4 + r) * 64] = g[kt][half * VTR + h][r];
            if (KR > 0 && half == 0) {
#pragma unroll
                for (int j = 0; j < KR; ++j)
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) {
                        const double t = rows_sum(grem[j][vt]);  // the four q groups (lanes l, l^16, l^32, l^48)
                        if (q == 0) remL[(wv * KR + j) * VMAX + 16 * vt + c16] = t;
                    }
            }
            __syncthreads();
            FK_TICK(13);  // (non-MVJ: parking of the accumulator tiles + barrier)
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int t = wv + WAVES * i;  // (uniform) the i-th tile this wave owns
                if (t < NT) {
                    const int kt = t / VTR, vt = half * VTR + (t - kt * VTR);
                    const double* from = lds + (size_t)t * 256 + lane;
                    double acc[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[r] = from[r * 64];
#pragma unroll
                        for (int w = 1; w < WAVES; ++w) acc[r] += from[(size_t)w * NT * 256 + r * 64];
                    }
                    if (COOP && coopwg) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] += cc[half][i][r];
                    }
                    const int v = 16 * vt + c16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 16 * kt + q + 4 * r;
                        if (k < K && v < V) st_shared<PERSIST>(&out[k * VMAX + v], acc[r]);
                    }
                }
            }
            if (KR > 0 && half == 0) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int i = tid + it * BLOCK;
                    if (i < KR * VMAX) {
                        const int j = i / VMAX, v = i - j * VMAX;
                        double t = ((remL[j * VMAX + v] + remL[(KR + j) * VMAX + v]) + remL[(2 * KR + j) * VMAX + v]) + remL[(3 * KR + j) * VMAX + v];
                        if (COOP && coopwg) t += crem[it];
                        if (v < V) st_shared<PERSIST>(&out[(KB + j) * VMAX + v], t);
                    }
                }
            }
            if (half + 1 < ROUNDS) __syncthreads();
        }
        FK_TICK(14);  // (non-MVJ: the owners' sums and the slab stores issued)
    }
    if (PERSIST) {
        if (!persist_publish_and_tail(p.sync, p.abort_host, p.Gpart, p.G, p.Wmut, K, V, p.n_given, lds, step, tid)) return;
    }
    if (MVU) {
        __syncthreads();
        // column k = 16kt + c16 (or KB + j) of the updated H: sum the 4 waves x 4 q-groups in fixed order
        double* S = lds;  // [16*KT + 16][16]: row = column index k, 16 slots = (wave, q)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) S[(16 * kt + c16) * 16 + wave * 4 + q] = hsum[kt];
        if (KR > 0) {
            // owner lanes: value index idx = (j, r); sum the r's of one j inside the q-group first
            const int idx = NVP == 16 ? c16 : (NVP == 8 ? (c16 >> 1) : (c16 >> 2));
            const int lowmask = NVP == 16 ? 0 : (NVP == 8 ? 1 : 3);
            const int j = idx >> 2;
            for (int jj = 0; jj < KR; ++jj) {
                double v = (j == jj && (c16 & lowmask) == 0) ? hsum_rem : 0.0;
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
                if (c16 == 0) S[(KB + jj) * 16 + wave * 4 + q] = v;
            }
        }
        __syncthreads();
        if (tid < K) {
            double t = 0.0;
            for (int i = 0; i < 16; ++i) t += S[tid * 16 + i];
            p.Hsumpart[(int64_t)blockIdx.x * K + tid] = t;
        }
    }
    if (MVJ) {  // the numerator half's KL partial (f0 of the next step), reduced by the tail like a numerator pass's
        __syncthreads();
        double* Kb = lds;
        Kb[tid] = klacc_b;
        __syncthreads();
        for (int h = BLOCK / 2; h > 0; h >>= 1) {
            if (tid < h) Kb[tid] += Kb[tid + h];
            __syncthreads();
        }
        if (tid == 0) p.KLpartB[blockIdx.x] = Kb[0];
    }
    if (DO_STATS && ((DO_G && !MVJ) || p.KLpart != nullptr)) {
        __syncthreads();
        double* Ks = lds;  // [BLOCK], fixed binary tree
        Ks[tid] = klacc;
        __syncthreads();
        for (int h = BLOCK / 2; h > 0; h >>= 1) {
            if (tid < h) Ks[tid] += Ks[tid + h];
            __syncthreads();
        }
        if (!(MVU && p.kl_out != nullptr)) {
            if (tid == 0) p.KLpart[blockIdx.x] = Ks[0];
        } else {
            // in-launch final sum (cdna_hip_programming.md, guideline 16, counter form): the partial is published by a
            // write-through store of ONE lane, which drains it and draws a ticket; whoever draws the last ticket reads
            // all partials with sc1 loads and adds them in the order of sum_partials_kernel (same bits)
            if (tid == 0) {
                __hip_atomic_store((gdouble*)(p.KLpart + blockIdx.x), Ks[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned ticket = __hip_atomic_fetch_add((gsync_t*)p.kl_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                Ks[1] = (ticket == (unsigned)nwg - 1u) ? 1.0 : 0.0;
            }
            __syncthreads();
            const bool last = Ks[1] != 0.0;  // (uniform)
            __syncthreads();
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the loads below the ticket)
                double sum = 0.0;
                for (int i = tid; i < nwg; i += BLOCK) sum += ld_shared<true>(p.KLpart + i);
                Ks[tid] = sum;
                __syncthreads();
                for (int h = BLOCK / 2; h > 0; h >>= 1) {
                    if (tid < h) Ks[tid] += Ks[tid + h];
                    __syncthreads();
                }
                if (tid == 0) {
                    p.kl_out[0] = Ks[0];
                    __hip_atomic_store((gsync_t*)p.kl_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    }  // step
    FK_TICK(15);
#ifdef SALNMF_DEV_PROFILE
    if (p.prof != nullptr && (threadIdx.x & 63) == 0) {  // (one row per wave of the grid: no atomics, launches are serialised)
        unsigned long long* row = p.prof + ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * (FK_NSEC + 1);
        for (int i = 0; i < FK_NSEC; ++i) row[i] += tks[i];
        row[FK_NSEC] += ntl;
    }
#endif
#undef FK_TICK
}

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
