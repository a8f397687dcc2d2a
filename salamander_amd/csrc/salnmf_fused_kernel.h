// The fused update pass of the KL-NMF path (fused_kernel): P = H W, R = X / P, G += H^T R, U = R W^T, H <- clip(H * U) per tile of
// 16 samples, one wave per tile.  Layout, lane maps and the helpers it uses: salnmf_kernels.h (which includes this file).
#pragma once
#include "salnmf_kernels.h"

namespace salnmf {

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
// BLOCKED (n_features > 96; the update_H pass alone or, round 5, with the block's numerator G in the same pass: every block's
// numerator is formed from the OLD H, which only the last block's pass rewrites): this launch covers ONE 96-feature block of X and W; the product
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
    static_assert(!BLOCKED || (DO_U && !DO_STATS && WTS && !PERSIST), "feature blocks: the weighted-capable update_H pass, alone or with the block's numerator");
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
    constexpr bool COOP = DO_G && DO_U && !RGIVEN && !MVJ && !BLOCKED;  // (with per-sample weights too: process_tile_coop honours them)
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
        // (BLOCKED) the earlier blocks' running sum of U for this tile: requested now, it is the U product's starting value a
        // whole P / G phase later -- read after the product, as rounds 4 had it, every tile paid a memory round trip that
        // nothing hid (one wave per SIMD).  U = (U_0 + U_1 + ...) with block b's own products added onto the running sum.
        d4 uacc0[BLOCKED ? KT : 1];
        if (BLOCKED && TU) {
            const double* ua = p.Uacc + (n0 + q) * KP + c16;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) uacc0[kt][r] = p.ublock != 1 ? ua[4 * r * KP + 16 * kt] : 0.0;
        }
        double uarem0 = 0.0;  // (the same for the remainder column this lane owns after the reduce-scatter of the U phase)
        if (BLOCKED && TU && KR > 0 && p.ublock != 1) {
            const int ridx = NVP == 16 ? c16 : (NVP == 8 ? (c16 >> 1) : (c16 >> 2));
            const int jr0 = ridx >> 2, rr0 = ridx & 3;
            if (jr0 < KR && (c16 & (NVP == 16 ? 0 : (NVP == 8 ? 1 : 3))) == 0) uarem0 = p.Uacc[(n0 + q + 4 * rr0) * KP + KB + jr0];
        }
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
            for (int kt = 0; kt < KT; ++kt) u[kt] = BLOCKED ? uacc0[kt] : (d4){0, 0, 0, 0};
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
                double* ua = p.Uacc + (n0 + q) * KP + c16;  // (the running sum is in u already: uacc0)
                const int jr = rs_idx >> 2, rr = rs_idx & 3;
                const bool rem_owner = KR > 0 && jr < KR && (c16 & (NVP == 16 ? 0 : (NVP == 8 ? 1 : 3))) == 0;
                double* uar = p.Uacc + (n0 + q + 4 * rr) * KP + KB + jr;
                if (KR > 0 && rem_owner) urem[0] += uarem0;  // (0 for the first block)
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

}  // namespace salnmf
