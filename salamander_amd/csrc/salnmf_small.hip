// Small cohorts: the joint KL step (update_WH, _utils_klnmf.py:281-347, unweighted) for up to 64 tiles (1 024 samples),
// n_signatures <= 16, as ONE workgroup that runs n_steps steps per launch.
//
// At the size of the reference's own data set (data/pcawg_breast_sbs.csv: 192 samples, 5 signatures) the per-step path
// is two launches per step -- fused_kernel on 3 workgroups + tail_kernel -- each shorter than its dispatch: 13.5 us per
// iteration, all of it launch latency, W / slab round trips through global memory and cold first loads.
// The grid-wide persistent kernel of round 3 replaced the launches by grid-wide hand-offs and was slower at every size.
// A cohort this small needs no grid: 12 tiles are 12 wavefronts.  One workgroup of up to 16 waves keeps W in LDS and --
// when every wave has one tile -- X and H of its tile in registers across the steps; every wait is a workgroup barrier.
//
// Same bits as the per-step path.  That path (<= 1 024 tiles: one tile per wave) computes
//   slab_g = ((T_4g + T_4g+1) + T_4g+2) + T_4g+3        fused_kernel's cross-wave sum, T_t = tile t's H^T R from zero
//   part_p = (0 + slab_p) + slab_p+8                    tail_row, first stage
//   G      = (((0 + part_0) + part_1) + ...) + part_7   tail_row, second stage
// and then W' = W * G, row sums in two levels (12 groups of 8 features), normalise, keep given rows, clip.  Here wave w
// of the workgroup is "virtual wave" w % 4 of "virtual workgroup" w / 4 + NG * iteration: the four waves of a group run
// fused_kernel's reduce-scatter among themselves, the owners keep the parts in registers, and the tail below restates
// tail_row's arithmetic for all rows at once.  Per entry the arithmetic of a tile is process_tile's (same MFMA chains,
// same division sequence), so W, H and G come out bit for bit as from salnmf_kl_step's other path
// (tests/test_gpu_small.py compares them).
#define SALNMF_TEMPLATES_ONLY 1
#include "salnmf_launch.h"

namespace salnmf {
namespace {

constexpr int SM_KP = 16, SM_LS = 18;  // H tile in LDS: [16][18] (2 * odd: conflict-free A reads), as Geo<KS <= 4>
constexpr int SM_RSH = 50;             // half of the ratio tile, [16][50]
constexpr int SM_REGION = 1152;        // doubles per wave: H tile + half ratio tile (1 088), or its share of the parked tiles
constexpr int SM_ITEMS = 16 * VMAX;    // (k, v) entries of W / G

// MULTI: a wave has more than one tile (only with NG = 4): the H tiles are read and written every step, and a group's
// slabs fall into two parts.  Otherwise H stays in registers and `part` has one half.
template <int KS, int NG, bool MULTI>
__global__ void __launch_bounds__(256 * NG, 1) small_kl_kernel(SmallParams p) {
    constexpr int NW = 4 * NG, NT = 64 * NW;
    static_assert(KS <= 4, "n_signatures <= 16: one signature tile");
    static_assert(!MULTI || NG == 4, "several tiles per wave only in the full workgroup");
    __shared__ __attribute__((aligned(16))) double lds[16 * WS + NW * SM_REGION];
    double* Wl = lds;                // [16][WS], padded as stage_W leaves it; rewritten by the tail of every step
    double* space = lds + 16 * WS;   // the waves' regions; parked tiles, parts and the tail's scratch reuse them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
    const int grp = wave >> 2, vw = wave & 3;
    double* Hl = space + wave * SM_REGION;
    double* Rh = Hl + 16 * SM_LS;
    const int V = p.V, K = p.K, ntiles = p.ntiles;
    const int tpw = MULTI ? (ntiles + NW - 1) / NW : 1;  // iterations of the tile loop (uniform)
    constexpr bool resident = !MULTI;
    const int nslabs = (ntiles + 3) / 4, nparts = nslabs < 8 ? nslabs : 8;

    if (tid < BLOCK) stage_W<16>(Wl, p.W, K, V, V, tid);
    // H of the wave's tile in the accumulator layout (rows q + 4r, column c16): 4 registers, resident across the steps when
    // the wave has one tile.  X is read again every step (from L2; the loads fly under the P product): sixteen waves
    // share a CU's register file, 128 registers each, and 48 of them for a resident X tile do not fit beside R and T.
    double h[4] = {0.0, 0.0, 0.0, 0.0};
    auto load_h = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = p.H[((int64_t)t * 16 + q + 4 * r) * SM_KP + c16];
    };
    if (resident && wave < ntiles) load_h(wave);
    __syncthreads();

    // accumulator tiles vt = 0..5 of the numerator; virtual wave o owns the tiles vt = o, o + 4 (fused_kernel: tile t is
    // owned by wave t % 4, slot t / 4)
    auto owned = [](int o) constexpr { return o < 2 ? 2 : 1; };
    auto pbase = [&](int o) constexpr { int b = 0; for (int i = 0; i < o; ++i) b += 3 * owned(i); return b; };
    double* park = space + grp * 4 * SM_REGION;  // the group's 18 parked tiles of 256 doubles
    double* parts = space;                       // [nparts][16][VMAX]
    double* wn = space + 8 * SM_ITEMS;           // NG = 4; smaller workgroups have fewer parts: see below
    if (NG < 4) wn = space + NG * SM_ITEMS;
    double* rs1 = wn + SM_ITEMS;                 // [16][12]
    double* rsum = rs1 + 16 * 12;                // [16]
    static_assert((NG < 4 ? NG : 8) * SM_ITEMS + SM_ITEMS + 16 * 12 + 16 <= NW * SM_REGION, "parts and the tail's scratch fit the regions");

    for (int step = 0; step < p.nsteps; ++step) {
        const bool last_step = step + 1 == p.nsteps;
        constexpr int NPAR = MULTI ? 2 : 1;
        double part[NPAR][2][4];  // [slab parity][owned tile][register]
#pragma unroll
        for (int a = 0; a < NPAR; ++a)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[a][i][r] = 0.0;
        for (int it = 0; it < tpw; ++it) {
            const int t = wave + NW * it;
            const bool has = t < ntiles;  // (uniform per wave)
            d4 g[VT];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) g[vt] = (d4){0, 0, 0, 0};
            if (has) {
                double x[VT][4];
                {
                    const double* xsrc = p.X + ((int64_t)t * 16 + q) * VMAX + c16;
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[vt][r] = xsrc[4 * r * VMAX + 16 * vt];
                }
                if (!resident) load_h(t);
                // ---- stage the H tile, P = Ht . W (process_tile's chain over the k-steps)
#pragma unroll
                for (int r = 0; r < 4; ++r) Hl[(q + 4 * r) * SM_LS + c16] = h[r];
                __builtin_amdgcn_wave_barrier();
                d4 pr[VT];
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) pr[vt] = (d4){0, 0, 0, 0};
                const double* ha = Hl + c16 * SM_LS + q;
                const double* wb = Wl + q * WS + c16;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ha[4 * s];
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) pr[vt] = mfma(a, wb[4 * s * WS + 16 * vt], pr[vt]);
                }
                // ---- R = X / P in place (div_path: the same six instructions per quotient)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[vt][r] = div_path(x[vt][r], pr[vt][r]);
                // ---- U = R . W^T: the chain over the 24 feature k-steps in order, R transposed through LDS one half
                // (48 features) at a time (A = R[n = c16][v = 4s + q], B = W[k = c16][v = 4s + q])
                d4 u = (d4){0, 0, 0, 0};
                const double* wu = Wl + c16 * WS + q;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Rh[(q + 4 * r) * SM_RSH + 16 * j + c16] = pr[3 * half + j][r];
                    __builtin_amdgcn_wave_barrier();
                    const double* ra = Rh + c16 * SM_RSH + q;
#pragma unroll
                    for (int s = 0; s < 12; ++s) u = mfma(ra[4 * s], wu[4 * (12 * half + s)], u);
                    __builtin_amdgcn_wave_barrier();
                }
                // ---- T = Ht^T . R from zero (A = H[n = 4r + q][k = c16], B = register r of R); after U: the accumulators of T then
                // live only from here to the park (128 registers per wave with sixteen waves)
                double ga[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ga[r] = Hl[(4 * r + q) * SM_LS + c16];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) g[vt] = mfma(ga[r], pr[vt][r], g[vt]);
                // ---- H update (_utils_klnmf.py:343-347); the tile stays in registers when it is this wave's only one
                double* hdst = p.H + ((int64_t)t * 16 + q) * SM_KP + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[r] = clip_lo(h[r] * u[r], kEps);
                    if (!resident || last_step) hdst[4 * r * SM_KP] = h[r];
                }
            }
            // ---- the group's four waves add their tiles up as fused_kernel's epilogue does: everybody parks the tiles it
            // does not own, the owner adds wave 0 + 1 + 2 + 3 in that order
            __syncthreads();  // the regions are free (every wave is past its LDS reads of this iteration)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int o = vt % 4, i = vt / 4;
                if (vw != o) {
                    const int src = vw < o ? vw : vw - 1;
                    double* dst = park + ((pbase(o) + src * owned(o) + i) * 4) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * 64] = g[vt][r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int o = vt % 4, i = vt / 4;
                if (vw == o) {
                    double acc[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int src = w < o ? w : w - 1;
                        const double* from = park + ((pbase(o) + src * owned(o) + i) * 4) * 64 + lane;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double v = (w == o) ? g[vt][r] : from[r * 64];
                            acc[r] = (w == 0) ? v : acc[r] + v;
                        }
                    }
                    // this slab is number g = grp + NG it; tail_row's first stage adds the slabs p, p + 8, ... of part p
                    // in ascending order, from zero (NG = 4 when there is more than one iteration: parity of `it`)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[MULTI ? (it & 1) : 0][i][r] = (it < 2 ? 0.0 : part[MULTI ? (it & 1) : 0][i][r]) + acc[r];
                }
            }
            if (it + 1 < tpw) __syncthreads();  // the parked tiles are read: the next tile may be staged
        }
        // ---- the parts in accumulator layout -> [part][k][v]
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NPAR; ++a) {
            const int pidx = grp + NG * a;  // (the group's slabs grp, grp + NG, grp + 2 NG, ...: parts grp and grp + 4)
            if (pidx < nparts && (a == 0 || tpw > 1)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int vt = vw + 4 * i;
                    if (vt < VT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) parts[pidx * SM_ITEMS + (q + 4 * r) * VMAX + 16 * vt + c16] = part[a][i][r];
                    }
                }
            }
        }
        __syncthreads();
        // ---- the W tail for all rows at once (tail_row's arithmetic and summation orders)
        constexpr int IPT = (SM_ITEMS + NT - 1) / NT;
        double gsum[IPT], wold[IPT];
#pragma unroll
        for (int c = 0; c < IPT; ++c) {
            const int item = tid + c * NT, k = item / VMAX, v = item - k * VMAX;
            gsum[c] = 0.0;
            wold[c] = 0.0;
            if (item < SM_ITEMS) {
                double t = 0.0;
                for (int i = 0; i < nparts; ++i) t += parts[i * SM_ITEMS + item];
                gsum[c] = t;
                const bool live = k < K && v < V;
                wold[c] = live ? Wl[k * WS + v] : 0.0;
                wn[item] = live ? wold[c] * t : 0.0;
            }
        }
        __syncthreads();
        if (tid < 16 * 12) {
            const int k = tid / 12, g8 = tid - 12 * k;
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += wn[k * VMAX + 8 * g8 + i];
            rs1[tid] = t;
        }
        __syncthreads();
        if (tid < 16) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 12; ++i) t += rs1[tid * 12 + i];
            rsum[tid] = t;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < IPT; ++c) {
            const int item = tid + c * NT, k = item / VMAX, v = item - k * VMAX;
            if (item < SM_ITEMS && k < K && v < V) {
                double w = wn[item] / rsum[k];
                if (k < p.n_given) {
                    w = wold[c];
                    if (p.clip_mode == 0) w = clip_lo(w, kEps);
                } else {
                    w = clip_lo(w, kEps);
                }
                Wl[k * WS + v] = w;
                if (last_step) {
                    p.Wout[k * V + v] = w;
                    p.G[k * V + v] = gsum[c];
                }
            }
        }
        __syncthreads();  // the new W is in place; the regions are free for the next step's tiles
    }
}

template <int KS>
int launch_ks(const SmallParams& p, hipStream_t stream) {
    const int ng = std::min(4, (p.ntiles + 3) / 4);
    switch (ng) {
        case 1: hipLaunchKernelGGL((small_kl_kernel<KS, 1, false>), dim3(1), dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((small_kl_kernel<KS, 2, false>), dim3(1), dim3(512), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((small_kl_kernel<KS, 3, false>), dim3(1), dim3(768), 0, stream, p); break;
        default:
            if (p.ntiles <= 16)
                hipLaunchKernelGGL((small_kl_kernel<KS, 4, false>), dim3(1), dim3(1024), 0, stream, p);
            else
                hipLaunchKernelGGL((small_kl_kernel<KS, 4, true>), dim3(1), dim3(1024), 0, stream, p);
            break;
    }
    return 0;
}

}  // namespace

int launch_small_kl_steps(int KS, const SmallParams& p, hipStream_t stream) {
    if (p.ntiles < 1 || p.ntiles > SMALL_MAX_TILES || p.K > 16) return 1;
    switch (KS) {
        case 1: return launch_ks<1>(p, stream);
        case 2: return launch_ks<2>(p, stream);
        case 4: return launch_ks<4>(p, stream);
        default: return 1;
    }
}

}  // namespace salnmf
