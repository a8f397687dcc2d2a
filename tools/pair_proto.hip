// Prototype, measurement only: the "pair of waves" variant of the fused update pass.
// Two waves of one SIMD share a 16-sample tile: P, R and G are split by feature halves (48 columns each), U by
// signature tiles, R is exchanged through LDS as in the product kernel; two barriers per tile between the two waves of
// a pair only (an LDS counter; workgroup-wide barriers cost 3 % more).
// Restricted to K = 48 (three full signature tiles, no remainder columns), V = 96, no weights.
// Compared against the product kernel fused_kernel<13,3,0,G,U> on the same data (H must agree bit for bit).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Isalamander_amd/csrc tools/pair_proto.hip -o exp/pair_proto
#include "salnmf_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace proto {
using namespace salnmf;
constexpr int PKS = 12, PKT = 3, PKP = 48, PLS = PKP + 2;
constexpr int PAIRS = 4, PBLOCK = 512;
constexpr int HLSZ = 16 * PLS, RLSZ = 16 * RS;
constexpr int PLDS = PKP * WS + PAIRS * (2 * HLSZ + RLSZ);

// barrier between the two waves of a pair only (both are resident on one SIMD): an LDS counter that each wave
// bumps once per barrier; LDS operations of a wave execute in order, so the partner's earlier LDS writes are
// visible once its increment is.  `target` is the count that means "both arrived" for this barrier.
__device__ __forceinline__ void pair_barrier(unsigned* cnt, unsigned& target, int lane) {
    target += 2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void __launch_bounds__(PBLOCK, 1) pair_kernel(FusedParams p) {
    __shared__ __attribute__((aligned(16))) double lds[PLDS];
    __shared__ unsigned pcnt[PAIRS];
    if (threadIdx.x < PAIRS) pcnt[threadIdx.x] = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave & 3, half = wave >> 2;  // waves w and w + 4 sit on the same SIMD
    const int c16 = lane & 15, q = lane >> 4;
    const int V = p.V, K = p.K;
    double* Wl = lds;
    double* Hl = lds + PKP * WS + pair * (2 * HLSZ + RLSZ);
    double* Rl = Hl + 2 * HLSZ;
    for (int idx = tid; idx < PKP * VMAX; idx += PBLOCK) {
        const int k = idx / VMAX, v = idx - k * VMAX;
        Wl[k * WS + v] = (k < K) ? ((v < V) ? p.W[k * V + v] : 1.0) : 0.0;
    }
    d4 g[PKT][3];
#pragma unroll
    for (int kt = 0; kt < PKT; ++kt)
#pragma unroll
        for (int vt = 0; vt < 3; ++vt) g[kt][vt] = (d4){0, 0, 0, 0};

    // this wave's half of an H tile: 3 of the 6 16-byte pieces per lane
    int hrow[3], hcol[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = 2 * (lane + 64 * (3 * half + j));
        hrow[j] = e / PKP;
        hcol[j] = e - hrow[j] * PKP;
    }
    d2 hpre[3];
    double x[3][4];
    auto load_tile = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t n0 = t * 16;
        const d2* hsrc = reinterpret_cast<const d2*>(p.H + n0 * PKP) + lane + 64 * 3 * half;
#pragma unroll
        for (int j = 0; j < 3; ++j) hpre[j] = hsrc[64 * j];
        const double* xsrc = p.X + (n0 + q) * VMAX + 48 * half + c16;
#pragma unroll
        for (int vt = 0; vt < 3; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[vt][r] = xsrc[4 * r * VMAX + 16 * vt];
    };
    auto stage_H = [&](double* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<d2*>(dst + hrow[j] * PLS + hcol[j]) = hpre[j];
    };

    const int64_t tstride = (int64_t)gridDim.x * PAIRS;
    int64_t tile = (int64_t)blockIdx.x * PAIRS + pair;
    // every pair of a workgroup runs the same number of rounds (barriers are workgroup wide): pad with idle rounds
    const int64_t rounds = (p.ntiles + tstride - 1) / tstride;
    if (tile < p.ntiles) {
        load_tile(tile);
        stage_H(Hl);
    }
    __syncthreads();  // W and the counters are in LDS
    unsigned target = 0;
    int buf = 0;
    const int kt0 = half == 0 ? 0 : 2, kt1 = half == 0 ? 2 : 3;  // signature tiles of the U phase of this wave
    for (int64_t r_ = 0; r_ < rounds; ++r_, tile += tstride) {
        const bool live = tile < p.ntiles;
        const int64_t n0 = tile * 16;
        double* Hc = Hl + buf * HLSZ;
        pair_barrier(&pcnt[pair], target, lane);  // B1: the H tile is complete; the previous tile's readers of Rl are done
        d4 pr[3];
        double ga[4][PKT];
        if (live) {
#pragma unroll
            for (int vt = 0; vt < 3; ++vt) pr[vt] = (d4){0, 0, 0, 0};
            const double* ha = Hc + c16 * PLS + q;
            const double* wb = Wl + q * WS + 48 * half + c16;
            double a[2], b[2][3];
            a[0] = ha[0];
#pragma unroll
            for (int vt = 0; vt < 3; ++vt) b[0][vt] = wb[16 * vt];
#pragma unroll
            for (int s = 0; s < PKS; ++s) {
                if (s + 1 < PKS) {
                    a[(s + 1) & 1] = ha[4 * (s + 1)];
#pragma unroll
                    for (int vt = 0; vt < 3; ++vt) b[(s + 1) & 1][vt] = wb[4 * (s + 1) * WS + 16 * vt];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int vt = 0; vt < 3; ++vt) pr[vt] = mfma(a[s & 1], b[s & 1][vt], pr[vt]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < PKT; ++kt) ga[r][kt] = Hc[(4 * r + q) * PLS + 16 * kt + c16];
#pragma unroll
            for (int vt = 0; vt < 3; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[vt][r] = div_path(x[vt][r], pr[vt][r]);
            if (tile + tstride < p.ntiles) load_tile(tile + tstride);
#pragma unroll
            for (int vt = 0; vt < 3; ++vt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Rl[(q + 4 * r) * RS + 48 * half + 16 * vt + c16] = pr[vt][r];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < PKT; ++kt)
#pragma unroll
                    for (int vt = 0; vt < 3; ++vt) mfma_agpr(g[kt][vt], ga[r][kt], pr[vt][r]);
        }
        pair_barrier(&pcnt[pair], target, lane);  // B2: both halves of R are in LDS
        // U phase split by signature tiles: wave 0 of the pair takes tiles {0, 1}, wave 1 tile {2}.  (Splitting the
        // middle tile's feature steps between the two waves to balance 36 / 36 MFMAs needs a third barrier and an
        // exchange of partial sums: measured 2 % slower than this 48 / 24 split.)
        if (live) {
            const double* ra = Rl + c16 * RS + q;
            double* hdst = p.Hout + (n0 + q) * PKP + c16;
#pragma unroll
            for (int kt = 0; kt < PKT; ++kt) {
                if (kt < kt0 || kt >= kt1) continue;
                d4 u = (d4){0, 0, 0, 0};
                double hcur[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hcur[r] = Hc[(q + 4 * r) * PLS + 16 * kt + c16];
                const double* wb = Wl + (16 * kt + c16) * WS + q;
#pragma unroll
                for (int s = 0; s < VSTEPS; ++s) u = mfma(ra[4 * s], wb[4 * s], u);
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(fmax(hcur[r] * u[r], p.hfloor), &hdst[4 * r * PKP + 16 * kt]);
            }
            if (tile + tstride < p.ntiles) stage_H(Hl + (buf ^ 1) * HLSZ);
        }
        buf ^= 1;
    }
    __syncthreads();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < PKT; ++kt)
#pragma unroll
        for (int vt = 0; vt < 3; ++vt) asm volatile("" : "+a"(g[kt][vt]));
    // per-wave slab [K][48 of this half]; the host sums the waves (timing prototype: no in-kernel reduction)
    double* out = p.Gpart + ((int64_t)blockIdx.x * 8 + wave) * PKP * 48;
#pragma unroll
    for (int kt = 0; kt < PKT; ++kt)
#pragma unroll
        for (int vt = 0; vt < 3; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * kt + q + 4 * r) * 48 + 16 * vt + c16] = g[kt][vt][r];
}
}  // namespace proto

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    using namespace salnmf;
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int K = 48, V = 96, KP = 48, grid = 256;
    const int64_t Np = (N + 15) / 16 * 16;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0.1, 1.0);
    const int KPA = 64;  // the product kernel's geometry for K = 48 is KS = 13: H rows of 64 doubles (pad columns 0)
    std::vector<double> X(Np * 96), H(Np * KP), HA(Np * KPA, 0.0), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    for (int64_t n = 0; n < Np; ++n) for (int k = 0; k < K; ++k) HA[n * KPA + k] = H[n * KP + k];
    double *dX, *dH, *dHA, *dHa, *dHb, *dW, *dG;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dHA, HA.size() * 8)); CK(hipMalloc(&dHa, HA.size() * 8)); CK(hipMalloc(&dHb, H.size() * 8));
    CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dG, (size_t)grid * 8 * K * 96 * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dHA, HA.data(), HA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    FusedParams p{}; p.X = dX; p.W = dW; p.Gpart = dG; p.N = N; p.V = V; p.K = K; p.ntiles = Np / 16; p.hfloor = kEps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto once = [&](int which, double* hout) {
        p.Hout = hout;
        p.H = which == 0 ? dHA : dH;
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL((fused_kernel<13, 3, 0, true, true, false>), dim3(grid), dim3(256), 0, 0, p);
        else hipLaunchKernelGGL(proto::pair_kernel, dim3(grid), dim3(512), 0, 0, p);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
    };
    std::vector<float> ta, tb;
    for (int r = 0; r < 5; ++r) { once(0, dHa); once(1, dHb); }
    for (int r = 0; r < 30; ++r) { ta.push_back(once(0, dHa)); tb.push_back(once(1, dHb)); }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    printf("N=%lld K=48: product %.2f us (min %.2f) | pair prototype %.2f us (min %.2f) | ratio %.4f\n", (long long)N, ta[15] * 1e3, ta[0] * 1e3, tb[15] * 1e3, tb[0] * 1e3, tb[15] / ta[15]);
    std::vector<double> ha(HA.size()), hb(H.size());
    CK(hipMemcpy(ha.data(), dHa, ha.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), dHb, hb.size() * 8, hipMemcpyDeviceToHost));
    size_t diff = 0; for (int64_t n = 0; n < N; ++n) for (int k = 0; k < K; ++k) diff += ha[n * KPA + k] != hb[n * KP + k];
    double maxrel = 0; for (int64_t n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { double a = ha[n * KPA + k], b = hb[n * KP + k]; maxrel = std::max(maxrel, std::abs(a - b) / std::abs(a)); }
    printf("H entries differing: %zu of %lld (max rel diff %.2e)\n", diff, (long long)N * K, maxrel);
    return 0;
}
