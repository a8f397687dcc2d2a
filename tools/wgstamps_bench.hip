// Development aid: where a KL step's time goes, on the chip-wide 100 MHz clock (s_memrealtime).
// Runs the product's step (fused kernel + W tail) back to back on one stream and prints, for the later steps,
// when each wave of the fused kernel passed its stamps and when the tail ran.
//   python tools/mk_wgstamps_exp.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 \
//       -Iexp/wgstamps tools/wgstamps_bench.hip -o exp/wgstamps/bench && exp/wgstamps/bench [N] [K]
#include "salnmf_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace salnmf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    int K = 50, V = 96;
    const int KSV = 13, KP = 64, STEPS = 24;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0.1, 1.0);
    const int64_t Np = (N + 15) / 16 * 16;
    std::vector<double> X(Np * 96), H(Np * KP), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    double *dX, *dH, *dW, *dG, *dGr;
    int grid = 256;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dW, W.size() * 8));
    CK(hipMalloc(&dG, (size_t)grid * K * VMAX * 8)); CK(hipMalloc(&dGr, (size_t)K * V * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    unsigned long long *dbgF, *dbgT;
    const size_t FS = (size_t)grid * WAVES * 8, TS = 2 * 64;
    CK(hipMalloc(&dbgF, STEPS * FS * 8)); CK(hipMalloc(&dbgT, STEPS * TS * 8));
    CK(hipMemset(dbgF, 0, STEPS * FS * 8)); CK(hipMemset(dbgT, 0, STEPS * TS * 8));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    FusedParams p{}; p.X = dX; p.H = dH; p.Hout = dH; p.hfloor = kEps; p.W = dW; p.Gpart = dG; p.N = N; p.V = V; p.K = K; p.ntiles = (N + 15) / 16;
    TailParams t{}; t.Gpart = dG; t.G = dGr; t.W = dW; t.nslabs = grid; t.V = V; t.K = K; t.n_given = 0; t.clip_mode = 0; t.do_tail = 1; t.nparts = grid;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int s = 0; s < STEPS; ++s) {
        p.dbg = dbgF + s * FS; t.dbg = dbgT + s * TS;
        hipLaunchKernelGGL((fused_kernel<KSV, 3, 2, true, true, false>), dim3(grid), dim3(BLOCK), 0, st, p);
        hipLaunchKernelGGL(tail_kernel, dim3(K), dim3(TAIL_BLOCK), 0, st, t);
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("N=%lld: %d steps, %.2f us/step (events)\n", (long long)N, STEPS, ms / STEPS * 1e3);
    std::vector<unsigned long long> F(STEPS * FS), T(STEPS * TS);
    CK(hipMemcpy(F.data(), dbgF, F.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(T.data(), dbgT, T.size() * 8, hipMemcpyDeviceToHost));
    const int NW = grid * WAVES;
    printf("all times in us relative to the first wave's entry into that step's fused kernel (clock tick = 10 ns)\n");
    printf("step | entry max | W staged avg max | 1st tile done avg | loop done min avg max | coop tile done max | wg barrier max | end min avg max | tail start min, end max | next entry min\n");
    for (int s = 8; s < STEPS - 1; ++s) {
        const unsigned long long* f = &F[s * FS];
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < NW; ++w) t0 = std::min(t0, f[w * 8]);
        auto stat = [&](int i, double& mn, double& av, double& mx) {
            mn = 1e30; mx = -1e30; av = 0;
            for (int w = 0; w < NW; ++w) { double v = (double)(long long)(f[w * 8 + i] - t0) * 0.01; mn = std::min(mn, v); mx = std::max(mx, v); av += v; }
            av /= NW;
        };
        double a[7][3];
        for (int i = 0; i < 7; ++i) stat(i, a[i][0], a[i][1], a[i][2]);
        double ts = 1e30, te = -1e30;
        for (int k = 0; k < K; ++k) { ts = std::min(ts, (double)(long long)(T[s * TS + 2 * k] - t0) * 0.01); te = std::max(te, (double)(long long)(T[s * TS + 2 * k + 1] - t0) * 0.01); }
        unsigned long long n0 = ~0ull;
        for (int w = 0; w < NW; ++w) n0 = std::min(n0, F[(s + 1) * FS + w * 8]);
        printf("%4d | %5.2f | %5.2f %5.2f | %5.2f | %6.2f %6.2f %6.2f | %6.2f | %6.2f | %6.2f %6.2f %6.2f | %6.2f %6.2f | %6.2f\n", s, a[0][2], a[1][1], a[1][2], a[2][1],
               a[3][0], a[3][1], a[3][2], a[6][2], a[4][2], a[5][0], a[5][1], a[5][2], ts, te, (double)(long long)(n0 - t0) * 0.01);
    }
    // loop-done histogram of the last measured step: waves with 6 vs 7 tiles
    {
        const int s = STEPS - 2; const unsigned long long* f = &F[s * FS];
        unsigned long long t0 = ~0ull; for (int w = 0; w < NW; ++w) t0 = std::min(t0, f[w * 8]);
        const int64_t ntiles = p.ntiles; double sum[2] = {0, 0}; int cnt[2] = {0, 0}; double per[2] = {0, 0};
        const int64_t base = ntiles / NW;
        for (int w = 0; w < NW; ++w) { int64_t nt = (ntiles - w + NW - 1) / NW; int c = nt > base ? 1 : 0; cnt[c]++; sum[c] += (double)(long long)(f[w * 8 + 3] - t0) * 0.01; per[c] += (double)(long long)(f[w * 8 + 3] - f[w * 8 + 1]) * 0.01 / nt; }
        for (int c = 0; c < 2; ++c) if (cnt[c]) printf("waves with %lld tiles: %d, loop done avg %.2f us, %.2f us per tile\n", (long long)(base + c), cnt[c], sum[c] / cnt[c], per[c] / cnt[c]);
    }
    // who is slow?  workgroup-level loop-done time (max over its 4 waves), averaged over the measured steps: by XCD
    // (workgroups are dealt round-robin: XCD = blockIdx % 8), its spread, and how persistent a workgroup's rank is
    {
        std::vector<double> wg(grid, 0.0), first(grid, 0.0), second(grid, 0.0);
        int ns = 0;
        for (int s = 8; s < STEPS - 1; ++s, ++ns) {
            const unsigned long long* f = &F[s * FS];
            unsigned long long t0 = ~0ull; for (int w = 0; w < NW; ++w) t0 = std::min(t0, f[w * 8]);
            for (int b = 0; b < grid; ++b) {
                double m = 0; for (int w = 0; w < WAVES; ++w) m = std::max(m, (double)(long long)(f[(b * WAVES + w) * 8 + 3] - t0) * 0.01);
                wg[b] += m; (ns % 2 ? second : first)[b] += m;
            }
        }
        double xs[8] = {0}, all = 0; for (int b = 0; b < grid; ++b) { wg[b] /= ns; xs[b % 8] += wg[b] / (grid / 8); all += wg[b] / grid; }
        printf("loop done per workgroup (max of its waves), mean over %d steps: all %.2f us; by XCD:", ns, all);
        for (int x = 0; x < 8; ++x) printf(" %.2f", xs[x]);
        std::vector<double> sorted(wg); std::sort(sorted.begin(), sorted.end());
        printf("\n  fastest %.2f, 10%% %.2f, median %.2f, 90%% %.2f, slowest %.2f\n", sorted[0], sorted[grid / 10], sorted[grid / 2], sorted[grid * 9 / 10], sorted[grid - 1]);
        // persistence: correlation between a workgroup's mean over the even and over the odd measured steps
        double ma = 0, mb = 0; for (int b = 0; b < grid; ++b) { ma += first[b]; mb += second[b]; } ma /= grid; mb /= grid;
        double sab = 0, saa = 0, sbb = 0; for (int b = 0; b < grid; ++b) { sab += (first[b] - ma) * (second[b] - mb); saa += (first[b] - ma) * (first[b] - ma); sbb += (second[b] - mb) * (second[b] - mb); }
        printf("  correlation of a workgroup's loop-done time between even and odd steps: %.3f\n", sab / std::sqrt(saa * sbb));
        double c0 = 0, c1 = 0; int n0 = 0, n1 = 0; const int64_t nleft = p.ntiles % ((int64_t)grid * WAVES);
        for (int b = 0; b < grid; ++b) { if (b < nleft) { c0 += wg[b]; ++n0; } else { c1 += wg[b]; ++n1; } }
        if (n0 && n1) printf("  workgroups with a cooperative tile (blockIdx < %lld): %.2f us; the others: %.2f us\n", (long long)nleft, c0 / n0, c1 / n1);
    }
    return 0;
}
