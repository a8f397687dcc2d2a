// Development aid: timeline of the persistent KL kernel's steps on the chip-wide 100 MHz clock.
//   python tools/mk_pstamps_exp.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 \
//       -Iexp/pstamps tools/pstamps_bench.hip -o exp/pstamps/bench && exp/pstamps/bench [N]
#include "salnmf_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace salnmf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int K = 50, V = 96, KP = 64, STEPS = 12;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0.1, 1.0);
    const int64_t Np = (N + 15) / 16 * 16, ntiles = Np / 16;
    const int grid = (int)std::min<int64_t>(256, (ntiles + 3) / 4);
    std::vector<double> X(Np * 96), H(Np * KP), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    double *dX, *dH, *dW, *dG, *dGr; unsigned *sync, *abort_h; unsigned long long* dbg;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dW, W.size() * 8));
    CK(hipMalloc(&dG, (size_t)grid * K * VMAX * 8)); CK(hipMalloc(&dGr, (size_t)K * V * 8));
    CK(hipMalloc(&sync, SYNC_WORDS * 4)); CK(hipHostMalloc((void**)&abort_h, 64, hipHostMallocDefault)); *abort_h = 0;
    const size_t DS = (size_t)STEPS * grid * 8;
    CK(hipMalloc(&dbg, DS * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    FusedParams p{}; p.X = dX; p.H = dH; p.Hout = dH; p.hfloor = kEps; p.W = dW; p.Wmut = dW; p.Gpart = dG; p.G = dGr; p.N = N; p.V = V; p.K = K; p.ntiles = ntiles;
    p.nsteps = STEPS; p.n_given = 0; p.sync = sync; p.abort_host = abort_h; p.dbg = dbg;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(sync, 0, SYNC_WORDS * 4)); CK(hipMemset(dbg, 0, DS * 8));
        hipLaunchKernelGGL((fused_kernel<13, 3, 2, true, true, false, false, true>), dim3(grid), dim3(BLOCK), 0, 0, p);
        CK(hipDeviceSynchronize());
    }
    if (*abort_h) { printf("ABORTED\n"); return 1; }
    std::vector<unsigned long long> D(DS);
    CK(hipMemcpy(D.data(), dbg, DS * 8, hipMemcpyDeviceToHost));
    printf("N=%lld grid=%d; us relative to the earliest workgroup's step start; min/avg/max over workgroups (owners only for the last two)\n", (long long)N, grid);
    printf("step | W seen | W staged | tiles done | slab stored | published | (owner) slabs ready | (owner) row published | next step start (min)\n");
    for (int s = 2; s < STEPS - 1; ++s) {
        unsigned long long t0 = ~0ull, n0 = ~0ull;
        for (int w = 0; w < grid; ++w) { t0 = std::min(t0, D[((size_t)s * grid + w) * 8]); n0 = std::min(n0, D[((size_t)(s + 1) * grid + w) * 8]); }
        printf("%4d |", s);
        for (int i = 1; i < 8; ++i) {
            double mn = 1e30, mx = -1e30, av = 0; int c = 0;
            for (int w = 0; w < grid; ++w) { unsigned long long v = D[((size_t)s * grid + w) * 8 + i]; if (!v) continue; double t = (double)(long long)(v - t0) * 0.01; mn = std::min(mn, t); mx = std::max(mx, t); av += t; ++c; }
            if (c) printf(" %6.2f %6.2f %6.2f |", mn, av / c, mx); else printf("   -   |");
        }
        printf(" %6.2f\n", (double)(long long)(n0 - t0) * 0.01);
    }
    return 0;
}
