// Standalone timing harness for the fused update kernel (development aid, not product code).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I<dir with salnmf_kernels.h> tools/fused_bench.hip -o fb && ./fb [N] [K]
#include "salnmf_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
using namespace salnmf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#ifndef KSV
#define KSV 13
#define KTMV 3
#define KRV 2
#endif
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    int K = argc > 2 ? atoi(argv[2]) : 50, V = 96;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0.1, 1.0);
    const int KP = 16 * ((KSV + 3) / 4); const int64_t Np = (N + 15) / 16 * 16;
    std::vector<double> X(Np * 96), H(Np * KP), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    double *dX, *dH, *dH0, *dW, *dG;
    int grid = 256;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dH0, H.size() * 8)); CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dG, (size_t)grid * K * V * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH0, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    FusedParams p{}; p.X = dX; p.H = dH; p.W = dW; p.Gpart = dG; p.N = N; p.V = V; p.K = K; p.ntiles = (N + 15) / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kernel, double flops_per_elem) {
        float best = 1e30f, sum = 0; int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemcpy(dH, dH0, H.size() * 8, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, 0, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        double fl = flops_per_elem * V * K * N;
        printf("%-28s avg %.1f us  best %.1f us  -> %.1f TF/s algorithmic (%.1f%% of 78.6)\n", name, sum / reps * 1e3, best * 1e3, fl / (best * 1e-3) / 1e12, fl / (best * 1e-3) / 1e12 / 78.6 * 100);
    };
    run("fused G+U", fused_kernel<KSV, KTMV, KRV, true, true, false>, 6);
    run("fused U only", fused_kernel<KSV, KTMV, KRV, false, true, false>, 4);
    run("fused G only", fused_kernel<KSV, KTMV, KRV, true, false, false>, 4);
    double *dHs, *dKL; CK(hipMalloc(&dHs, (size_t)grid * K * 8)); CK(hipMalloc(&dKL, grid * 8)); p.Hsumpart = dHs; p.KLpart = dKL;
    run("fused G + KL stats (MvNMF)", fused_kernel<KSV, KTMV, KRV, true, false, true>, 4);
    run("fused U + rowsums (MvNMF)", fused_kernel<KSV, KTMV, KRV, false, true, true>, 4);
    return 0;
}
