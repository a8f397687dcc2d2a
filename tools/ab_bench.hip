// In-process A/B of two versions of salnmf_kernels.h (exp/ab/a vs exp/ab/b), interleaved rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Iexp/ab tools/ab_bench.hip -o ab && ./ab [N]
#define salnmf salnmf_A
#include "a/salnmf_kernels.h"
#undef salnmf
#define salnmf salnmf_B
#include "b/salnmf_kernels.h"
#undef salnmf
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int K = 50, V = 96, KP = 64, grid = 256;
    const int64_t Np = (N + 15) / 16 * 16;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0.1, 1.0);
    std::vector<double> X(Np * 96), H(Np * KP), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    double *dX, *dH, *dH0, *dW, *dG;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dH0, H.size() * 8)); CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dG, (size_t)grid * K * V * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH0, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    salnmf_A::FusedParams pa{}; pa.X = dX; pa.H = dH; pa.W = dW; pa.Gpart = dG; pa.N = N; pa.V = V; pa.K = K; pa.ntiles = Np / 16;
    salnmf_B::FusedParams pb{}; pb.X = dX; pb.H = dH; pb.W = dW; pb.Gpart = dG; pb.N = N; pb.V = V; pb.K = K; pb.ntiles = Np / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double *dGr, *dW2; CK(hipMalloc(&dGr, K * V * 8)); CK(hipMalloc(&dW2, K * V * 8)); CK(hipMemcpy(dW2, dW, K * V * 8, hipMemcpyDeviceToDevice));
    salnmf_A::TailParams tp{}; tp.Gpart = dG; tp.G = dGr; tp.W = dW2; tp.nslabs = grid; tp.V = V; tp.K = K; tp.n_given = 0; tp.clip_mode = 0; tp.do_tail = 1;  // PAIR: fused + tail
    std::vector<float> ta, tb;
    auto once = [&](int which) {
        CK(hipMemcpy(dH, dH0, H.size() * 8, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL((salnmf_A::fused_kernel<13, 3, 2, true, true, false>), dim3(grid), dim3(256), 0, 0, pa);
        else hipLaunchKernelGGL((salnmf_B::fused_kernel<13, 3, 2, true, true, false>), dim3(grid), dim3(256), 0, 0, pb);
        hipLaunchKernelGGL(salnmf_A::tail_kernel, dim3(K), dim3(salnmf_A::TAIL_BLOCK), 0, 0, tp);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
    };
    for (int r = 0; r < 6; ++r) { once(0); once(1); }
    for (int r = 0; r < 40; ++r) { ta.push_back(once(0)); tb.push_back(once(1)); }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    printf("A: median %.2f us  min %.2f us | B: median %.2f us  min %.2f us | B/A median %.4f\n", ta[20] * 1e3, ta[0] * 1e3, tb[20] * 1e3, tb[0] * 1e3, tb[20] / ta[20]);
    // outputs must agree bit for bit (same arithmetic)
    std::vector<double> ga((size_t)grid * K * V), gb(ga.size()), ha(H.size()), hb(H.size());
    once(0); CK(hipMemcpy(ga.data(), dG, ga.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ha.data(), dH, ha.size() * 8, hipMemcpyDeviceToHost));
    once(1); CK(hipMemcpy(gb.data(), dG, gb.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), dH, hb.size() * 8, hipMemcpyDeviceToHost));
    size_t dg = 0, dh = 0; for (size_t i = 0; i < ga.size(); ++i) dg += ga[i] != gb[i]; for (size_t i = 0; i < ha.size(); ++i) dh += ha[i] != hb[i];
    printf("outputs differing: Gpart %zu, H %zu\n", dg, dh);
    return 0;
}
