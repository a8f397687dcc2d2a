// In-process A/B of two versions of salnmf_kernels.h (exp/ab/a vs exp/ab/b): blocks of 50 product-like steps
// (fused kernel + W tail, back to back on one stream), A and B interleaved, medians over the blocks.
//   mkdir -p exp/ab/a exp/ab/b; git show HEAD:salamander_amd/csrc/salnmf_kernels.h > exp/ab/a/salnmf_kernels.h
//   cp salamander_amd/csrc/salnmf_kernels.h exp/ab/b/
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Iexp/ab tools/ab_bench.hip -o exp/ab/bench && exp/ab/bench [N]
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
#ifndef AB_KERNEL_A
#define AB_KERNEL_A (salnmf_A::fused_kernel<13, 3, 2, true, true, false>)
#endif
#ifndef AB_KERNEL_B
#define AB_KERNEL_B (salnmf_B::fused_kernel<13, 3, 2, true, true, false>)
#endif
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int K = 50, V = 96, KP = 64, grid = 256, STEPS = 50, ROUNDS = 12;
    const int64_t Np = (N + 15) / 16 * 16;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0.1, 1.0);
    std::vector<double> X(Np * 96), H(Np * KP), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    double *dX, *dH, *dH0, *dW, *dW0, *dG, *dGr;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dH0, H.size() * 8)); CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dW0, W.size() * 8));
    CK(hipMalloc(&dG, (size_t)grid * K * 96 * 8)); CK(hipMalloc(&dGr, K * V * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH0, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW0, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    salnmf_A::FusedParams pa{}; pa.X = dX; pa.H = dH; pa.Hout = dH; pa.hfloor = salnmf_A::kEps; pa.W = dW; pa.Gpart = dG; pa.N = N; pa.V = V; pa.K = K; pa.ntiles = Np / 16;
    salnmf_B::FusedParams pb{}; pb.X = dX; pb.H = dH; pb.Hout = dH; pb.hfloor = salnmf_B::kEps; pb.W = dW; pb.Gpart = dG; pb.N = N; pb.V = V; pb.K = K; pb.ntiles = Np / 16;
    salnmf_A::TailParams ta{}; ta.Gpart = dG; ta.G = dGr; ta.W = dW; ta.nslabs = grid; ta.V = V; ta.K = K; ta.do_tail = 1; ta.nparts = grid;
    salnmf_B::TailParams tb{}; tb.Gpart = dG; tb.G = dGr; tb.W = dW; tb.nslabs = grid; tb.V = V; tb.K = K; tb.do_tail = 1; tb.nparts = grid;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto block = [&](int which, int steps) {
        CK(hipMemcpyAsync(dH, dH0, H.size() * 8, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(dW, dW0, W.size() * 8, hipMemcpyDeviceToDevice, st));
        // two untimed steps warm the caches behind the copies
        for (int s = -2; s < steps; ++s) {
            if (s == 0) CK(hipEventRecord(e0, st));
            if (which == 0) { hipLaunchKernelGGL(AB_KERNEL_A, dim3(grid), dim3(256), 0, st, pa); hipLaunchKernelGGL(salnmf_A::tail_kernel, dim3(K), dim3(salnmf_A::TAIL_BLOCK), 0, st, ta); }
            else { hipLaunchKernelGGL(AB_KERNEL_B, dim3(grid), dim3(256), 0, st, pb); hipLaunchKernelGGL(salnmf_B::tail_kernel, dim3(K), dim3(salnmf_B::TAIL_BLOCK), 0, st, tb); }
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / steps * 1e3f;
    };
    std::vector<float> va, vb;
    block(0, 10); block(1, 10);
    for (int r = 0; r < ROUNDS; ++r) { va.push_back(block(0, STEPS)); vb.push_back(block(1, STEPS)); }
    std::sort(va.begin(), va.end()); std::sort(vb.begin(), vb.end());
    printf("N=%lld us/step (fused + tail): A median %.2f min %.2f | B median %.2f min %.2f | B/A median %.4f\n", (long long)N, va[ROUNDS / 2], va[0], vb[ROUNDS / 2], vb[0], vb[ROUNDS / 2] / va[ROUNDS / 2]);
    // the trajectories after 5 steps must agree bit for bit when the arithmetic is the same
    std::vector<double> wa(K * V), wb(K * V), ha(H.size()), hb(H.size());
    block(0, 3); CK(hipMemcpy(wa.data(), dW, wa.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ha.data(), dH, ha.size() * 8, hipMemcpyDeviceToHost));
    block(1, 3); CK(hipMemcpy(wb.data(), dW, wb.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), dH, hb.size() * 8, hipMemcpyDeviceToHost));
    size_t dw = 0, dh = 0; for (size_t i = 0; i < wa.size(); ++i) dw += wa[i] != wb[i]; for (size_t i = 0; i < ha.size(); ++i) dh += ha[i] != hb[i];
    printf("after 5 steps, entries differing: W %zu of %zu, H %zu of %zu\n", dw, wa.size(), dh, ha.size());
    return 0;
}
