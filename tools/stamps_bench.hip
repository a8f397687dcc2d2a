// Development aid: cycles per phase of one tile of the fused kernel (s_memtime stamps, tools/mk_stamps_exp.py).
//   python tools/mk_stamps_exp.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 \
//       -Iexp/stamps tools/stamps_bench.hip -o exp/stamps/bench && exp/stamps/bench [N]
#include "salnmf_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace salnmf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    int K = 50, V = 96;
    const int KP = 64;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0.1, 1.0);
    const int64_t Np = (N + 15) / 16 * 16;
    std::vector<double> X(Np * 96), H(Np * KP), W(K * V);
    for (auto& v : X) v = (double)(int)(U(rng) * 40);
    for (auto& v : H) v = U(rng) * 10;
    for (auto& v : W) v = U(rng) / 50;
    double *dX, *dH, *dW, *dG;
    int grid = 256;
    CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dG, (size_t)grid * K * V * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    unsigned long long* dbg; const size_t FS = (size_t)grid * WAVES * 8;
    CK(hipMalloc(&dbg, FS * 8)); CK(hipMemset(dbg, 0, FS * 8));
    FusedParams p{}; p.X = dX; p.H = dH; p.Hout = dH; p.hfloor = kEps; p.W = dW; p.Gpart = dG; p.N = N; p.V = V; p.K = K; p.ntiles = (N + 15) / 16; p.dbg = dbg;
    for (int s = 0; s < 10; ++s) hipLaunchKernelGGL((fused_kernel<13, 3, 2, true, true, false>), dim3(grid), dim3(BLOCK), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> F(FS);
    CK(hipMemcpy(F.data(), dbg, FS * 8, hipMemcpyDeviceToHost));
    const int NW = grid * WAVES; double ph[8] = {0}; double tiles = 0;
    for (int w = 0; w < NW; ++w) { int64_t nt = (p.ntiles - w + NW - 1) / NW; tiles += nt; for (int i = 0; i < 8; ++i) ph[i] += (double)F[w * 8 + i]; }
    const char* names[8] = {"stage H + P (78 MFMA)", "G operand reads + divide", "U (72 MFMA) + remainder reduce-scatter", "prefetch issue", "H update + stores", "loop overhead", "-", "tile entry"};
    // order of the stamps in the loop: 7 (entry), 0 (P done), 1 (div done), 3 (prefetch issued), 2 (R transpose + G + U remainder partials), 4 (U done), 5 (H update done)
    const char* seg[8] = {"[0] H staging + P product (78 MFMA)", "[1] G-operand LDS reads + 24 divides", "[2] R transpose write + G product (72 MFMA) + remainder partials", "[3] issue next tile's loads", "[4] U product (72 MFMA) + reduce-scatter", "[5] H update + stores", "[6] -", "[7] loop back edge"};
    double tot = 0; for (int i = 0; i < 8; ++i) tot += ph[i];
    printf("N=%lld, cycles per tile by segment (s_memtime; 64 cycles per f64 MFMA):\n", (long long)N);
    for (int i = 0; i < 8; ++i) printf("  %-70s %8.0f\n", seg[i], ph[i] / tiles);
    printf("  total %8.0f cycles per tile\n", tot / tiles);
    (void)names;
    return 0;
}
