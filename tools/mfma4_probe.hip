// Development aid: issue rate of v_mfma_f64_4x4x4_4b_f64 against v_mfma_f64_16x16x4_f64 on gfx950 (cycles per instruction
// of one wave issuing back to back into independent accumulators; all four SIMDs of every CU busy).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o build/tools/mfma4_probe tools/mfma4_probe.hip && ./build/tools/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) probe(double* out, int iters, long long* cyc) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 acc16[8];
    double acc4[8];
    for (int i = 0; i < 8; ++i) { acc16[i] = (d4){0, 0, 0, 0}; acc4[i] = 0.0; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += MODE == 0 ? acc16[i][0] + acc16[i][3] : acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8;
        const double macs = mode == 0 ? 16.0 * 16 * 4 : 4.0 * 4 * 4 * 4;
        printf("%s: %.3f ms for %d x 8 instructions per wave: %.1f ns per instruction, %.2f TMAC/s over 1024 waves (memtime ticks per instr %.2f)\n",
               mode == 0 ? "v_mfma_f64_16x16x4" : "v_mfma_f64_4x4x4_4b", ms, iters, ms * 1e6 / n, macs * n * 1024 / (ms * 1e-3) / 1e12, (double)c / n);
    }
    return 0;
}
