// What issues for free in the shadow of v_mfma_f64_16x16x4_f64 (64 cycles/SIMD)?  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int KIND, int NF>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
    __shared__ double sm[1024];
    sm[threadIdx.x] = seed; sm[threadIdx.x + 256] = seed; __syncthreads();
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    unsigned u[8]; float f[8]; double dd[8];
    for (int i = 0; i < 8; ++i) { u[i] = threadIdx.x + i; f[i] = seed * i; dd[i] = seed; }
    const double* sp = sm + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                if (KIND == 0) u[j & 7] = u[j & 7] * 3u + 1u;                       // v_mad_u32 / int VALU
                if (KIND == 1) f[j & 7] = __builtin_fmaf(f[j & 7], 1.0001f, 0.5f);   // f32 VALU
                if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(dd[j & 7]) : "v"((unsigned)((threadIdx.x & 63) * 8 + 64 * j)) : "memory");
                if (KIND == 3) asm volatile("s_nop 0");
                if (KIND == 4) u[j & 7] = (u[j & 7] & 1) ? u[(j + 1) & 7] : u[(j + 2) & 7]; // v_cndmask
            }
            if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += u[i] + f[i] + dd[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + sp[0];
}

template <typename F> float tm(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}
int main() {
    double* out; CK(hipMalloc(&out, 8 * 256 * 256));
    int iters = 20000;
    auto rep = [&](const char* n, float ms) { printf("%-34s %.3f ms  -> %.1f cycles per MFMA at 2.4 GHz\n", n, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0)); };
    rep("mfma only", tm([&] { k<3, 0><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 4 int VALU", tm([&] { k<0, 4><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 8 int VALU", tm([&] { k<0, 8><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 16 int VALU", tm([&] { k<0, 16><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 8 f32 fma", tm([&] { k<1, 8><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 16 f32 fma", tm([&] { k<1, 16><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 8 cndmask", tm([&] { k<4, 8><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 2 ds_read_b64 (+wait)", tm([&] { k<2, 2><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 4 ds_read_b64 (+wait)", tm([&] { k<2, 4><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 8 s_nop", tm([&] { k<3, 8><<<256, 256>>>(out, iters, 1.0); }));
    rep("mfma + 16 s_nop", tm([&] { k<3, 16><<<256, 256>>>(out, iters, 1.0); }));
    return 0;
}
