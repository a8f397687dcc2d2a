// Probe for v_mfma_f64_16x16x4_f64 on gfx950: operand/accumulator lane maps and issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// D(16x16) = A(16x4) * B(4x16); A row-major [16][4], B row-major [4][16], D row-major [16][16]
__global__ void layout_kernel(const double* A, const double* B, double* D) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];   // A[i=l&15][k=l>>4]
    double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][j=l&15]
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row=(l>>4)+4r col=l&15
}

// chained use: second MFMA takes register r of D as B operand (rows 4r..4r+3 of D = k-slab)
// E(16x16) = A2(16x16) * D(16x16)
__global__ void chain_kernel(const double* A, const double* B, const double* A2, double* E) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    d4 e = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) {
        double a2 = A2[(l & 15) * 16 + 4 * r + (l >> 4)];  // A2[i][k=4r+(l>>4)]
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, c[r], e, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) E[((l >> 4) + 4 * r) * 16 + (l & 15)] = e[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA + independent VALU f64 FMA stream in the same wave: does VALU fp64 overlap the MFMA pipe?
template <int NFMA>
__global__ void __launch_bounds__(256) mix_kernel(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NFMA; ++j) f[j & 7] = __builtin_fma(f[j & 7], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// pure VALU f64 fma rate
__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = seed * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) f[j] = __builtin_fma(f[j], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// f64 division throughput (IEEE correctly rounded)
__global__ void __launch_bounds__(256) div_kernel(double* out, int iters, double seed) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x;
    double d = 1.0000001 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = x[j] / d;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// real shader clock: s_memtime ticks (shader cycles) vs s_memrealtime (100 MHz) around an MFMA loop
__global__ void __launch_bounds__(256) clock_kernel(unsigned long long* out, double* sink, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    // ---- layout
    std::vector<double> A(64), B(64), A2(256), D(256), E(256), Dr(256), Er(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 - j * 5 + (j * j % 7);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) A2[i * 16 + k] = 3 + i * 2 - k * 13 + (i * k % 5);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; Dr[i * 16 + j] = s; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += A2[i * 16 + k] * Dr[k * 16 + j]; Er[i * 16 + j] = s; }
    double *dA, *dB, *dA2, *dD, *dE;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dA2, 256 * 8)); CK(hipMalloc(&dD, 256 * 8)); CK(hipMalloc(&dE, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dA2, A2.data(), 256 * 8, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dD);
    chain_kernel<<<1, 64>>>(dA, dB, dA2, dE);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(E.data(), dE, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0, bad2 = 0;
    for (int i = 0; i < 256; ++i) { if (D[i] != Dr[i]) ++bad; if (E[i] != Er[i]) ++bad2; }
    printf("layout: %s (%d mismatches)  chain(acc as B operand): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad, bad2 ? "FAIL" : "PASS", bad2);
    // ---- rate
    int nblk = p.multiProcessorCount * 1;  // 1 block of 4 waves per CU => one wave per SIMD
    double* out; CK(hipMalloc(&out, sizeof(double) * nblk * 8 * 256));
    int iters = 20000;
    auto report = [&](const char* name, float ms, double nmfma_per_wave, double nfma_per_wave, int waves) {
        double flops = (nmfma_per_wave * 2048.0 + nfma_per_wave * 128.0) * waves;
        double cyc = ms * 1e-3 * p.clockRate * 1e3;
        printf("%-34s %8.3f ms  %7.2f TF/s  cycles/MFMA(at %d kHz)=%.1f\n", name, ms, flops / ms * 1e-9, p.clockRate, nmfma_per_wave > 0 ? cyc / nmfma_per_wave : 0.0);
    };
    float ms;
    ms = time_ms([&] { rate_kernel<1><<<nblk, 256>>>(out, iters, 1.0); });      report("mfma 1 acc (dependent), 1w/SIMD", ms, iters * 1.0, 0, nblk * 4);
    ms = time_ms([&] { rate_kernel<2><<<nblk, 256>>>(out, iters, 1.0); });      report("mfma 2 acc, 1w/SIMD", ms, iters * 2.0, 0, nblk * 4);
    ms = time_ms([&] { rate_kernel<4><<<nblk, 256>>>(out, iters, 1.0); });      report("mfma 4 acc, 1w/SIMD", ms, iters * 4.0, 0, nblk * 4);
    ms = time_ms([&] { rate_kernel<4><<<nblk * 2, 256>>>(out, iters, 1.0); });  report("mfma 4 acc, 2w/SIMD", ms, iters * 4.0, 0, nblk * 8);
    ms = time_ms([&] { fma_kernel<<<nblk, 256>>>(out, iters, 1.0); });          report("v_fma_f64 x16, 1w/SIMD", ms, 0, iters * 16.0, nblk * 4);
    ms = time_ms([&] { fma_kernel<<<nblk * 2, 256>>>(out, iters, 1.0); });      report("v_fma_f64 x16, 2w/SIMD", ms, 0, iters * 16.0, nblk * 8);
    ms = time_ms([&] { mix_kernel<4><<<nblk, 256>>>(out, iters, 1.0); });       report("mfma + 4 fma/gap, 1w/SIMD", ms, iters * 4.0, iters * 16.0, nblk * 4);
    ms = time_ms([&] { mix_kernel<8><<<nblk, 256>>>(out, iters, 1.0); });       report("mfma + 8 fma/gap, 1w/SIMD", ms, iters * 4.0, iters * 32.0, nblk * 4);
    ms = time_ms([&] { mix_kernel<12><<<nblk, 256>>>(out, iters, 1.0); });      report("mfma + 12 fma/gap, 1w/SIMD", ms, iters * 4.0, iters * 48.0, nblk * 4);
    ms = time_ms([&] { mix_kernel<16><<<nblk, 256>>>(out, iters, 1.0); });      report("mfma + 16 fma/gap, 1w/SIMD", ms, iters * 4.0, iters * 64.0, nblk * 4);
    ms = time_ms([&] { rate_kernel<4><<<nblk * 4, 256>>>(out, iters, 1.0); });  report("mfma 4 acc, 4w/SIMD", ms, iters * 4.0, 0, nblk * 16);
    ms = time_ms([&] { rate_kernel<8><<<nblk, 256>>>(out, iters, 1.0); });      report("mfma 8 acc, 1w/SIMD", ms, iters * 8.0, 0, nblk * 4);
    ms = time_ms([&] { mix_kernel<8><<<nblk * 2, 256>>>(out, iters, 1.0); });   report("mfma + 8 fma/gap, 2w/SIMD", ms, iters * 4.0, iters * 32.0, nblk * 8);
    {
        unsigned long long* ck; CK(hipMalloc(&ck, 16 * nblk * 2));
        for (int w = 1; w <= 2; ++w) {
            std::vector<unsigned long long> h(2 * nblk * w);
            for (int rep = 0; rep < 3; ++rep) clock_kernel<<<nblk * w, 256>>>(ck, out, iters, 1.0);
            CK(hipMemcpy(h.data(), ck, 16 * nblk * w, hipMemcpyDeviceToHost));
            double st = 0, rt = 0; for (int i = 0; i < nblk * w; ++i) { st += h[2 * i]; rt += h[2 * i + 1]; }
            printf("in-kernel (%dw/SIMD): shader cycles per MFMA (per wave) = %.1f, clock = %.3f GHz\n", w, st / (nblk * w) / (iters * 4.0), st / rt * 0.1);
        }
    }
    {
        ms = time_ms([&] { div_kernel<<<nblk, 256>>>(out, 2000, 1.0); });
        double ndiv = 2000.0 * 8 * 64 * nblk * 4;
        printf("f64 divide: %.3f ms, %.1f Gdiv/s, cycles per wave-divide(1w/SIMD)=%.1f\n", ms, ndiv / ms * 1e-6, ms * 1e-3 * p.clockRate * 1e3 / (2000.0 * 8));
    }
    return 0;
}
