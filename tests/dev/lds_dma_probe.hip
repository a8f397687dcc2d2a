// Probe (development aid): where does global_load_lds_dwordx4 put lane l's 16 bytes?  hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, double* __restrict__ dst, int masked) {
    __shared__ __attribute__((aligned(16))) double buf[256];
    const int lane = threadIdx.x;
    buf[2 * lane] = -1.0, buf[2 * lane + 1] = -1.0, buf[128 + 2 * lane] = -2.0, buf[128 + 2 * lane + 1] = -2.0;
    __syncthreads();
    const double* mine = src + 2 * ((lane * 7) % 64);  // a permutation of the 16-byte pieces
    if (!masked || (lane & 1))                         // with `masked`: only odd lanes load
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mine, (__attribute__((address_space(3))) void*)(buf + 128), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) dst[i] = buf[i];
}
int main() {
    std::vector<double> h(128), out(256);
    for (int i = 0; i < 128; ++i) h[i] = i;
    double *s, *d;
    hipMalloc(&s, 128 * 8), hipMalloc(&d, 256 * 8);
    hipMemcpy(s, h.data(), 128 * 8, hipMemcpyHostToDevice);
    for (int masked = 0; masked < 2; ++masked) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, masked);
        hipMemcpy(out.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int l = 0; l < 64; ++l) {
            const bool loaded = !masked || (l & 1);
            const double want0 = loaded ? 2 * ((l * 7) % 64) : -2.0, want1 = loaded ? want0 + 1 : -2.0;
            if (out[128 + 2 * l] != want0 || out[128 + 2 * l + 1] != want1) ok = 0;
        }
        printf("masked=%d: lane l's 16 bytes at base + 16 l: %s; first entries %g %g %g %g; untouched half %g\n", masked, ok ? "yes" : "NO", out[128], out[129], out[130], out[131], out[0]);
    }
    return 0;
}
