// Development aid: what would the last-arriving workgroup of an XCD pay for reducing that XCD's 32
// numerator slabs inside the fused kernel?  It has to read 31 foreign slabs of K x 96 doubles (38.4 KB each at K = 50,
// 1.19 MB) with ONE workgroup of 256 threads before it can store the XCD's slab, while the other 31 workgroups are done.
// This probe times exactly that read-and-sum (8 workgroups, one per XCD, 32 slabs each, all loads of a thread in
// flight in batches) against the product's tail kernel shape (50 workgroups x 768 threads over all 256 slabs), on
// slabs that were just written by a 256-workgroup kernel (so they sit in the L2s / MALL as they would in the step).
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/xcd_reduce_probe tools/xcd_reduce_probe.hip && ./build/tools/xcd_reduce_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

constexpr int K = 50, V = 96, SLAB = K * V, NSLAB = 256;

__global__ void write_slabs(double* g) {  // what the fused kernel's epilogue leaves behind: one slab per workgroup
    double* out = g + (size_t)blockIdx.x * SLAB;
    for (int i = threadIdx.x; i < SLAB; i += blockDim.x) out[i] = 1.0 + blockIdx.x * 1e-3 + i * 1e-6;
}

// one workgroup per XCD: sum the 32 slabs blockIdx.x, blockIdx.x + 8, ... in fixed order, write one slab
__global__ void __launch_bounds__(256) xcd_reduce(const double* __restrict__ g, double* __restrict__ out) {
    for (int i0 = threadIdx.x * 2; i0 < SLAB; i0 += 512) {
        double2 t[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) t[s] = *reinterpret_cast<const double2*>(g + (size_t)(blockIdx.x + 8 * s) * SLAB + i0);
        double2 a = t[0];
#pragma unroll
        for (int s = 1; s < 32; ++s) { a.x += t[s].x; a.y += t[s].y; }
        *reinterpret_cast<double2*>(out + (size_t)blockIdx.x * SLAB + i0) = a;
    }
}

// the tail's shape: one workgroup per signature row, 768 threads = 96 features x 8 parts, 32 slabs per thread
__global__ void __launch_bounds__(768) tail_shape(const double* __restrict__ g, double* __restrict__ out, int nslabs) {
    __shared__ double red[8][V];
    const int k = blockIdx.x, part = threadIdx.x / V, v = threadIdx.x % V;
    double t[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int sl = part + 8 * j;
        t[j] = sl < nslabs ? g[(size_t)sl * SLAB + k * V + v] : 0.0;
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += t[j];
    red[part][v] = s;
    __syncthreads();
    if (threadIdx.x < V) {
        double a = 0.0;
        for (int i = 0; i < 8; ++i) a += red[i][threadIdx.x];
        out[k * V + threadIdx.x] = a;
    }
}

int main() {
    double *g, *o;
    hipMalloc(&g, sizeof(double) * SLAB * NSLAB);
    hipMalloc(&o, sizeof(double) * SLAB * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipStream_t st;
    hipStreamCreate(&st);
    auto time = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0;
        const int reps = 200;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(write_slabs, dim3(NSLAB), dim3(256), 0, st, g);
            launch(a, b);
            hipStreamSynchronize(st);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-58s mean %.2f us  best %.2f us\n", name, sum / reps * 1e3, best * 1e3);
    };
    time("xcd_reduce: 8 workgroups x 256 threads, 32 slabs each", [&](hipEvent_t s, hipEvent_t e) {
        hipExtLaunchKernelGGL(xcd_reduce, dim3(8), dim3(256), 0, st, s, e, 0, (const double*)g, o);
    });
    time("tail shape: 50 workgroups x 768 threads, 256 slabs", [&](hipEvent_t s, hipEvent_t e) {
        hipExtLaunchKernelGGL(tail_shape, dim3(K), dim3(768), 0, st, s, e, 0, (const double*)g, o, NSLAB);
    });
    time("tail shape: 50 workgroups x 768 threads, 8 slabs", [&](hipEvent_t s, hipEvent_t e) {
        hipExtLaunchKernelGGL(tail_shape, dim3(K), dim3(768), 0, st, s, e, 0, (const double*)g, o, 8);
    });
    return 0;
}
