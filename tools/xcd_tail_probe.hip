// Development aid (round 5): would a tail whose first stage runs on the XCD that WROTE the slabs it reads be faster?  The fused
// kernel's workgroup b runs on XCD b % 8 (round-robin dispatch) and leaves its slab in that XCD's L2 (written back to memory at
// the kernel boundary, but possibly still valid there).  Shapes timed on slabs just written by a 256-workgroup kernel:
//   tail shape        : the product's -- 50 workgroups x 768 threads, each reads its row of all 256 slabs
//   two-stage matched : 400 workgroups (row k, XCD x) at blockIdx = 8 k + x -> XCD x: sum the 32 slabs written on XCD x
//                       (x, x + 8, ...), store a partial row, ticket; the last of a row's eight adds the partials in order
//   two-stage shifted : the same with every workgroup reading the slabs of XCD x + 3 (no L2 locality): the control
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/xcd_tail_probe tools/xcd_tail_probe.hip && ./build/tools/xcd_tail_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

constexpr int K = 50, V = 96, SLAB = K * V, NSLAB = 256;

__global__ void write_slabs(double* g) {
    double* out = g + (size_t)blockIdx.x * SLAB;
    for (int i = threadIdx.x; i < SLAB; i += blockDim.x) out[i] = 1.0 + blockIdx.x * 1e-3 + i * 1e-6;
}

__global__ void __launch_bounds__(768) tail_shape(const double* __restrict__ g, double* __restrict__ out) {
    __shared__ double red[8][V];
    const int k = blockIdx.x, part = threadIdx.x / V, v = threadIdx.x % V;
    double t[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) t[j] = g[(size_t)(part + 8 * j) * SLAB + k * V + v];
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

// 384 threads = 96 features x 4 quarters of the XCD's 32 slabs (8 loads in flight per thread)
__global__ void __launch_bounds__(384) two_stage(const double* __restrict__ g, double* __restrict__ partial, unsigned* __restrict__ tickets,
                                                 double* __restrict__ out, int shift) {
    __shared__ double red[4][V];
    __shared__ unsigned last;
    const int k = blockIdx.x >> 3, x = blockIdx.x & 7, xs = (x + shift) & 7;
    const int qtr = threadIdx.x / V, v = threadIdx.x % V;
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = g[(size_t)(xs + 8 * (qtr + 4 * j)) * SLAB + k * V + v];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += t[j];
    red[qtr][v] = s;
    __syncthreads();
    if (threadIdx.x < V) {
        const double a = ((red[0][v] + red[1][v]) + red[2][v]) + red[3][v];
        __hip_atomic_store(partial + ((size_t)k * 8 + x) * V + v, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(tickets + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7u;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) tickets[k] = 0;
    if (threadIdx.x < V) {
        double p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __hip_atomic_load(partial + ((size_t)k * 8 + i) * V + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double a = p[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) a += p[i];
        out[k * V + v] = a;
    }
}

int main() {
    double *g, *o, *part;
    unsigned* tick;
    hipMalloc(&g, sizeof(double) * SLAB * NSLAB);
    hipMalloc(&o, sizeof(double) * SLAB);
    hipMalloc(&part, sizeof(double) * SLAB * 8);
    hipMalloc(&tick, sizeof(unsigned) * K);
    hipMemset(tick, 0, sizeof(unsigned) * K);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipStream_t st;
    hipStreamCreate(&st);
    auto time = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0;
        const int reps = 300;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(write_slabs, dim3(NSLAB), dim3(256), 0, st, g);
            launch(a, b);
            hipStreamSynchronize(st);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-62s mean %.2f us  best %.2f us\n", name, sum / reps * 1e3, best * 1e3);
    };
    for (int rep = 0; rep < 2; ++rep) {
        time("tail shape: 50 workgroups x 768 threads, 256 slabs", [&](hipEvent_t s, hipEvent_t e) {
            hipExtLaunchKernelGGL(tail_shape, dim3(K), dim3(768), 0, st, s, e, 0, (const double*)g, o);
        });
        time("two-stage, first stage on the XCD that wrote its slabs", [&](hipEvent_t s, hipEvent_t e) {
            hipExtLaunchKernelGGL(two_stage, dim3(8 * K), dim3(384), 0, st, s, e, 0, (const double*)g, part, tick, o, 0);
        });
        time("two-stage, first stage reading another XCD's slabs (control)", [&](hipEvent_t s, hipEvent_t e) {
            hipExtLaunchKernelGGL(two_stage, dim3(8 * K), dim3(384), 0, st, s, e, 0, (const double*)g, part, tick, o, 3);
        });
    }
    return 0;
}
