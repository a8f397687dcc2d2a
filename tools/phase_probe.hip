// What do the LDS operand reads of the fused kernel's MFMA phases cost?  Every wave of a full grid (256 workgroups x 4
// waves, one per SIMD, as the product kernel) runs one phase shape REP times; time per MFMA against the same chains fed
// from registers.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/phase_probe.hip -o exp/phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int WS = 98, LS = 66, KS = 13, VT = 6, KT = 3, VSTEPS = 24, RS = 98;
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(double* out, unsigned long long* ticks, int rep) {
    __shared__ double lds[64 * WS + 4 * (16 * LS + 16 * RS)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < 64 * WS + 4 * (16 * LS + 16 * RS); i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double* Wl = lds;
    double* Hl = lds + 64 * WS + wave * (16 * LS + 16 * RS);
    double* Rl = Hl + 16 * LS;
    d4 acc[VT];
    for (int v = 0; v < VT; ++v) acc[v] = (d4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < rep; ++it) {
        if (MODE == 0) {  // register operands, 6 chains, 78 MFMAs
            double a = 1.0 + it, b = 2.0;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int v = 0; v < VT; ++v) acc[v] = mfma(a, b, acc[v]);
        } else if (MODE == 1) {  // the P phase: 1 + 6 LDS reads per 6 MFMAs, one k-step ahead
            const double* ha = Hl + c16 * LS + q;
            const double* wb = Wl + q * WS + c16;
            double a[2], b[2][VT];
            a[0] = ha[0];
#pragma unroll
            for (int v = 0; v < VT; ++v) b[0][v] = wb[16 * v];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    a[(s + 1) & 1] = ha[4 * (s + 1)];
#pragma unroll
                    for (int v = 0; v < VT; ++v) b[(s + 1) & 1][v] = wb[4 * (s + 1) * WS + 16 * v];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < VT; ++v) acc[v] = mfma(a[s & 1], b[s & 1][v], acc[v]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 2) {  // the U phase: 1 + 3 LDS reads per 3 MFMAs, 72 MFMAs
            const double* ra = Rl + c16 * RS + q;
            const double* wb = Wl + c16 * WS + q;
            double a[2], b[2][KT];
            a[0] = ra[0];
#pragma unroll
            for (int k = 0; k < KT; ++k) b[0][k] = wb[16 * k * WS];
#pragma unroll
            for (int s = 0; s < VSTEPS; ++s) {
                if (s + 1 < VSTEPS) {
                    a[(s + 1) & 1] = ra[4 * (s + 1)];
#pragma unroll
                    for (int k = 0; k < KT; ++k) b[(s + 1) & 1][k] = wb[16 * k * WS + 4 * (s + 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = mfma(a[s & 1], b[s & 1][k], acc[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 3) {  // register operands, 3 chains, 72 MFMAs
            double a = 1.0 + it, b = 2.0;
#pragma unroll
            for (int s = 0; s < VSTEPS; ++s)
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = mfma(a, b, acc[k]);
        } else if (MODE == 5) {  // the P phase with the LDS reads spread between the MFMAs of the k-step
            const double* ha = Hl + c16 * LS + q;
            const double* wb = Wl + q * WS + c16;
            double a[2], b[2][VT];
            a[0] = ha[0];
#pragma unroll
            for (int v = 0; v < VT; ++v) b[0][v] = wb[16 * v];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < KS) {
                    a[(s + 1) & 1] = ha[4 * (s + 1)];
#pragma unroll
                    for (int v = 0; v < VT; ++v) b[(s + 1) & 1][v] = wb[4 * (s + 1) * WS + 16 * v];
                }
#pragma unroll
                for (int v = 0; v < VT; ++v) acc[v] = mfma(a[s & 1], b[s & 1][v], acc[v]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 6) {  // the U phase likewise
            const double* ra = Rl + c16 * RS + q;
            const double* wb = Wl + c16 * WS + q;
            double a[2], b[2][KT];
            a[0] = ra[0];
#pragma unroll
            for (int k = 0; k < KT; ++k) b[0][k] = wb[16 * k * WS];
#pragma unroll
            for (int s = 0; s < VSTEPS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < VSTEPS) {
                    a[(s + 1) & 1] = ra[4 * (s + 1)];
#pragma unroll
                    for (int k = 0; k < KT; ++k) b[(s + 1) & 1][k] = wb[16 * k * WS + 4 * (s + 1)];
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = mfma(a[s & 1], b[s & 1][k], acc[k]);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 7) {  // U phase, reads spread, two steps ahead
            const double* ra = Rl + c16 * RS + q;
            const double* wb = Wl + c16 * WS + q;
            double a[3], b[3][KT];
#pragma unroll
            for (int s0 = 0; s0 < 2; ++s0) {
                a[s0] = ra[4 * s0];
#pragma unroll
                for (int k = 0; k < KT; ++k) b[s0][k] = wb[16 * k * WS + 4 * s0];
            }
#pragma unroll
            for (int s = 0; s < VSTEPS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 2 < VSTEPS) {
                    a[(s + 2) % 3] = ra[4 * (s + 2)];
#pragma unroll
                    for (int k = 0; k < KT; ++k) b[(s + 2) % 3][k] = wb[16 * k * WS + 4 * (s + 2)];
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = mfma(a[s % 3], b[s % 3][k], acc[k]);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 8) {  // P phase, reads spread, two steps ahead
            const double* ha = Hl + c16 * LS + q;
            const double* wb = Wl + q * WS + c16;
            double a[3], b[3][VT];
#pragma unroll
            for (int s0 = 0; s0 < 2; ++s0) {
                a[s0] = ha[4 * s0];
#pragma unroll
                for (int v = 0; v < VT; ++v) b[s0][v] = wb[4 * s0 * WS + 16 * v];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 2 < KS) {
                    a[(s + 2) % 3] = ha[4 * (s + 2)];
#pragma unroll
                    for (int v = 0; v < VT; ++v) b[(s + 2) % 3][v] = wb[4 * (s + 2) * WS + 16 * v];
                }
#pragma unroll
                for (int v = 0; v < VT; ++v) acc[v] = mfma(a[s % 3], b[s % 3][v], acc[v]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 4) {  // P phase with all operands of the phase read up front (no reads between the MFMAs)
            const double* ha = Hl + c16 * LS + q;
            const double* wb = Wl + q * WS + c16;
            double a[KS], b[KS][VT];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                a[s] = ha[4 * s];
#pragma unroll
                for (int v = 0; v < VT; ++v) b[s][v] = wb[4 * s * WS + 16 * v];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int v = 0; v < VT; ++v) acc[v] = mfma(a[s], b[s][v], acc[v]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int v = 0; v < VT; ++v) s += acc[v][0] + acc[v][1] + acc[v][2] + acc[v][3];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE>
void run(const char* name, int nm) {
    const int grid = 256, rep = 400;
    double* out; unsigned long long* ticks;
    hipMalloc(&out, grid * 256 * 8); hipMalloc(&ticks, grid * 4 * 8);
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, ticks, rep); }
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(grid * 4);
    hipMemcpy(t.data(), ticks, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(t.begin(), t.end());
    const double ns = t[t.size() / 2] * 10.0 / ((double)rep * nm);
    printf("%-58s %6.2f ns per MFMA (median wave) = %5.1f cycles at 2.4 GHz, %5.1f at 2.2\n", name, ns, ns * 2.4, ns * 2.2);
    hipFree(out); hipFree(ticks);
}
int main() {
    run<0>("registers, 6 chains (78 MFMA)", 78);
    run<1>("P phase: 7 LDS reads per 6 MFMAs, one k-step ahead", 78);
    run<4>("P phase: all 91 LDS reads up front", 78);
    run<5>("P phase: the reads spread between the MFMAs", 78);
    run<8>("P phase: spread, two k-steps ahead", 78);
    run<3>("registers, 3 chains (72 MFMA)", 72);
    run<2>("U phase: 4 LDS reads per 3 MFMAs, one step ahead", 72);
    run<6>("U phase: the reads spread between the MFMAs", 72);
    run<7>("U phase: spread, two steps ahead", 72);
    return 0;
}
