// Opt-in fast mode of the joint KL step (plan item; not the default and never the benchmarked path):
// the same fused pass as salnmf_kernels.h -- P = H W, R = X / P, G += H^T R, U = R W^T, H <- clip(H * U) per tile of 16
// samples, one wave per tile stream -- on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: 32 cycles per issue against 64
// for the fp64 form; peak 157.3 TF/s), reading fp32 shadow copies of X and H.  What stays fp64: W in HBM, the numerator
// slabs (each wave's partial is summed over its tiles in fp32, the cross-wave / cross-workgroup sums and the whole W
// tail are the fp64 code of the default path), objectives, everything outside the step.
//
// Tolerance of the mode (tests/test_gpu_fast_mode.py): W and H within 1e-5 rel-L2 of the fp64 path after 20 steps and
// within 1e-3 after 500; the north star's 1e-4 bound is the fp64 path's business.  The reference computes in fp64
// (_utils_klnmf.py:7-9, 318-361); this mode is for exploratory fits where ~1.5x more steps per second matter more.
//
// Differences from the fp64 kernel, all forced by the instruction:
//   * D layout: register r of lane (q, c16) is row 4q + r (fp64: q + 4r), column c16.  The G phase still feeds the R
//     accumulator registers straight in as the B operand: k-step r then covers samples {r, 4 + r, 8 + r, 12 + r}, and the
//     A operand (H^T from LDS) is read with the same index.
//   * LDS holds floats; bank-conflict-free strides differ per access pattern (32 banks of 4 bytes, half a wave per
//     cycle): W is kept twice, stride 112 for the P phase (rows q + 4s: 112 = 16 mod 32) and stride 98 for the U phase
//     (rows 16kt + c16: 98 = 2 mod 32); H tile stride KP + 2, R tile stride 98.
//   * all K columns go through MFMA (KT = KP / 16 output tiles); the fp64 kernel's VALU remainder columns, weights,
//     MvNMF statistics, cooperative leftover tile and persistent mode are not replicated here.
#pragma once
#include "salnmf_kernels.h"

namespace salnmf {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 mfma32(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int WSP32 = 112;  // LDS row stride (floats) of the P-phase copy of W
constexpr int WSU32 = 98;   // ... of the U-phase copy
constexpr int RS32 = 98;    // ... of the ratio tile

template <int KS>
struct Geo32 {
    static constexpr int KT = (KS + 3) / 4;
    static constexpr int KP = 16 * KT;
    static constexpr int LS = KP + 2;
    static constexpr int HL = 16 * LS, RL = 16 * RS32;
    static constexpr int HV = KP / 16;  // 16-byte loads per lane that fetch one H tile (16 x KP floats)
    static constexpr int LOOP_FLOATS = KP * WSP32 + KP * WSU32 + WAVES * (HL + RL);
    static constexpr int PARK_FLOATS = WAVES * KT * VT * 256;  // the epilogue parks the four waves' accumulators
    static constexpr int LDS_FLOATS = LOOP_FLOATS > PARK_FLOATS ? LOOP_FLOATS : PARK_FLOATS;
};

struct Fused32Params {
    const float* __restrict__ X;   // [Np][VMAX]
    float* __restrict__ H;         // [Np][KP] in / out
    const double* __restrict__ W;  // [K][V]
    double* __restrict__ Gpart;    // [grid][K][VMAX]
    int64_t N, ntiles;
    int V, K;
    float hfloor;
};

template <int KS>
__global__ void __launch_bounds__(BLOCK, 1) fused_f32_kernel(Fused32Params p) {
    using G_ = Geo32<KS>;
    constexpr int KT = G_::KT, KP = G_::KP, LS = G_::LS, HV = G_::HV;
    __shared__ __attribute__((aligned(16))) float lds[G_::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
    const int V = p.V, K = p.K;
    float* Wp = lds;
    float* Wu = lds + KP * WSP32;
    float* Hl = lds + KP * (WSP32 + WSU32) + wave * (G_::HL + G_::RL);
    float* Rl = Hl + G_::HL;

    // ---- W (fp64 in HBM) -> the two LDS images
    for (int i = tid; i < KP * VMAX; i += BLOCK) {
        const int k = i / VMAX, v = i - k * VMAX;
        // pad columns of a real row hold 1 (P > 0 there, so that 0 / P = 0 for the zero pad columns of X), pad rows 0
        const float w = k < K ? (v < V ? (float)p.W[k * V + v] : 1.0f) : 0.0f;
        Wp[k * WSP32 + v] = w;
        Wu[k * WSU32 + v] = w;
    }
    for (int i = tid; i < KP * (WSP32 - VMAX); i += BLOCK) {  // the P-phase image is read up to column 95 only; keep the pad defined
        const int k = i / (WSP32 - VMAX), v = VMAX + i - k * (WSP32 - VMAX);
        Wp[k * WSP32 + v] = 0.0f;
    }
    __syncthreads();

    f4 g[KT][VT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) g[kt][vt] = (f4){0, 0, 0, 0};

    const int64_t tstride = (int64_t)gridDim.x * WAVES;
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;

    // lane's slice of an H tile: 4 consecutive floats e = 4*lane + 256*j of the contiguous [16][KP] block
    int hrow[HV], hcol[HV];
#pragma unroll
    for (int j = 0; j < HV; ++j) {
        const int e = 4 * lane + 256 * j;
        hrow[j] = e / KP;
        hcol[j] = e - hrow[j] * KP;
    }
    f4 hpre[HV];
    float x[VT][4];
    auto load_tile = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t n0 = t * 16;
        const f4* hsrc = reinterpret_cast<const f4*>(p.H + n0 * KP) + lane;
#pragma unroll
        for (int j = 0; j < HV; ++j) hpre[j] = hsrc[64 * j];
        const float* xsrc = p.X + (n0 + 4 * q) * VMAX + c16;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[vt][r] = xsrc[r * VMAX + 16 * vt];
    };

    if (tile < p.ntiles) load_tile(tile);
    for (; tile < p.ntiles; tile += tstride) {
        const int64_t n0 = tile * 16;
        // ---- stage the H tile (wave private)
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            float* dst = Hl + hrow[j] * LS + hcol[j];  // 8-byte aligned (LS even, hcol a multiple of 4)
            *reinterpret_cast<f2*>(dst) = (f2){hpre[j][0], hpre[j][1]};
            *reinterpret_cast<f2*>(dst + 2) = (f2){hpre[j][2], hpre[j][3]};
        }
        __builtin_amdgcn_wave_barrier();

        // ---- P = H W   (A = H[n=c16][k=4s+q], B = W[k=4s+q][v=16vt+c16]); D: rows 4q + r
        f4 pr[VT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) pr[vt] = (f4){0, 0, 0, 0};
        {
            const float* ha = Hl + c16 * LS + q;
            const float* wb = Wp + q * WSP32 + c16;
            float a[2], b[2][VT];
            a[0] = ha[0];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) b[0][vt] = wb[16 * vt];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < KS) {
                    a[(s + 1) & 1] = ha[4 * (s + 1)];
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) b[(s + 1) & 1][vt] = wb[4 * (s + 1) * WSP32 + 16 * vt];
                }
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) pr[vt] = mfma32(a[s & 1], b[s & 1][vt], pr[vt]);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // G-phase A operands (H^T): samples 4q + r of k-step r
        float ga[4][KT];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* ha = Hl + (4 * q + r) * LS + c16;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) ga[r][kt] = ha[16 * kt];
        }
        // ---- R = X / P in place (v_rcp_f32 is good to 1 ulp; pads give 0 * 1/P = 0)
        // (reciprocals first, products second, and all of it before the G phase: see the fp64 kernel's division chains)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[vt][r] = __builtin_amdgcn_rcpf(pr[vt][r]);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[vt][r] *= x[vt][r];
        asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]));
        if (tile + tstride < p.ntiles) load_tile(tile + tstride);
        // ---- transpose R through LDS for the U phase
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Rl[(4 * q + r) * RS32 + 16 * vt + c16] = pr[vt][r];
        // ---- G += H^T R   (A = H[n=4q+r][k=16kt+c16], B = register r of R)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) g[kt][vt] = mfma32(ga[r][kt], pr[vt][r], g[kt][vt]);
        __builtin_amdgcn_wave_barrier();
        // ---- U = R W^T   (A = R[n=c16][v=4s+q], B = W[k=16kt+c16][v=4s+q]); D: rows 4q + r, columns 16kt + c16
        f4 u[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) u[kt] = (f4){0, 0, 0, 0};
        float hcur[4][KT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) hcur[r][kt] = Hl[(4 * q + r) * LS + 16 * kt + c16];
        {
            const float* ra = Rl + c16 * RS32 + q;
            const float* wb = Wu + c16 * WSU32 + q;
            float a[2], b[2][KT];
            a[0] = ra[0];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b[0][kt] = wb[16 * kt * WSU32];
#pragma unroll
            for (int s = 0; s < VSTEPS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < VSTEPS) {
                    a[(s + 1) & 1] = ra[4 * (s + 1)];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) b[(s + 1) & 1][kt] = wb[16 * kt * WSU32 + 4 * (s + 1)];
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) u[kt] = mfma32(a[s & 1], b[s & 1][kt], u[kt]);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- H update (_utils_klnmf.py:343-347), unmasked: pad rows / columns receive finite filler
        float* hdst = p.H + (n0 + 4 * q) * KP + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const float hn = hcur[r][kt] * u[kt][r];
                __builtin_nontemporal_store(hn < p.hfloor ? p.hfloor : hn, &hdst[r * KP + 16 * kt]);  // a NaN stays a NaN
            }
    }

    // ---- numerator slab: the four waves' accumulators summed in fp64, fixed order
    __syncthreads();
    float* park = lds + wave * (KT * VT * 256);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) park[((kt * VT + vt) * 4 + r) * 64 + lane] = g[kt][vt][r];
    __syncthreads();
    double* out = p.Gpart + (int64_t)blockIdx.x * K * VMAX;
    for (int i = tid; i < KT * VT * 256; i += BLOCK) {
        const int t = i >> 8, r = (i >> 6) & 3, l = i & 63;
        const int kt = t / VT, vt = t - kt * VT;
        const int k = 16 * kt + 4 * (l >> 4) + r, v = 16 * vt + (l & 15);
        if (k < K && v < V) {
            const double s = (((double)lds[i] + (double)lds[KT * VT * 256 + i]) + (double)lds[2 * KT * VT * 256 + i]) + (double)lds[3 * KT * VT * 256 + i];
            out[k * VMAX + v] = s;
        }
    }
}

#ifndef SALNMF_TEMPLATES_ONLY
// fp64 <-> fp32 images of the padded device arrays (X: [Np][VMAX], H: [Np][KP])
__global__ void cvt_f64_f32_kernel(float* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ void cvt_f32_f64_kernel(double* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

#endif  // SALNMF_TEMPLATES_ONLY

}  // namespace salnmf
