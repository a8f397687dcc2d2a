// Dense pieces of correlated NMF on the padded device layout of the KL-NMF
// engine: exposures H = exp(beta_k + alpha_n + <L_k, U_n>), both scaling updates, column sums
// of aux, and the constant sum gammaln(1 + X) of the Poisson log-likelihood.
// Reference arithmetic: src/salamander/models/_utils_corrnmf.py:11-25 (compute_exposures),
// :103-138 (update_signature_scalings), :141-179 (update_sample_scalings);
// _utils_klnmf.py:98-160 (poisson_llh).  aux itself (:28-52) is the U phase of fused_kernel
// with an H-multiply epilogue, see salnmf_corr_compute_aux in salnmf_host_corr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "salnmf_kernels.h"
#include "salnmf_newtoncg.h"
#include "salnmf_corr_params.h"

namespace salnmf {

constexpr int CORR_DMAX = 64;   // dim_embeddings <= n_signatures <= 64
constexpr int CORR_LD = 65;     // odd LDS stride: rows of L / of the logit tile fall into different banks
constexpr int CORR_TILE = 64;   // samples per workgroup pass (one per lane)
constexpr int CORR_BLOCK = 256; // 4 waves: wave g handles signatures k = g, g+4, ...
constexpr int CORR_SLOTS = 16;  // ceil(64 / 4) signatures per thread

struct CorrParams {
    const double* __restrict__ alpha;    // [Np] sample scalings             (modes 1, 2)
    const double* __restrict__ beta;     // [K]  signature scalings          (modes 0, 1)
    const double* __restrict__ L;        // [K][dim] signature embeddings
    const double* __restrict__ U;        // [N][dim] sample embeddings (compact)
    const double* __restrict__ xrowsum;  // [Np] sum_v X[n][v]               (mode 0)
    double* __restrict__ out;            // mode 0: alpha [Np]; mode 1: H [Np][KP]; mode 2: partial [gridDim.x][K]
    int64_t N, Np;
    int K, KP, dim;
};

// One pass over the logits  S[n][k] = <L_k, U_n>  followed by
//   MODE 0: alpha_n = log(sum_v X[n][v]) - log(sum_k exp(beta_k + S[n][k]))        (:170-179)
//   MODE 1: H[n][k] = exp((beta_k + alpha_n) + S[n][k]), pad rows 1, pad columns 0   (:21-25)
//   MODE 2: partial_k = sum over this workgroup's samples of exp(alpha_n + S[n][k])  (:134-136)
// The logits on the fp64 MFMA units: S = U . L^T per 16-sample tile of a wave (A[i = sample c16][k = component 4 ks + q]
// straight from global memory, one tile ahead; B[k][j = signature 16 jt + c16] from an LDS copy of L^T; D[row = sample
// q + 4 r][col = signature c16]).  The tile of logits goes through a wave-private LDS tile and the exponentials run in
// ROLLED loops over its elements, with the tile's per-sample scalars in LDS as well (a global load inside such a loop is a
// round trip per iteration).  The VALU form this replaces (lane = sample, every row read once per wave of the workgroup,
// FMAs against LDS broadcasts) took 90-120 us per pass at c5's 200 000 x 40 x 40; this one 70-90 us with the two
// modalities' passes running side by side.
//   mode 0 / 2 sums (over signatures / over samples) are taken in a fixed order of this kernel's own.
constexpr int CLM_LD = 80;  // row stride of the LDS copy of L^T: 160 dwords = 32 banks mod 64 -> the two k rows of a half-wave do not collide
// KSQ: k-steps of the product, dim rounded up to 16 components (compile-time, and all four signature tiles always: an MFMA
// behind a uniform branch of its own costs the register allocator ~100 accumulator copies -- 6 000 of the kernel's 8 000
// instructions with runtime counts)
template <int MODE, int KSQ>
__global__ void __launch_bounds__(CORR_BLOCK) corr_logit_mfma_kernel(CorrParams p) {
    __shared__ double Lt[CORR_DMAX * CLM_LD];  // L^T[component][signature], zero filled
    __shared__ double Tt[4][16 * CORR_LD];      // the waves' tiles of logits [sample][signature]
    __shared__ double bl[CORR_DMAX];
    __shared__ double arow[4][16];              // the tile's sample scalings (modes 1, 2) / row sums of X (mode 0)
    __shared__ double wsum[4 * 64];             // (mode 2) the waves' column sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
    const int K = p.K, dim = p.dim;
    // (L^T: ten unconditional loads in flight per lane and round, masked afterwards -- one load per loop iteration is a memory
    // round trip per iteration, twenty of them in front of every workgroup's first tile)
    for (int i0 = tid; i0 < CORR_DMAX * CLM_LD; i0 += 10 * CORR_BLOCK) {
        double v[10];
        bool in[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int i = i0 + u * CORR_BLOCK, m = i / CLM_LD, k = i - m * CLM_LD;
            in[u] = i < CORR_DMAX * CLM_LD && k < K && m < dim;
            v[u] = p.L[in[u] ? k * dim + m : 0];
        }
#pragma unroll
        for (int u = 0; u < 10; ++u)
            if (i0 + u * CORR_BLOCK < CORR_DMAX * CLM_LD) Lt[i0 + u * CORR_BLOCK] = in[u] ? v[u] : 0.0;
    }
    if (tid < CORR_DMAX) bl[tid] = (MODE != 2 && tid < K) ? p.beta[tid] : 0.0;
    __syncthreads();
    const double* lb = Lt + q * CLM_LD + c16;
    double* T = Tt[wave];
    double colacc = 0.0;  // (mode 2) lane k: the column's sum over this wave's samples
    const int64_t ntiles = p.Np / 16, tstride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    // A operand of a tile: row n0 + c16, components 4 ks + q (clamped addresses; rows beyond N and components beyond dim
    // are masked when used)
    double a_next[KSQ];
    auto load_a = [&](int64_t t) {
        const int64_t n = t * 16 + c16, nc = n < p.N ? n : p.N - 1;
        const double* row = p.U + nc * dim;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) a_next[ks] = row[4 * ks + q < dim ? 4 * ks + q : dim - 1];
    };
    if (tile < ntiles) load_a(tile);
    for (; tile < ntiles; tile += tstride) {
        const int64_t n0 = tile * 16;
        const bool rowlive = n0 + c16 < p.N;
        d4 S[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) S[jt] = (d4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
            const double a = (rowlive && 4 * ks + q < dim) ? a_next[ks] : 0.0;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) S[jt] = mfma(a, lb[4 * ks * CLM_LD + 16 * jt], S[jt]);
        }
        // (the tile's per-sample scalars through LDS: a global load inside the rolled loops below is a round trip per iteration)
        {
            const int64_t n = n0 + c16, nc = n < p.N ? n : p.N - 1;
            const double v = MODE == 0 ? p.xrowsum[nc] : p.alpha[nc];
            if (q == 0) arow[wave][c16] = v;
        }
        if (tile + tstride < ntiles) load_a(tile + tstride);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(q + 4 * r) * CORR_LD + 16 * jt + c16] = S[jt][r];
        __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations execute in order)
        if (MODE == 0) {
            // alpha_n = log(sum_v X[n][v]) - log(sum_k exp(beta_k + S[n][k])): four lanes per sample (signatures sub, sub + 4,
            // ... in order), then the four
            const int row = lane >> 2, sub = lane & 3;
            double part = 0.0;
#pragma unroll 1
            for (int k = sub; k < K; k += 4) part += exp(bl[k] + T[row * CORR_LD + k]);
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            const int64_t n = n0 + row;
            if (sub == 0) p.out[n] = n < p.N ? log(arow[wave][row]) - log(part) : 0.0;
        } else if (MODE == 1) {
            // H[n][k] = exp((beta_k + alpha_n) + S[n][k]); pad rows 1, pad columns 0; whole rows of the tile, coalesced
            const int KP = p.KP, total = 16 * KP;
#pragma unroll 1
            for (int e = lane; e < total; e += 64) {
                const int row = e / KP, col = e - row * KP;
                const int64_t n = n0 + row;
                double h = 0.0;
                if (col < K) h = exp((bl[col] + arow[wave][row]) + T[row * CORR_LD + col]);
                if (n >= p.N) h = 1.0;
                p.out[n * KP + col] = h;
            }
        } else {
            // partial_k += sum over the tile's samples of exp(alpha_n + S[n][k]): lane k, the samples in order
            if (lane < K) {
#pragma unroll 1
                for (int row = 0; row < 16; ++row)
                    if (n0 + row < p.N) colacc += exp(arow[wave][row] + T[row * CORR_LD + lane]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // (the tile is rewritten by the next product's stores)
    }
    if (MODE == 2) {
        wsum[wave * 64 + lane] = colacc;
        __syncthreads();
        if (tid < K) p.out[(int64_t)blockIdx.x * K + tid] = ((wsum[tid] + wsum[64 + tid]) + wsum[128 + tid]) + wsum[192 + tid];
    }
}

// out[n] = sum_v X[n][v] on the padded layout [Np][96] (pads are 0): 16 lanes per row, 6 features each
__global__ void __launch_bounds__(256) rowsum_X_kernel(const double* __restrict__ X, int64_t Np, int ldx, int add, double* __restrict__ out) {
    const int c16 = threadIdx.x & 15;
    int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t stride = (int64_t)gridDim.x * 16;
    for (; row < Np; row += stride) {  // Np is a multiple of 16: all 16-lane groups of a wave stay in step
        double s = 0.0;
        for (int v = c16; v < ldx; v += 16) s += X[row * ldx + v];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
        if (c16 == 0) out[row] = add ? out[row] + s : s;
    }
}

// partial[b][c] = sum over the rows n < N of workgroup b's contiguous chunk of A[n][c]  (A is [.][ld], ld <= 64)
__global__ void __launch_bounds__(256) colsum_partial_kernel(const double* __restrict__ A, int64_t N, int ld,
                                                             double* __restrict__ partial) {
    __shared__ double red[256];
    const int groups = 256 / ld;  // row groups per pass
    const int c = threadIdx.x % ld, g = threadIdx.x / ld;
    const int64_t chunk = (N + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < N) ? lo + chunk : N;
    double s = 0.0;
    if (g < groups) {
        int64_t n = lo + g;
        for (; n + 7 * groups < hi; n += 8 * groups) {  // (eight loads in flight per round, added in the same order)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = A[(n + (int64_t)u * groups) * ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; n < hi; n += groups) s += A[n * ld + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0) {
        double t = 0.0;
        for (int i = 0; i < groups; ++i) t += red[i * ld + c];
        partial[(int64_t)blockIdx.x * ld + c] = t;
    }
}

// partial[b] = sum over workgroup b's grid-stride slice of gammaln(1 + X[n][v]), n < N, v < V
__global__ void __launch_bounds__(256) lgamma_partial_kernel(const double* __restrict__ X, int64_t N, int V, int ldx,
                                                             double* __restrict__ partial) {
    __shared__ double red[256];
    double s = 0.0;
    const int64_t total = N * ldx;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % ldx);
        if (v < V) s += lgamma(1.0 + X[i]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// partial[b] = sum over workgroup b's grid-stride slice of a[i]^2 (fixed order; the variance update and the
// Gaussian prior terms of the ELBO, corrnmf_det.py:65-69 / _utils_corrnmf.py:93-98)
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const double* __restrict__ a, int64_t n, double* __restrict__ partial) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s = __builtin_fma(a[i], a[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// beta_k = log(first_k) - log(second_k)   (_utils_corrnmf.py:137)
__global__ void corr_log_ratio_kernel(const double* __restrict__ first, const double* __restrict__ second, int K,
                                      double* __restrict__ beta) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) beta[k] = log(first[k]) - log(second[k]);
}

// ----------------------------------------------------------------------------------------------
// Sample embeddings (CorrNMFDet.update_sample_embeddings, corrnmf_det.py:115-141;
// MultimodalCorrNMF.update_sample_embeddings, mmcorrnmf.py:398-428): one strictly convex problem per
// sample n over u in R^dim whose terms are the signatures of all modalities,
//   minimise  -[ sum_i aux_i[n] <L_i, u> - sum_i exp(alpha_{mod(i)}[n] + beta_i + <L_i, u>) - |u|^2 / (2 var) ]
// (_utils_corrnmf.py:182-239; gradient :242-293, Hessian :296-351), solved by the Newton-CG of
// salnmf_newtoncg.h with maxiter = 3 (corrnmf_det.py:140, mmcorrnmf.py:427) from the current
// embedding; finally entries within EPSILON of zero are pushed to +-EPSILON (_utils_corrnmf.py:408-409).
// One wavefront per sample: lane m <-> embedding component m, lane l <-> terms l (and l + 64 when
// the modalities have more than 64 signatures in total, TPL = 2).

template <int TPL>
struct SampleEmbeddingEval {
    const double* O;   // LDS [terms][ld]: the signature embeddings of all modalities
    int ld;            // its row stride: dim rounded up to an odd number (rows fall into different banks)
    const double* so;  // LDS [terms]: signature scalings
    double c[TPL];     // sample scaling of the modality of this lane's term(s)
    double a[TPL];     // aux of this lane's term(s) for this sample
    double hw[TPL];    // exp(c + so + <L_i, x>) at the point the Hessian is fixed
    double sg;         // lane m: sum_i aux_i L[i][m]
    double variance;
    int T, dim, lane;  // T = number of terms

    // component m of a vector held one component per lane (wave-uniform m): two v_readlane_b32 instead of a broadcast
    // read from LDS.  The kernel is bound by the LDS pipe (profiles/r03/c5_roofline.md: LDS busy 13 of 16.9 ms per call,
    // VALU 5); a broadcast read costs the pipe as much as any other 512-byte wave read, a readlane costs it nothing.
    // Same summation orders as before (four partial sums by index mod 4): same bits.
    // Measured and NOT adopted on top of this (profiles/r03/c5_roofline.md): combine's column of L in registers (48 terms:
    // the 256 registers that two waves per SIMD leave each wave hold no more without spilling) -- 15.1 instead of
    // 11.2 ms per call; a dense per-sample Hessian formed on the MFMA units once per Newton iteration -- 18.9 ms (its
    // LDS staging halves the occupancy, and the Gram product costs more than the few CG iterations it serves).
    static __device__ __forceinline__ double lane_value(double v, int m) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), m), __builtin_amdgcn_readlane(__double2loint(v), m));
    }
    // <L_i, y> for this lane's term(s)
    __device__ __forceinline__ void products(double y, double (&s)[TPL]) {
        const double* row[TPL];
#pragma unroll
        for (int t = 0; t < TPL; ++t) row[t] = O + ((lane + 64 * t < T) ? lane + 64 * t : 0) * ld;
        // four partial sums (m mod 4) per term: one chain of dim dependent FMAs would wait out every FMA's latency
        double a[TPL][4];
#pragma unroll
        for (int t = 0; t < TPL; ++t) a[t][0] = a[t][1] = a[t][2] = a[t][3] = 0.0;
        int m = 0;
        for (; m + 4 <= dim; m += 4) {
            double yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) yv[u] = lane_value(y, m + u);
#pragma unroll
            for (int t = 0; t < TPL; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) a[t][u] = __builtin_fma(row[t][m + u], yv[u], a[t][u]);
        }
        for (; m < dim; ++m) {
            const double yv = lane_value(y, m);
#pragma unroll
            for (int t = 0; t < TPL; ++t) a[t][0] = __builtin_fma(row[t][m], yv, a[t][0]);
        }
#pragma unroll
        for (int t = 0; t < TPL; ++t) s[t] = (lane + 64 * t < T) ? (a[t][0] + a[t][1]) + (a[t][2] + a[t][3]) : 0.0;
    }
    // lane m: sum_i w_i L[i][m]
    __device__ __forceinline__ double combine(const double (&w)[TPL]) {
        const double* col = O + (lane < dim ? lane : 0);
        double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;  // (four partial sums, as in products)
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const int n = T - 64 * t < 64 ? T - 64 * t : 64;  // terms held by this register of the lanes
            const double* c = col + 64 * t * ld;
            int i = 0;
            for (; i + 4 <= n; i += 4) {
                r0 = __builtin_fma(lane_value(w[t], i), c[i * ld], r0);
                r1 = __builtin_fma(lane_value(w[t], i + 1), c[(i + 1) * ld], r1);
                r2 = __builtin_fma(lane_value(w[t], i + 2), c[(i + 2) * ld], r2);
                r3 = __builtin_fma(lane_value(w[t], i + 3), c[(i + 3) * ld], r3);
            }
            for (; i < n; ++i) r0 = __builtin_fma(lane_value(w[t], i), c[i * ld], r0);
        }
        return lane < dim ? (r0 + r1) + (r2 + r3) : 0.0;
    }
    __device__ __forceinline__ double rate(int t, double s) const {
        const int i = lane + 64 * t;
        return i < T ? exp((c[t] + so[i]) + s) : 0.0;
    }
    __device__ inline double fun(double y) {
        double s[TPL];
        products(y, s);
        double lin = 0.0, ex = 0.0;
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            lin += (lane + 64 * t < T) ? s[t] * a[t] : 0.0;
            ex += rate(t, s[t]);
        }
        double v = ncg::wave_sum(lin);
        v -= ncg::wave_sum(ex);
        v -= ncg::wave_sum(y * y) / (2 * variance);
        return -v;
    }
    __device__ inline double grad(double y) {
        double s[TPL], w[TPL];
        products(y, s);
#pragma unroll
        for (int t = 0; t < TPL; ++t) w[t] = rate(t, s[t]);
        double g = -combine(w);
        g += sg;
        g -= y / variance;
        return lane < dim ? -g : 0.0;
    }
    __device__ inline void fun_grad(double y, double& f, double& g) {
        double s[TPL], w[TPL];
        products(y, s);
        double lin = 0.0, ex = 0.0;
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            w[t] = rate(t, s[t]);
            lin += (lane + 64 * t < T) ? s[t] * a[t] : 0.0;
            ex += w[t];
        }
        double v = ncg::wave_sum(lin);
        v -= ncg::wave_sum(ex);
        v -= ncg::wave_sum(y * y) / (2 * variance);
        f = -v;
        double gg = -combine(w);
        gg += sg;
        gg -= y / variance;
        g = lane < dim ? -gg : 0.0;
    }
    __device__ inline void prepare_hess(double x) {
        double s[TPL];
        products(x, s);
#pragma unroll
        for (int t = 0; t < TPL; ++t) hw[t] = rate(t, s[t]);
    }
    __device__ inline double hessp(double p) {
        double s[TPL], w[TPL];
        products(p, s);
#pragma unroll
        for (int t = 0; t < TPL; ++t) w[t] = hw[t] * s[t];
        const double r = combine(w);
        return lane < dim ? r + p / variance : 0.0;
    }
    __device__ inline bool exhausted() const { return false; }  // every loop of the solve is bounded and cheap here
};


template <int TPL>
__global__ void __launch_bounds__(CORR_BLOCK) corr_sample_embeddings_kernel(SampleEmbeddingParams p) {
    // the term matrix in LDS, sized by the problem (T rows of `ld` doubles): at c5 (80 terms, dim 40) 26 KB instead of the
    // 66.5 KB of the largest case, which is what lets three workgroups share a CU (the kernel is a chain of dependent
    // evaluations per wave: more waves per SIMD hide more of its latency)
    extern __shared__ __attribute__((aligned(16))) double Ll[];
    __shared__ double bl[64 * TPL];
    __shared__ int tmod[64 * TPL], tk[64 * TPL];  // term -> (modality, signature)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dim = p.dim;
    int T = 0;
    for (int mo = 0; mo < p.n_mod; ++mo) T += p.K[mo];
    for (int i = tid; i < 64 * TPL; i += CORR_BLOCK) {
        int mo = 0, k = i;
        while (mo < p.n_mod && k >= p.K[mo]) { k -= p.K[mo]; ++mo; }
        tmod[i] = mo < p.n_mod ? mo : 0;
        tk[i] = mo < p.n_mod ? k : 0;
        bl[i] = mo < p.n_mod ? p.beta[mo][k] : 0.0;
    }
    __syncthreads();
    const int ld = dim | 1;  // (odd; the column beyond dim, if any, is zero)
    for (int i = tid; i < T * ld; i += CORR_BLOCK) {
        const int t = i / ld, m = i - t * ld;
        Ll[i] = m < dim ? p.L[tmod[t]][tk[t] * dim + m] : 0.0;
    }
    __syncthreads();
    // from here on the waves run independently (no workgroup barrier below)
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < p.N; n += (int64_t)gridDim.x * 4) {
        SampleEmbeddingEval<TPL> ev;
        ev.O = Ll;
        ev.ld = ld;
        ev.so = bl;
        ev.variance = p.variance;
        ev.T = T;
        ev.dim = dim;
        ev.lane = lane;
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const int i = lane + 64 * t;
            const bool live = i < T;
            const int mo = tmod[i];
            ev.c[t] = live ? p.alpha[mo][n] : 0.0;
            ev.a[t] = live ? p.aux[mo][n * p.KP[mo] + tk[i]] : 0.0;
            ev.hw[t] = 0.0;
        }
        ev.sg = ev.combine(ev.a);
        double x = lane < dim ? p.U[n * dim + lane] : 0.0;
        const int st = ncg::minimize(ev, x, dim, p.maxiter);
        if (x > 0.0 && x < kEps) x = kEps;
        if (x < 0.0 && x > -kEps) x = -kEps;
        if (lane < dim) p.U[n * dim + lane] = x;
        if (p.status && lane == 0) p.status[n] = st;
    }
}

// ----------------------------------------------------------------------------------------------
// Signature embeddings (CorrNMFDet.update_signature_embeddings, corrnmf_det.py:88-113): one problem
// per signature k over l in R^dim whose terms run over ALL samples,
//   minimise  -[ sum_n aux[k][n] <U_n, l> - sum_n exp(beta_k + alpha_n + <U_n, l>) - |l|^2 / (2 var) ]
// One WORKGROUP per signature.  Every wave runs the Newton-CG control flow of salnmf_newtoncg.h
// redundantly on bit-identical scalars (vectors are replicated per wave, lane m = component m); an
// evaluation is a cooperative pass over the samples in tiles of SIGT rows staged in LDS:
//   phase A  thread j: s_j = <U_j, y>, weight_j (exp / aux)
//   phase B  wave w, lane m: sum over the tile's samples j = w, w + 4, ... of weight_j U[j][m]
// followed by fixed-order cross-wave sums, so that all waves see the same bits.  The Hessian
// sum_n w_n U_n U_n^T + I / var is formed densely ONCE per Newton iteration (as SciPy does with hess=) on the
// fp64 MFMA units -- per tile a (dim x 256) . (256 x dim) product -- and kept in LDS, so that the
// conjugate-gradient iterations need no pass over the samples at all.
constexpr int SIGT = 256;       // samples per tile = threads per workgroup
constexpr int SIG_BUDGET = 60000;  // evaluation passes per solve; SciPy's own limits are far above any real solve

struct SignatureEmbeddingParams {
    const double* __restrict__ aux;    // [Np][KP]
    const double* __restrict__ alpha;  // [Np]
    const double* __restrict__ beta;   // [K]
    const double* __restrict__ U;      // [N][dim]
    double* __restrict__ L;            // [K][dim]  in / out
    int* __restrict__ status;          // [K] or null
    const int* __restrict__ only;      // [K] or null: solve only the signatures with only[k] == ONLY_VALUE (lockstep fallback)
    double variance;
    int64_t N, Np;
    int K, KP, dim, maxiter;
};
constexpr int SIG_ONLY_VALUE = 2;

struct SignatureEmbeddingEval {
    const SignatureEmbeddingParams* p;
    double* Ut;     // LDS [SIGT][ldu]: the staged tile of U; after a Hessian pass also 4 staging copies [64][CORR_LD]
    double* Al;     // LDS [64][CORR_LD]: the dense Hessian sum (a region of its own when spec, else the start of Ut)
    double* wt;     // LDS [SIGT]
    double* ybuf;   // LDS [64]
    double* red;    // LDS [4][64]
    double* sred;   // LDS [SIGT]
    double c;       // beta_k
    double sg;      // lane m: sum_n aux[k][n] U[n][m]
    double variance;
    double hpt;     // lane m: component of the point the Hessian in Al belongs to
    bool hvalid;    // Al holds the Hessian at hpt
    bool spec;      // Al does not alias the tile buffer: every objective+gradient pass also forms the Hessian
    int k, dim, ldu, DT, tid, lane, wave, budget;
    int64_t n_begin, n_end;  // the samples this workgroup passes over (all of them, or one chunk: lockstep solves)

    __device__ inline void broadcast(double y) {
        __syncthreads();  // previous readers of ybuf are done
        if (wave == 0) ybuf[lane] = y;
        __syncthreads();
    }
    // stage rows [t0, t0 + SIGT) of U into LDS (coalesced), zero beyond N; loads issued in batches of 8
    __device__ inline void stage(int64_t t0) {
        const int64_t base = t0 * dim, end = p->N * dim;
        const int total = SIGT * dim;
        for (int i0 = tid; i0 < total; i0 += 8 * SIGT) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * SIGT;
                v[u] = (i < total && base + i < end) ? p->U[base + i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * SIGT;
                if (i < total) {
                    const int j = i / dim, m = i - j * dim;
                    Ut[j * ldu + m] = v[u];
                }
            }
        }
        // the MFMA tiles of the Hessian read component columns up to the next multiple of 16: keep them zero
        const int dpad = 16 * DT - dim;
        for (int i = tid; i < SIGT * dpad; i += SIGT) {
            const int j = i / dpad, m = dim + (i - j * dpad);
            Ut[j * ldu + m] = 0.0;
        }
        __syncthreads();
    }
    __device__ inline double row_dot(int j) const {
        double s = 0.0;
        const double* row = Ut + j * ldu;
        for (int m = 0; m < dim; ++m) s = __builtin_fma(row[m], ybuf[m], s);
        return s;
    }
    // lane m of every wave: the tile's contribution  sum_j wt[j] * U[j][m]  of this wave's samples j = wave, wave + 4, ...
    __device__ inline double tile_weighted(double r) const {
        if (lane < dim) {
#pragma unroll 4
            for (int j = wave; j < SIGT; j += 4) r = __builtin_fma(wt[j], Ut[j * ldu + lane], r);
        }
        return r;
    }
    // fixed-order sum of one per-lane value over the 4 waves; every wave returns the same bits
    __device__ inline double cross_wave(double r) {
        red[wave * 64 + lane] = r;
        __syncthreads();
        const double tot = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
        __syncthreads();
        return lane < dim ? tot : 0.0;
    }
    // deterministic workgroup sum of one value per thread; every thread returns the same bits
    __device__ inline double block_sum(double v) {
        sred[tid] = v;
        __syncthreads();
        for (int h = SIGT / 2; h > 0; h >>= 1) {
            if (tid < h) sred[tid] += sred[tid + h];
            __syncthreads();
        }
        const double t = sred[0];
        __syncthreads();
        return t;
    }
    // ---- dense Hessian  sum_n w_n U[n][m] U[n][j]  on the fp64 MFMA units, accumulated over the tiles of a pass.
    // Every wave multiplies its 64 staged samples: A operand U^T (component x sample), B operand diag(w) U
    // (sample x component) -- the same LDS elements, the B side scaled by the weight in wt.
    __device__ inline void hess_tile(d4 (&acc)[10]) const {
        const int c16 = lane & 15, q = lane >> 4;
        const double* base = Ut + (64 * wave + q) * ldu + c16;
        // operands one step ahead of the MFMAs that use them: the LDS latency of a step's reads used to be exposed in
        // front of its six MFMAs, sixteen times per tile
        double wv[2], a[2][4];
        wv[0] = wt[64 * wave + q];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[0][t] = t < DT ? base[16 * t] : 0.0;
#pragma unroll
        for (int sgrp = 0; sgrp < 16; ++sgrp) {  // 4 samples per MFMA step
            const int cur = sgrp & 1, nxt = cur ^ 1;
            if (sgrp + 1 < 16) {
                wv[nxt] = wt[64 * wave + 4 * (sgrp + 1) + q];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[nxt][t] = t < DT ? base[4 * (sgrp + 1) * ldu + 16 * t] : 0.0;
            }
            double b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = wv[cur] * a[cur][t];
            int idx = 0;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = mt; nt < 4; ++nt, ++idx)
                    if (nt < DT) acc[idx] = mfma(a[cur][mt], b[nt], acc[idx]);  // uniform over the workgroup
        }
    }
    // after the last tile: cross-wave sum in fixed order into Al; the tile buffer is free and serves as staging
    __device__ inline void hess_finish(const d4 (&acc)[10], double at) {
        const int c16 = lane & 15, q = lane >> 4;
        double* mine = Ut + wave * (16 * DT) * CORR_LD;
        int idx = 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = mt; nt < 4; ++nt, ++idx)
                if (nt < DT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + q + 4 * r, col = 16 * nt + c16;
                        mine[row * CORR_LD + col] = acc[idx][r];
                        if (mt != nt) mine[col * CORR_LD + row] = acc[idx][r];  // lower triangle = mirror
                    }
                }
        __syncthreads();
        const int n = 16 * DT * CORR_LD;
        for (int i = tid; i < n; i += SIGT) {
            const double t = ((Ut[i] + Ut[n + i]) + Ut[2 * n + i]) + Ut[3 * n + i];
            Al[i] = t;  // Al == Ut when not spec: element i is read and written by this thread only
        }
        __syncthreads();
        hpt = at;
        hvalid = spec;  // an aliased Al does not survive the next pass
    }

    // lane m of every wave: sum over all samples of weight_n * U[n][m];  MODE 0: weight aux[k][n] (summand_grad),
    // MODE 1: weight exp((c + alpha_n) + <U_n, y>) (gradient)
    template <int MODE>
    __device__ inline double weighted_sum(double y) {
        --budget;
        if (MODE != 0) broadcast(y);
        double r = 0.0;
        for (int64_t t0 = n_begin; t0 < n_end; t0 += SIGT) {
            stage(t0);
            const int64_t n = t0 + tid;
            double w = 0.0;
            if (n < n_end) w = (MODE == 0) ? p->aux[n * p->KP + k] : exp((c + p->alpha[n]) + row_dot(tid));
            wt[tid] = w;
            __syncthreads();
            r = tile_weighted(r);
            __syncthreads();
        }
        return cross_wave(r);
    }
    __device__ inline double fun(double y) {
        --budget;
        broadcast(y);
        double lin = 0.0, ex = 0.0;
        for (int64_t t0 = n_begin; t0 < n_end; t0 += SIGT) {
            stage(t0);
            const int64_t n = t0 + tid;
            if (n < n_end) {
                const double s = row_dot(tid);
                lin = __builtin_fma(s, p->aux[n * p->KP + k], lin);
                ex += exp((c + p->alpha[n]) + s);
            }
            __syncthreads();
        }
        double v = block_sum(lin);
        v -= block_sum(ex);
        v -= ncg::wave_sum(y * y) / (2 * variance);
        return -v;
    }
    __device__ inline double grad(double y) {
        double g = -weighted_sum<1>(y);
        g += sg;
        g -= y / variance;
        return lane < dim ? -g : 0.0;
    }
    // Objective and gradient in ONE pass over the samples (a line-search evaluation needs both).  The weights of
    // the gradient are those of the Hessian at the same point, and the first trial point of a line search is
    // usually accepted: when Al has a region of its own, the pass also forms the Hessian there, and the
    // prepare_hess of the next Newton iteration finds it ready.
    __device__ inline void fun_grad(double y, double& f, double& g) {
        --budget;
        broadcast(y);
        double lin = 0.0, ex = 0.0, r = 0.0;
        d4 acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = (d4){0, 0, 0, 0};
        for (int64_t t0 = n_begin; t0 < n_end; t0 += SIGT) {
            stage(t0);
            const int64_t n = t0 + tid;
            double w = 0.0;
            if (n < n_end) {
                const double s = row_dot(tid);
                lin = __builtin_fma(s, p->aux[n * p->KP + k], lin);
                w = exp((c + p->alpha[n]) + s);
                ex += w;
            }
            wt[tid] = w;
            __syncthreads();
            r = tile_weighted(r);
            if (spec) hess_tile(acc);
            __syncthreads();
        }
        const double tot = cross_wave(r);
        double v = block_sum(lin);
        v -= block_sum(ex);
        v -= ncg::wave_sum(y * y) / (2 * variance);
        f = -v;
        double gg = -tot;
        gg += sg;
        gg -= y / variance;
        g = lane < dim ? -gg : 0.0;
        if (spec) hess_finish(acc, y);
    }
    // Dense Hessian at x into Al, unless the last objective+gradient pass already left it there
    __device__ inline void prepare_hess(double x) {
        if (hvalid && __all(x == hpt || lane >= dim)) return;
        --budget;
        broadcast(x);
        d4 acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = (d4){0, 0, 0, 0};
        for (int64_t t0 = n_begin; t0 < n_end; t0 += SIGT) {
            stage(t0);
            const int64_t n = t0 + tid;
            wt[tid] = (n < n_end) ? exp((c + p->alpha[n]) + row_dot(tid)) : 0.0;
            __syncthreads();
            hess_tile(acc);
            __syncthreads();
        }
        hess_finish(acc, x);
    }
    // (Hessian at the fixed point) . v from the LDS copy: no pass over the samples
    __device__ inline double hessp(double v) {
        broadcast(v);
        double r = 0.0;
        if (lane < dim) {
            const double* row = Al + lane * CORR_LD;
            for (int j = 0; j < dim; ++j) r = __builtin_fma(row[j], ybuf[j], r);
        }
        return lane < dim ? r + v / variance : 0.0;
    }
    __device__ inline bool exhausted() const { return budget <= 0; }
};

// LDS pool: the tile [SIGT][16 DT + 1] followed (DT <= 3) by a Hessian region [64][CORR_LD] of its own; with
// DT = 4 the tile alone fills the pool and the Hessian aliases it (no speculation)
constexpr int SIG_POOL = SIGT * 49 + 64 * CORR_LD;
static_assert(SIG_POOL >= SIGT * CORR_LD, "the widest tile must fit the pool");

__global__ void __launch_bounds__(SIGT) corr_signature_embeddings_kernel(SignatureEmbeddingParams p) {
    if (p.only && p.only[blockIdx.x] != SIG_ONLY_VALUE) return;  // uniform over the workgroup
    __shared__ double pool[SIG_POOL];
    __shared__ double wt[SIGT], sred[SIGT], ybuf[64], red[4 * 64];
    SignatureEmbeddingEval ev;
    ev.p = &p;
    ev.dim = p.dim;
    ev.DT = (p.dim + 15) / 16;
    ev.ldu = 16 * ev.DT + 1;
    ev.spec = ev.DT <= 3;
    ev.Ut = pool;
    ev.Al = ev.spec ? pool + SIGT * 49 : pool;
    ev.wt = wt;
    ev.ybuf = ybuf;
    ev.red = red;
    ev.sred = sred;
    ev.k = blockIdx.x;
    ev.c = p.beta[ev.k];
    ev.variance = p.variance;
    ev.tid = threadIdx.x;
    ev.lane = threadIdx.x & 63;
    ev.wave = threadIdx.x >> 6;
    ev.budget = SIG_BUDGET;
    ev.n_begin = 0;
    ev.n_end = p.N;
    ev.hpt = 0.0;
    ev.hvalid = false;
    ev.sg = 0.0;
    ev.sg = ev.weighted_sum<0>(0.0);
    double x = ev.lane < p.dim ? p.L[ev.k * p.dim + ev.lane] : 0.0;
    const int st = ncg::minimize(ev, x, p.dim, p.maxiter);
    if (x > 0.0 && x < kEps) x = kEps;
    if (x < 0.0 && x > -kEps) x = -kEps;
    if (ev.wave == 0 && ev.lane < p.dim) p.L[ev.k * p.dim + ev.lane] = x;
    if (p.status && threadIdx.x == 0) p.status[ev.k] = st;
}

}  // namespace salnmf
