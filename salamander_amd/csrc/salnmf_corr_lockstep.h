// Signature-embedding solves of correlated NMF in LOCKSTEP (mmcorrnmf.py:347-396, corrnmf_det.py:88-113,
// _utils_corrnmf.py:354-410): all K Newton-CG solves advance together, one objective / gradient / Hessian evaluation
// per round.
//
// The single-kernel form (corr_signature_embeddings_kernel: one workgroup per signature, every evaluation a pass of
// that workgroup over ALL samples) leaves 5/6 of the chip idle at K = 40 and cannot be sharded: each solve takes
// data-dependent decisions after every evaluation.  Here an evaluation round is its own launch:
//
//   ls_eval_kernel     grid (S chunks, K signatures): every workgroup passes over ITS chunk of the samples for the point
//                      its signature asked for and leaves partial sums: [lin | ex | sum_n w_n U_n (dim) | Hessian (dim^2)]
//                      -- the same tile code as the single-kernel form (SignatureEmbeddingEval), Hessian on fp64 MFMA
//   ls_reduce_kernel   fixed-order sum of the S partials per signature (then an all-reduce over sample shards: exactly
//                      1 + dim + dim^2 sums per evaluation, as the objective, gradient and Hessian are sums over samples)
//   ls_advance_kernel  one wavefront per signature: appends the evaluation to that signature's log and RE-RUNS
//                      ncg::minimize from the start point with an evaluator that answers from the log (ReplayEval).
//                      The solver is deterministic, so the replay takes the same decisions as before and runs until it
//                      asks for a point that is not in the log: that is the next request.  No solver state has to be
//                      saved and ncg::minimize is used unchanged -- the same code that solves the sample embeddings.
//                      A replay costs the CG iterations again (Hessian-vector products from the logged Hessians, no
//                      pass over samples); it resumes at the top of the last Newton iteration the previous rounds
//                      reached (ncg::Checkpoint), so only that iteration's CG solve and trial points are replayed.
//
// The host loops rounds until no signature asks for another evaluation (one 4-byte read-back per round).  Results:
// objective, gradient and Hessian are the same sums in a different order (chunks), so iterates agree with the
// single-kernel form to rounding; every rank of a sharded fit sees the same reduced sums and takes the same decisions.
#pragma once
#include "salnmf_corr_kernels.h"

namespace salnmf {

constexpr int LS_EVAL_MAX = 192;                    // evaluations per solve kept in the log; beyond: single-kernel fallback
constexpr int LS_REC = 2 + 64 + 64 * 64;            // doubles per evaluation record: [lin, ex | r (64) | Hessian (dim x dim, compact)]
constexpr int LS_NEED = 0, LS_DONE = 1, LS_FALLBACK = SIG_ONLY_VALUE;
constexpr int LS_WIN = 8;                          // entries of a solve's record that ls_advance_kernel keeps in LDS
constexpr int LS_CP = 2 * 64 + 8;                   // doubles per checkpoint: xk, g_next (64 each) | 3 scalars | k, flags, cursor, valid

struct LockstepParams {
    SignatureEmbeddingParams sig;  // aux, alpha, beta, U, L (in / out), status, variance, N, K, KP, dim, maxiter
    int S;                         // chunks per signature
    int64_t chunk;                 // samples per chunk (a multiple of SIGT)
    double* x0;                    // [K][64] start points
    double* req;                   // [K][64] requested evaluation points
    double* sg;                    // [K][64] sum_n aux[n][k] U[n][:]  (constant of a solve)
    int* state;                    // [K] LS_NEED / LS_DONE / LS_FALLBACK
    int* n_evals;                  // [K]
    int* active;                   // [1] signatures that still need an evaluation after this round
    double* part;                  // [K][S][LS_REC] partial sums of one evaluation round
    double* red;                   // [K][red_ld]    reduced (and all-reduced) sums
    int red_ld;                    // doubles per record of `red`: 66 + dim^2, the live part of a record -- what a sharded solve all-reduces
    double* log_y;                 // [K][LS_EVAL_MAX][64]
    double* log_f;                 // [K][LS_EVAL_MAX]
    double* log_g;                 // [K][LS_EVAL_MAX][64]
    double* log_H;                 // [K][LS_EVAL_MAX][dim * dim]
    double* cp;                    // [K][LS_CP]: the solve's state at the top of the last Newton iteration it reached (ncg::Checkpoint)
    int dyn;                       // the workgroups of a round are shared out among the groups that still have a live signature
                                   // (LsLive): more, shorter chunks per group as the solves finish (ls_eval_packed_kernel)
    int lin_from_sg;               // the evaluation kernel leaves the linear term sum_n aux[n][k] <U_n, y> to ls_advance_kernel, which forms
                                   // it as <y, sg_k> (sg = aux^T U is the solve's constant anyway): no aux traffic in the rounds
    long long* prof;               // development builds (SALNMF_DEV_PROFILE): [8] shader-clock ticks per section of ls_eval_packed_kernel, or null
};

__device__ inline void ls_setup_eval(SignatureEmbeddingEval& ev, const LockstepParams& q, double* pool, double* wt, double* sred, double* ybuf,
                                     double* red, int k, int s) {
    ev.p = &q.sig;
    ev.dim = q.sig.dim;
    ev.DT = (q.sig.dim + 15) / 16;
    ev.ldu = 16 * ev.DT + 1;
    ev.spec = ev.DT <= 3;
    ev.Ut = pool;
    ev.Al = ev.spec ? pool + SIGT * 49 : pool;
    ev.wt = wt;
    ev.ybuf = ybuf;
    ev.red = red;
    ev.sred = sred;
    ev.k = k;
    ev.c = q.sig.beta[k];
    ev.variance = q.sig.variance;
    ev.tid = threadIdx.x;
    ev.lane = threadIdx.x & 63;
    ev.wave = threadIdx.x >> 6;
    ev.budget = 1 << 30;
    ev.n_begin = (int64_t)s * q.chunk;
    ev.n_end = ev.n_begin + q.chunk < q.sig.N ? ev.n_begin + q.chunk : q.sig.N;
    ev.hpt = 0.0;
    ev.hvalid = false;
    ev.sg = 0.0;
}

// start of a solve: start points, first requests, and the per-chunk partials of sg = sum_n aux[n][k] U[n][:]
__global__ void __launch_bounds__(SIGT) ls_begin_kernel(LockstepParams q) {
    __shared__ double pool[SIG_POOL];
    __shared__ double wt[SIGT], sred[SIGT], ybuf[64], red[4 * 64];
    const int k = blockIdx.y, s = blockIdx.x;
    SignatureEmbeddingEval ev;
    ls_setup_eval(ev, q, pool, wt, sred, ybuf, red, k, s);
    if (s == 0 && threadIdx.x < 64) {
        const double x = ev.lane < q.sig.dim ? q.sig.L[k * q.sig.dim + ev.lane] : 0.0;
        q.x0[k * 64 + ev.lane] = x;
        q.req[k * 64 + ev.lane] = x;
        if (threadIdx.x == 0) {
            q.state[k] = LS_NEED;
            q.n_evals[k] = 0;
        }
    }
    const double t = ev.weighted_sum<0>(0.0);
    if (ev.wave == 0) q.part[((int64_t)k * q.S + s) * LS_REC + 2 + ev.lane] = t;
}

// one evaluation round: objective parts, gradient part and Hessian over this workgroup's chunk, at the requested point
__global__ void __launch_bounds__(SIGT) ls_eval_kernel(LockstepParams q) {
    const int k = blockIdx.y, s = blockIdx.x;
    if (q.state[k] != LS_NEED) return;  // uniform over the workgroup
    __shared__ double pool[SIG_POOL];
    __shared__ double wt[SIGT], sred[SIGT], ybuf[64], red[4 * 64];
    SignatureEmbeddingEval ev;
    ls_setup_eval(ev, q, pool, wt, sred, ybuf, red, k, s);
    const int dim = q.sig.dim;
    const double y = ev.lane < dim ? q.req[k * 64 + ev.lane] : 0.0;
    ev.broadcast(y);
    double lin = 0.0, ex = 0.0, r = 0.0;
    d4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int64_t t0 = ev.n_begin; t0 < ev.n_end; t0 += SIGT) {
        ev.stage(t0);
        const int64_t n = t0 + ev.tid;
        double w = 0.0;
        if (n < ev.n_end) {
            const double sdot = ev.row_dot(ev.tid);
            lin = __builtin_fma(sdot, q.sig.aux[n * q.sig.KP + k], lin);
            w = exp((ev.c + q.sig.alpha[n]) + sdot);
            ex += w;
        }
        wt[ev.tid] = w;
        __syncthreads();
        r = ev.tile_weighted(r);
        ev.hess_tile(acc);
        __syncthreads();
    }
    const double tot = ev.cross_wave(r);
    const double vlin = ev.block_sum(lin), vex = ev.block_sum(ex);
    ev.hess_finish(acc, y);  // the Hessian sum of this chunk in ev.Al [16 DT][CORR_LD]
    double* out = q.part + ((int64_t)k * q.S + s) * LS_REC;
    if (ev.tid == 0) {
        out[0] = vlin;
        out[1] = vex;
    }
    if (ev.wave == 0) out[2 + ev.lane] = tot;
    for (int i = ev.tid; i < dim * dim; i += SIGT) {
        const int m = i / dim, j = i - m * dim;
        out[66 + i] = ev.Al[m * CORR_LD + j];
    }
}

// ---- the same two passes for LS_GROUP signatures per workgroup (dim <= 48).  A pass over the samples is bound by
// reading U: with one signature per workgroup every signature re-reads all of U (2.6 GB per round at c5: 0.87 of the
// 1.73 ms of a round, measured by switching the arithmetic off), and aux is read one 8-byte column at a time.  Here a
// staged tile of U serves LS_GROUP signatures (U traffic / LS_GROUP, aux columns of a group share their cache lines);
// per signature the arithmetic, its order and the reduction over the chunks are those of the kernels above.
constexpr int LS_GROUP = 5;

// Hessian tiles of DT <= 3 in a compact array: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
__device__ inline void ls_hess_tile3(const SignatureEmbeddingEval& ev, d4 (&acc)[6]) {
    const int c16 = ev.lane & 15, q = ev.lane >> 4, DT = ev.DT;
    const double* base = ev.Ut + (64 * ev.wave + q) * ev.ldu + c16;
    double wv[2], a[2][3];
    wv[0] = ev.wt[64 * ev.wave + q];
#pragma unroll
    for (int t = 0; t < 3; ++t) a[0][t] = t < DT ? base[16 * t] : 0.0;
#pragma unroll
    for (int sgrp = 0; sgrp < 16; ++sgrp) {
        const int cur = sgrp & 1, nxt = cur ^ 1;
        if (sgrp + 1 < 16) {
            wv[nxt] = ev.wt[64 * ev.wave + 4 * (sgrp + 1) + q];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[nxt][t] = t < DT ? base[4 * (sgrp + 1) * ev.ldu + 16 * t] : 0.0;
        }
        double b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) b[t] = wv[cur] * a[cur][t];
        acc[0] = mfma(a[cur][0], b[0], acc[0]);
        if (DT > 1) {  // uniform over the workgroup
            acc[1] = mfma(a[cur][0], b[1], acc[1]);
            acc[3] = mfma(a[cur][1], b[1], acc[3]);
        }
        if (DT > 2) {
            acc[2] = mfma(a[cur][0], b[2], acc[2]);
            acc[4] = mfma(a[cur][1], b[2], acc[4]);
            acc[5] = mfma(a[cur][2], b[2], acc[5]);
        }
    }
}

// The start of the grouped solves as ONE matrix product on the fp64 MFMA units:  sg = aux^T U  (K x N times N x dim), every
// signature at once (a pass per group of five signatures with each component summed through a chain of dependent FMAs
// took 0.37 ms at c5, twice per update: 0.05 ms now); U and aux are read once, straight from global memory in operand
// layout (A[k][n] = aux[n][k]: lane (k = lane & 15, n = lane >> 4); B[n][m] = U[n][m]), four workgroups per CU.  part2 [gridDim.x][K][64]: the workgroups' partial sums, added in order by ls_reduce_sg_kernel.
__global__ void __launch_bounds__(256) ls_begin_mfma_kernel(LockstepParams q, double* __restrict__ part2, int64_t rows_per_wg) {
    __shared__ double red[64 * 48];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, qq = lane >> 4;
    const int K = q.sig.K, dim = q.sig.dim, KP = q.sig.KP;
    if (blockIdx.x == 0) {
        for (int k = wave; k < K; k += 4) {
            const double x = lane < dim ? q.sig.L[k * dim + lane] : 0.0;
            q.x0[k * 64 + lane] = x;
            q.req[k * 64 + lane] = x;
            if (lane == 0) {
                q.state[k] = LS_NEED;
                q.n_evals[k] = 0;
            }
        }
    }
    const int KT = (K + 15) / 16, DT = (dim + 15) / 16;  // <= 4, <= 3
    d4 acc[4][3];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) acc[kt][dt] = (d4){0, 0, 0, 0};
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t n1 = n0 + rows_per_wg < q.sig.N ? n0 + rows_per_wg : q.sig.N;
    for (int64_t t = n0 + 64 * wave; t < n1; t += 256) {
#pragma unroll 4
        for (int sgrp = 0; sgrp < 16; ++sgrp) {
            const int64_t n = t + 4 * sgrp + qq;
            const bool valid = n < n1;
            double a[4], b[3];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) a[kt] = (valid && 16 * kt + c16 < K) ? q.sig.aux[n * KP + 16 * kt + c16] : 0.0;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) b[dt] = (valid && 16 * dt + c16 < dim) ? q.sig.U[n * dim + 16 * dt + c16] : 0.0;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < KT) {
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt)
                        if (dt < DT) acc[kt][dt] = mfma(a[kt], b[dt], acc[kt][dt]);
                }
        }
    }
    // the four waves' sums in fixed order (((wave 0 + 1) + 2) + 3) through one staging buffer; D[row = q + 4 r][col = c16]
    // of tile (kt, dt)
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * kt + qq + 4 * r) * 48 + 16 * dt + c16] = acc[kt][dt][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[kt][dt][r] += red[(16 * kt + qq + 4 * r) * 48 + 16 * dt + c16];
        }
        __syncthreads();
    }
    if (wave == 0) {
        double* out = part2 + (int64_t)blockIdx.x * K * 64;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kt + qq + 4 * r, m = 16 * dt + c16;
                    if (k < K && m < dim) out[k * 64 + m] = acc[kt][dt][r];
                }
    }
}

// red[k][2 + m] = sum over the workgroups' partial sums of ls_begin_mfma_kernel: sixteen interleaved sub-sums (wave w:
// partials w, w + 16, ... in order), then the sixteen in order
__global__ void __launch_bounds__(1024) ls_reduce_sg_kernel(const double* __restrict__ part2, double* __restrict__ red, int red_ld, int nparts, int K, int dim) {
    __shared__ double sub[16][64];
    const int k = blockIdx.x, m = threadIdx.x & 63, w = threadIdx.x >> 6;
    double t = 0.0;
    if (m < dim) {
        // (eight loads in flight per round: one per iteration is a memory round trip per iteration)
        int s = w;
        for (; s + 16 * 7 < nparts; s += 16 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part2[((int64_t)(s + 16 * u) * K + k) * 64 + m];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; s < nparts; s += 16) t += part2[((int64_t)s * K + k) * 64 + m];
    }
    sub[w][m] = t;
    __syncthreads();
    if (w == 0) {
        double tot = sub[0][m];
        for (int i = 1; i < 16; ++i) tot += sub[i][m];
        red[(int64_t)k * red_ld + 2 + m] = tot;
    }
}

// component m of a vector held one component per lane (wave-uniform m): v_readlane instead of a broadcast read from LDS
__device__ __forceinline__ double ls_lane_value(double v, int m) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), m), __builtin_amdgcn_readlane(__double2loint(v), m));
}

// dim a multiple of 16 (no free component column in the 16-padded tile: see ls_eval_packed_kernel for the other case): the
// logits of the LS_GROUP signatures in ONE sweep over the sample's row (one LDS read per component, the requested points
// come from registers by v_readlane; per signature the FMA chain of row_dot, same bits); gradient part and sum of the
// weights on the VALU (two barriers per signature), the Hessian's Gram product on the MFMA units.
__global__ void __launch_bounds__(SIGT) ls_eval_multi_kernel(LockstepParams q) {
    const int kbase = blockIdx.y * LS_GROUP, s = blockIdx.x;
    bool live[LS_GROUP], any = false;
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
        live[g] = kbase + g < q.sig.K && q.state[kbase + g] == LS_NEED;  // uniform over the workgroup
        any |= live[g];
    }
    if (!any) return;
    __shared__ double pool[SIG_POOL];
    __shared__ double wt[SIGT], sred[SIGT], ybufs[LS_GROUP][64], red[4 * 64];
    SignatureEmbeddingEval ev;
    ls_setup_eval(ev, q, pool, wt, sred, ybufs[0], red, kbase, s);
    const int dim = q.sig.dim;
    double cg[LS_GROUP], yg[LS_GROUP];
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
        const int k = live[g] ? kbase + g : kbase;
        cg[g] = q.sig.beta[k];
        yg[g] = (live[g] && ev.lane < dim) ? q.req[k * 64 + ev.lane] : 0.0;
    }
    double lin[LS_GROUP], ex[LS_GROUP], r[LS_GROUP];
    d4 acc[LS_GROUP][6];
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
        lin[g] = ex[g] = r[g] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[g][i] = (d4){0, 0, 0, 0};
    }
    for (int64_t t0 = ev.n_begin; t0 < ev.n_end; t0 += SIGT) {
        ev.stage(t0);  // (ends with a workgroup barrier; rows 64 wave .. 64 wave + 63 are this wave's samples)
        const int64_t n = t0 + ev.tid;
        const bool in = n < ev.n_end;
        const double al = in ? q.sig.alpha[n] : 0.0;
        double* myrow = ev.Ut + ev.tid * ev.ldu;
        // <U_n, y_g> for the group's signatures in one sweep over the row (per signature: row_dot's FMA chain)
        double sd[LS_GROUP];
#pragma unroll
        for (int g = 0; g < LS_GROUP; ++g) sd[g] = 0.0;
        int m = 0;
        for (; m + 4 <= dim; m += 4) {
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = myrow[m + u];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < LS_GROUP; ++g) sd[g] = __builtin_fma(x[u], ls_lane_value(yg[g], m + u), sd[g]);
        }
        for (; m < dim; ++m) {
            const double x = myrow[m];
#pragma unroll
            for (int g = 0; g < LS_GROUP; ++g) sd[g] = __builtin_fma(x, ls_lane_value(yg[g], m), sd[g]);
        }
        double auxv[LS_GROUP];
#pragma unroll
        for (int g = 0; g < LS_GROUP; ++g) auxv[g] = (in && live[g]) ? q.sig.aux[n * q.sig.KP + kbase + g] : 0.0;
#pragma unroll
        for (int g = 0; g < LS_GROUP; ++g) {
            if (!live[g]) continue;  // uniform
            double w = 0.0;
            if (in) {
                lin[g] = __builtin_fma(sd[g], auxv[g], lin[g]);
                w = exp((cg[g] + al) + sd[g]);
                ex[g] += w;
            }
            wt[ev.tid] = w;
            __syncthreads();
            r[g] = ev.tile_weighted(r[g]);
            ls_hess_tile3(ev, acc[g]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
        if (!live[g]) continue;
        const int k = kbase + g;
        const double tot = ev.cross_wave(r[g]);
        const double vex = ev.block_sum(ex[g]);
        const double vlin = ev.block_sum(lin[g]);
        d4 full[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) full[i] = (d4){0, 0, 0, 0};
        full[0] = acc[g][0], full[1] = acc[g][1], full[2] = acc[g][2], full[4] = acc[g][3], full[5] = acc[g][4], full[7] = acc[g][5];
        ev.hess_finish(full, yg[g]);  // the Hessian sum of this chunk in ev.Al [16 DT][CORR_LD]
        double* out = q.part + ((int64_t)k * q.S + s) * LS_REC;
        if (ev.tid == 0) {
            out[0] = vlin;
            out[1] = vex;
        }
        if (ev.wave == 0) out[2 + ev.lane] = tot;
        for (int i = ev.tid; i < dim * dim; i += SIGT) {
            const int m = i / dim, j = i - m * dim;
            out[66 + i] = ev.Al[m * CORR_LD + j];
        }
        __syncthreads();  // Al / the staging copies are rewritten for the next signature
    }
}

// The rounds' workgroups shared out among the groups of LS_GROUP signatures that still have a live one: with B workgroups
// and ngl live groups every live group gets Sd = B / ngl chunks of cd samples (a multiple of SIGT).  Early rounds: every
// group live, Sd = S, the static partition.  Late rounds: the few remaining groups use the whole chip instead of their
// S workgroups (a round with two of eight groups live took 281 us of a full round's 485 at c5).  Evaluation and reduction
// derive the same map from the signatures' states; the partial records are indexed (ordinal of the group, slot, chunk).
struct LsLive {
    int ngl, Sd, ord, nchunks;  // live groups; chunks per live group; ordinal of THIS group (-1: none live in it); chunks that hold samples
    int64_t cd;                 // samples per chunk
};
__device__ inline LsLive ls_live_map(const int* __restrict__ state, int K, int B, int64_t N, int group) {
    LsLive m;
    m.ngl = 0;
    m.ord = -1;
    const int groups = (K + LS_GROUP - 1) / LS_GROUP;
    for (int g = 0; g < groups; ++g) {
        bool any = false;
        for (int j = 0; j < LS_GROUP; ++j) any |= g * LS_GROUP + j < K && state[g * LS_GROUP + j] == LS_NEED;
        if (g == group && any) m.ord = m.ngl;
        m.ngl += any;
    }
    m.Sd = m.ngl > 0 ? B / m.ngl : 0;
    m.cd = m.Sd > 0 ? ((N + m.Sd - 1) / m.Sd + SIGT - 1) / SIGT * SIGT : 0;
    m.nchunks = m.cd > 0 ? (int)((N + m.cd - 1) / m.cd) : 0;
    return m;
}
// group index of the live group with ordinal o
__device__ inline int ls_live_group(const int* __restrict__ state, int K, int o) {
    const int groups = (K + LS_GROUP - 1) / LS_GROUP;
    int seen = 0;
    for (int g = 0; g < groups; ++g) {
        bool any = false;
        for (int j = 0; j < LS_GROUP; ++j) any |= g * LS_GROUP + j < K && state[g * LS_GROUP + j] == LS_NEED;
        if (any && seen++ == o) return g;
    }
    return 0;
}

// ---- dim <= 48, NOT a multiple of 16: the 16-padded tile has a free component column.
// (i) The staged tile carries a column of ones at component `dim`, so the weighted Gram product that forms the Hessian,
// sum_n w_n [U_n 1]^T [U_n 1],  also yields the gradient part  sum_n w_n U_n  (column dim) and  sum_n w_n  (entry [dim][dim])
// on the MFMAs that run anyway: every wave works on ITS 64 samples of a tile from the logits to the Hessian with no
// workgroup barrier in between, no per-signature pass over the tile for the gradient, no cross-wave sum of it (the two sums
// come out in the MFMA's order: rounding level, the same on every rank of a sharded solve).
// (ii) The last block column is PACKED over the group's signatures.  Of the padded triangle's tiles, those of the last
// block column carry only w = dim + 1 - 16 (DT - 1) live columns (c5: dim 40 -> 9 of 16): 1.87x the algorithmic MACs.  The
// A operand of a tile (U^T, unweighted) is the same for every signature; only B = diag(weights_g) U differs.  So the
// LS_GROUP x w live columns of that block column are laid side by side -- packed column P = g w + c of tile P / 16 is
// weights_g[n] U[n][16 (DT - 1) + c]  -- and multiplied by the DT row tiles ONCE for the whole group: DT ceil(LS_GROUP w / 16)
// MFMAs per four samples instead of LS_GROUP DT (c5: 9 instead of 15; 24 with the 15 dense ones instead of 30).  An entry
// of a product does not depend on its column's neighbours: the bits are those of one product per signature.
// (iii) PF: the next tile of U is on its way to registers while this tile's products run.  FIX: the common tile counts as
// compile-time constants.  582 -> 458 us per round at c5 for (ii) + (iii) (profiles/r04/lockstep_packed.md).
//   wt5 [SIGT][6]: the tile's weights, sample-major (slot 5 = 0: the operand of packed columns beyond LS_GROUP w)
template <int PKMAX, int PF, bool AG, bool FIX, bool DMA = false>
__global__ void __launch_bounds__(SIGT) ls_eval_packed_kernel(LockstepParams q) {
    int kbase = blockIdx.y * LS_GROUP, s = blockIdx.x;
    LsLive lm{};
    if (q.dyn) {  // (uniform) this workgroup's group and chunk from the map of live groups
        const int B = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        lm = ls_live_map(q.state, q.sig.K, B, q.sig.N, -1);
        if (lm.ngl == 0 || b >= lm.Sd * lm.ngl) return;
        lm.ord = b / lm.Sd;
        s = b - lm.ord * lm.Sd;
        if (s >= lm.nchunks) return;
        kbase = ls_live_group(q.state, q.sig.K, lm.ord) * LS_GROUP;
    }
    bool live[LS_GROUP], any = false;
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
        live[g] = kbase + g < q.sig.K && q.state[kbase + g] == LS_NEED;  // uniform over the workgroup
        any |= live[g];
    }
    if (!any) return;
    __shared__ __attribute__((aligned(16))) double pool[SIG_POOL];  // (16 bytes: the LDS-DMA variant writes it in 16-byte pieces)
    __shared__ double wt5[SIGT * 6], sred[SIGT], ybuf0[64], red[4 * 64];
    SignatureEmbeddingEval ev;
    ls_setup_eval(ev, q, pool, wt5, sred, ybuf0, red, kbase, s);
    if (q.dyn) {
        ev.n_begin = (int64_t)s * lm.cd;
        ev.n_end = ev.n_begin + lm.cd < q.sig.N ? ev.n_begin + lm.cd : q.sig.N;
    }
    // (DMA: rows of 50 doubles = 25 sixteen-byte pieces, so that a row starts on a piece boundary)
    const int dim = q.sig.dim, DT = ev.DT, ldu = DMA ? 50 : ev.ldu;
    const int c16 = ev.lane & 15, qq = ev.lane >> 4;
    const int tail0 = 16 * (DT - 1), w = dim + 1 - tail0;  // the last block column: first component, live columns
    const int NPK = (LS_GROUP * w + 15) / 16;               // packed tiles (<= PKMAX: the launcher's choice)
    // this lane's column of packed tile p: signature slot pg (5 = none), component column pc
    int pg[PKMAX], pc[PKMAX];
#pragma unroll
    for (int p = 0; p < PKMAX; ++p) {
        const int P = 16 * p + c16;
        pg[p] = P / w < LS_GROUP ? P / w : LS_GROUP;
        pc[p] = tail0 + (P / w < LS_GROUP ? P % w : 0);
    }
    // The logits <U_n, y_g> of a tile as MFMA products too:  D[signature slot][sample] = Y . U^T  with A = Y (row i = slot
    // i < LS_GROUP, else 0; k = component) held in registers for the whole launch and B = U^T read from the staged tile.
    // Lane (q, c16) of the result holds, per 16-sample tile, slot q (register 0) and -- lanes q = 0 only -- slot 4
    // (register 1) of sample c16.  (As chains of FMAs with the points' components by v_readlane this phase took 11.6 k of
    // a tile's 48 k cycles: a readlane's scalar result stalls the FMA that consumes it.)
    static_assert(LS_GROUP == 5, "slot layout of the logits product");
    constexpr int KSMAX = FIX ? 10 : 12;  // dim <= 48 (FIX: <= 40)
    const int KSn = (dim + 3) / 4;
    int livemask = 0;
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) livemask |= live[g] ? 1 << g : 0;
    // (Y in LDS, [component][slot | 0]: as registers its ten k-steps were what pushed the kernel into scratch)
    __shared__ double ysh[48 * 6];
    for (int i = ev.tid; i < 48 * 6; i += SIGT) {
        const int comp = i / 6, slot = i - comp * 6;
        const bool mine = slot < LS_GROUP && ((livemask >> slot) & 1) && comp < dim;
        const int krow = mine ? kbase + slot : kbase;
        const double v = q.req[krow * 64 + comp];  // (unconditional load at a valid address, masked afterwards)
        ysh[i] = mine ? v : 0.0;
    }
    const double* yl = ysh + qq * 6 + (c16 < LS_GROUP ? c16 : LS_GROUP);  // this lane's A operand of k-step ks: yl[24 ks]
    const bool live_lo = (livemask >> qq) & 1;  // this lane's slot q
    const bool live_hi = live[4] && qq == 0;
    const int slot_lo = kbase + qq < q.sig.K ? kbase + qq : kbase, slot_hi = kbase + 4 < q.sig.K ? kbase + 4 : kbase;
    const double cg_lo = q.sig.beta[slot_lo], cg_hi = q.sig.beta[slot_hi];
    for (int i = ev.tid; i < SIGT * 6; i += SIGT) wt5[i] = 0.0;  // (signatures that asked for nothing keep weight 0)
    d4 dense[LS_GROUP][3], packed[3][PKMAX];
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
#pragma unroll
        for (int i = 0; i < 3; ++i) dense[g][i] = (d4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int p = 0; p < PKMAX; ++p) packed[i][p] = (d4){0, 0, 0, 0};
#ifdef SALNMF_DEV_PROFILE
    long long tks[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk0 = __builtin_amdgcn_s_memtime(), tk1;
#define LS_TICK(i) do { tk1 = __builtin_amdgcn_s_memtime(); tks[i] += tk1 - tk0; tk0 = tk1; } while (0)
#else
#define LS_TICK(i) do { } while (0)
#endif
    // PF: the NEXT tile of U travels from global memory to registers (element tid + 256 u of the tile's SIGT x dim block)
    // while this tile's products run, and to LDS behind them -- with one wave per SIMD nothing else hides that latency
    double pf[PF > 0 ? PF : 1];
    const int64_t u_end = q.sig.N * (int64_t)dim;
    // (a tile that lies inside the samples -- every one but possibly the last -- takes the plain forms: no clamp and no
    // select per element; the LDS position of element u + 1 follows from that of element u by additions)
    auto pf_load = [&](int64_t t0) {
        const double* src = q.sig.U + t0 * dim + ev.tid;
        if (t0 + SIGT <= q.sig.N) {  // uniform
#pragma unroll
            for (int u = 0; u < PF; ++u) pf[u] = src[u * SIGT];
        } else {
            const int64_t base = t0 * dim;
#pragma unroll
            for (int u = 0; u < PF; ++u) {  // (unconditional loads at clamped addresses: no branch around any of them)
                const int64_t i = base + ev.tid + (int64_t)u * SIGT;
                pf[u] = q.sig.U[i < u_end ? i : u_end - 1];  // (rows beyond N are zeroed when the value is used: pf_store)
            }
        }
    };
    const int pf_j0 = ev.tid / dim, pf_m0 = ev.tid - pf_j0 * dim, pf_dj = SIGT / dim, pf_dm = SIGT - pf_dj * dim;
    auto pf_store = [&](int64_t t0) {
        int m = pf_m0;
        double* dst = ev.Ut + pf_j0 * ldu + pf_m0;
        const int step = pf_dj * ldu + pf_dm, wrap = ldu - dim;  // to the next element; extra when it starts a new row
        const bool inside = t0 + SIGT <= q.sig.N;               // uniform
        const int64_t base = t0 * dim;
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if ((FIX && u < 32) || u < dim) {  // (FIX: dim > 32 -- no branch around the first 32)
                *dst = (inside || base + ev.tid + (int64_t)u * SIGT < u_end) ? pf[u] : 0.0;
                m += pf_dm;
                const bool over = m >= dim;
                dst += step + (over ? wrap : 0);
                m -= over ? dim : 0;
            }
        // (elements PF .. dim - 1 of this lane, if any: loaded here, eight at a time)
        for (int u0 = PF; u0 < dim; u0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = base + ev.tid + (int64_t)(u0 + u) * SIGT;
                v[u] = q.sig.U[i < u_end ? i : u_end - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u0 + u < dim) {
                    *dst = base + ev.tid + (int64_t)(u0 + u) * SIGT < u_end ? v[u] : 0.0;
                    m += pf_dm;
                    const bool over = m >= dim;
                    dst += step + (over ? wrap : 0);
                    m -= over ? dim : 0;
                }
        }
        LS_TICK(8);
        __syncthreads();
    };
    if (PF > 0) {
        // the component columns dim .. 16 DT - 1 of the tile are written once: the column of ones (component dim), zeros
        // behind it -- pf_store leaves them alone
        const int dpad = 16 * DT - dim;
        for (int i = ev.tid; i < SIGT * dpad; i += SIGT) {
            const int jj = i / dpad, cc = i - jj * dpad;
            ev.Ut[jj * ldu + dim + cc] = cc == 0 ? 1.0 : 0.0;
        }
    }
    // the per-sample scalars of the next tile likewise (issued BEFORE the tile's elements: loads return in order, and these
    // are needed first)
    double al[4];  // sample scalings of samples 16 st + c16 of this wave's 64, st = 0 .. 3: requested at the top of a tile
                   // (ahead of the next tile's elements: loads return in order), used behind the logits
    auto scalars_load = [&](int64_t t0) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int64_t n = t0 + 64 * ev.wave + 16 * st + c16;
            al[st] = q.sig.alpha[n < ev.n_end ? n : ev.n_end - 1];  // (samples beyond the chunk: values not used)
        }
    };
    // The products of one block of this wave's samples: NST tiles of 16 (a wave's 64 rows of a staged tile, or the 32 rows of
    // a DMA half).  rows: the block's first row in LDS; wrows: its weights; n_first: its first sample; alv: sample scalings
    auto compute = [&](auto NSTC, const double* rows, double* wrows, int64_t n_first, const double (&alv)[4]) {
        constexpr int NST = decltype(NSTC)::value, NSG = 4 * NST;
        // logits of the block's samples, 16 at a time (KSn k-steps each), and behind each product the weights
        // exp(beta + alpha + logit) of (sample 16 st + c16, slot q) and, lanes q = 0, (sample, slot 4)
        {
            const double* ubase = rows + c16 * ldu + qq;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                d4 sl = (d4){0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSMAX; ++ks)
                    if (FIX ? (ks < 8 || ks < KSn) : ks < KSn) sl = mfma(yl[24 * ks], ubase[16 * st * ldu + 4 * ks], sl);  // (uniform; FIX: dim > 32)
                const int64_t n = n_first + 16 * st + c16;
                const bool in = n < ev.n_end;
                double* wrow = wrows + (16 * st + c16) * 6;
                if (live_lo) {
                    double wgt = 0.0;
                    if (in) wgt = exp((cg_lo + alv[st]) + sl[0]);
                    wrow[qq] = wgt;
                }
                if (live_hi) {
                    double wgt = 0.0;
                    if (in) wgt = exp((cg_hi + alv[st]) + sl[1]);
                    wrow[4] = wgt;
                }
            }
        }
        LS_TICK(1);
        __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in order: its own rows and weights)
        // four samples per step: operands one step ahead of the MFMAs that use them
        const double* ub = rows + qq * ldu;
        const double* wb = wrows + qq * 6;
        // (FIX: the tile counts of the common shape -- three row tiles, three packed tiles -- as compile-time constants; as
        // runtime values every MFMA sits behind a scalar branch of its own)
        const int dt = FIX ? 3 : DT, npk = FIX ? 3 : NPK;
        LS_TICK(2);
        // the dense leading tiles.  DMA (blocks of 32 samples): step-outer, the group's signatures side by side -- one
        // operand-fetch prologue per block instead of one per signature; else signature by signature (one branch per
        // signature and block, not per step)
        if (DMA && dt > 1) {
            double a[2][2], wv[2][LS_GROUP];
            auto fetch = [&](int buf, int sgrp) {
                const double* ur = ub + 4 * sgrp * ldu;
                a[buf][0] = ur[c16];
                a[buf][1] = dt > 2 ? ur[16 + c16] : 0.0;
#pragma unroll
                for (int g = 0; g < LS_GROUP; ++g) wv[buf][g] = wb[4 * sgrp * 6 + g];
            };
            fetch(0, 0);
#pragma unroll
            for (int sgrp = 0; sgrp < NSG; ++sgrp) {
                const int cur = sgrp & 1;
                if (sgrp + 1 < NSG) fetch(cur ^ 1, sgrp + 1);
#pragma unroll
                for (int g = 0; g < LS_GROUP; ++g) {
                    if (!live[g]) continue;
                    const double b0 = wv[cur][g] * a[cur][0];
                    if (AG) mfma_agpr(dense[g][0], a[cur][0], b0); else dense[g][0] = mfma(a[cur][0], b0, dense[g][0]);
                    if (dt > 2) {
                        const double b1 = wv[cur][g] * a[cur][1];
                        if (AG) mfma_agpr(dense[g][1], a[cur][0], b1); else dense[g][1] = mfma(a[cur][0], b1, dense[g][1]);
                        if (AG) mfma_agpr(dense[g][2], a[cur][1], b1); else dense[g][2] = mfma(a[cur][1], b1, dense[g][2]);
                    }
                }
            }
        } else if (dt > 1) {
#pragma unroll
            for (int g = 0; g < LS_GROUP; ++g) {
                if (!live[g]) continue;
                double a[2][2], wv[2];
                auto fetch = [&](int buf, int sgrp) {
                    const double* ur = ub + 4 * sgrp * ldu;
                    a[buf][0] = ur[c16];
                    a[buf][1] = dt > 2 ? ur[16 + c16] : 0.0;
                    wv[buf] = wb[4 * sgrp * 6 + g];
                };
                fetch(0, 0);
#pragma unroll
                for (int sgrp = 0; sgrp < NSG; ++sgrp) {
                    const int cur = sgrp & 1;
                    if (sgrp + 1 < NSG) fetch(cur ^ 1, sgrp + 1);
                    const double b0 = wv[cur] * a[cur][0];
                    if (AG) mfma_agpr(dense[g][0], a[cur][0], b0); else dense[g][0] = mfma(a[cur][0], b0, dense[g][0]);
                    if (dt > 2) {
                        const double b1 = wv[cur] * a[cur][1];
                        if (AG) mfma_agpr(dense[g][1], a[cur][0], b1); else dense[g][1] = mfma(a[cur][0], b1, dense[g][1]);
                        if (AG) mfma_agpr(dense[g][2], a[cur][1], b1); else dense[g][2] = mfma(a[cur][1], b1, dense[g][2]);
                    }
                }
            }
        }
        LS_TICK(3);
        // the packed last block column, once for the group
        {
            double a[2][3], pu[2][PKMAX], pw[2][PKMAX];
            auto fetch = [&](int buf, int sgrp) {
                const double* ur = ub + 4 * sgrp * ldu;
                const double* wr = wb + 4 * sgrp * 6;
#pragma unroll
                for (int t = 0; t < 3; ++t) a[buf][t] = t < dt ? ur[16 * t + c16] : 0.0;
#pragma unroll
                for (int p = 0; p < PKMAX; ++p)
                    if (p < npk) {
                        pu[buf][p] = ur[pc[p]];
                        pw[buf][p] = wr[pg[p]];
                    }
            };
            fetch(0, 0);
#pragma unroll
            for (int sgrp = 0; sgrp < NSG; ++sgrp) {
                const int cur = sgrp & 1;
                if (sgrp + 1 < NSG) fetch(cur ^ 1, sgrp + 1);
#pragma unroll
                for (int p = 0; p < PKMAX; ++p)
                    if (p < npk) {
                        const double bp = pw[cur][p] * pu[cur][p];
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            if (t < dt) { if (AG) mfma_agpr(packed[t][p], a[cur][t], bp); else packed[t][p] = mfma(a[cur][t], bp, packed[t][p]); }
                    }
            }
        }
        LS_TICK(4);
    };
    if constexpr (DMA) {
        // ---- the tile by LDS-DMA, per wave, double-buffered in halves of 32 samples.  Every wave stages and consumes its OWN
        // rows: global_load_lds_dwordx4 puts lane l's 16 bytes at base + 16 l, so one instruction moves 64 pieces (2.56 rows of
        // 25 pieces; the five pad pieces of a row -- the column of ones, zeros -- are written once and skipped) from global
        // memory to LDS without passing through registers; half k + 1 travels while half k is multiplied, and no workgroup
        // barrier is left in the loop.  Rows of samples beyond the chunk keep older (finite) values: their weights are 0.
        double* myreg = ev.Ut + 64 * ev.wave * ldu;
        for (int i = ev.lane; i < 64 * ldu; i += 64) myreg[i] = 0.0;
        __builtin_amdgcn_wave_barrier();
        myreg[ev.lane * ldu + dim] = 1.0;
        __syncthreads();  // (wt5 zeroed above by all threads)
        const int64_t nunits = (ev.n_end - ev.n_begin + 31) / 32;
        const int hp = dim / 2;  // data pieces per row
        // piece 64 i + lane of a half's image: its offset in the unit's 32 x dim block of U (doubles), -1 for the pad pieces
        // and beyond the 32 rows (thirteen per lane, computed once)
        int poff[13];
        {
            int row = ev.lane / 25, slot = ev.lane - 25 * row;
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                poff[i] = (row < 32 && slot < hp) ? row * dim + 2 * slot : -1;
                row += 2, slot += 14;
                if (slot >= 25) slot -= 25, ++row;
            }
        }
        auto issue = [&](int64_t u, int half) {
            const int64_t base = ev.n_begin + 32 * u;
            const double* src = q.sig.U + base * dim;
            double* dst = myreg + half * 32 * ldu;
            const int64_t lim = (q.sig.N - base) * dim;  // (rows of the unit beyond N are not requested)
#pragma unroll
            for (int i = 0; i < 13; ++i)
                if (poff[i] >= 0 && poff[i] < lim)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + poff[i]), (__attribute__((address_space(3))) void*)(dst + 128 * i), 16, 0,
                                                     0);
        };
        double al_next[4] = {0.0, 0.0, 0.0, 0.0};
        auto scal = [&](int64_t u) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int64_t n = ev.n_begin + 32 * u + 16 * st + c16;
                al_next[st] = q.sig.alpha[n < ev.n_end ? n : ev.n_end - 1];
            }
        };
        int64_t u = ev.wave;
        int half = 0;
        if (u < nunits) {
            issue(u, 0);
            scal(u);
        }
        for (; u < nunits; u += 4, half ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this half and its scalars have arrived)
            LS_TICK(8);
            double alv[4] = {al_next[0], al_next[1], 0.0, 0.0};
            if (u + 4 < nunits) {
                issue(u + 4, half ^ 1);
                scal(u + 4);
            }
            LS_TICK(0);
            compute(std::integral_constant<int, 2>{}, myreg + half * 32 * ldu, wt5 + 64 * ev.wave * 6, ev.n_begin + 32 * u, alv);
        }
        __syncthreads();  // every wave is done with its rows: the finish below uses the tile buffer as staging
    } else {
    if (PF > 0) pf_load(ev.n_begin);
    for (int64_t t0 = ev.n_begin; t0 < ev.n_end; t0 += SIGT) {
        if (PF > 0) pf_store(t0);
        else ev.stage(t0);  // (ends with a workgroup barrier; rows 64 wave .. 64 wave + 63 are this wave's samples)
        LS_TICK(0);
        scalars_load(t0);
        if (PF > 0 && t0 + SIGT < ev.n_end) pf_load(t0 + SIGT);
        if (PF == 0) {
            ev.Ut[ev.tid * ldu + dim] = 1.0;  // (the column of ones of this thread's row: ev.stage zeroed it)
            __builtin_amdgcn_wave_barrier();
        }
        LS_TICK(9);
        compute(std::integral_constant<int, 4>{}, ev.Ut + 64 * ev.wave * ldu, wt5 + 64 * ev.wave * 6, t0 + 64 * ev.wave, al);
        __syncthreads();  // every wave is done with the tile before the next one is staged over it
        LS_TICK(5);
    }
    }
#pragma unroll
    for (int g = 0; g < LS_GROUP; ++g) {
        if (!live[g]) continue;
        const int k = kbase + g;
        // this signature's (augmented) Hessian sum: the waves' parts to their staging copies, then the fixed-order sum
        double* mine = ev.Ut + ev.wave * (16 * DT) * CORR_LD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = qq + 4 * r;
            if (DT > 1) mine[row * CORR_LD + c16] = dense[g][0][r];
            if (DT > 2) {
                mine[row * CORR_LD + 16 + c16] = dense[g][1][r];
                mine[(16 + c16) * CORR_LD + row] = dense[g][1][r];
                mine[(16 + row) * CORR_LD + 16 + c16] = dense[g][2][r];
            }
        }
#pragma unroll
        for (int p = 0; p < PKMAX; ++p)
            if (p < NPK && pg[p] == g) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t < DT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * t + qq + 4 * r;
                            mine[row * CORR_LD + pc[p]] = packed[t][p][r];
                            if (t < DT - 1) mine[pc[p] * CORR_LD + row] = packed[t][p][r];  // lower triangle = mirror
                        }
                    }
            }
        __syncthreads();
        const int nA = 16 * DT * CORR_LD;
        for (int i = ev.tid; i < nA; i += SIGT) ev.Al[i] = ((ev.Ut[i] + ev.Ut[nA + i]) + ev.Ut[2 * nA + i]) + ev.Ut[3 * nA + i];
        __syncthreads();
        double* out = q.part + (q.dyn ? ((int64_t)(lm.ord * LS_GROUP + g) * lm.Sd + s) : ((int64_t)k * q.S + s)) * LS_REC;
        if (ev.tid == 0) {
            out[0] = 0.0;  // (the linear term: ls_advance_kernel, LockstepParams::lin_from_sg)
            out[1] = ev.Al[dim * CORR_LD + dim];
        }
        if (ev.wave == 0) out[2 + ev.lane] = ev.lane < dim ? ev.Al[ev.lane * CORR_LD + dim] : 0.0;
        for (int i = ev.tid; i < dim * dim; i += SIGT) {
            const int mm = i / dim, j = i - mm * dim;
            out[66 + i] = ev.Al[mm * CORR_LD + j];
        }
        __syncthreads();  // Al / the staging copies are rewritten for the next signature
    }
    LS_TICK(6);
#ifdef SALNMF_DEV_PROFILE
    if (q.prof != nullptr && ev.lane == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(q.prof) + i, (unsigned long long)tks[i]);
        for (int i = 8; i < 10; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(q.prof) + i, (unsigned long long)tks[i]);
        atomicAdd(reinterpret_cast<unsigned long long*>(q.prof) + 7, 1ull);
    }
#endif
#undef LS_TICK
}

// red[k][e] = sum over the S chunk partials, fixed order; only the first `len` entries of a record are live
//   dynB > 0: the records were written under the live-group map of a round with dynB workgroups (LsLive)
__global__ void __launch_bounds__(256) ls_reduce_kernel(const double* __restrict__ part, double* __restrict__ red, int red_ld, const int* __restrict__ state,
                                                        int S, int first, int len, int all_signatures, int dynB = 0, int K = 0, int64_t N = 0,
                                                        int* active = nullptr) {
    // (the round's count of signatures that still ask for an evaluation starts from zero: ls_advance_kernel adds to it --
    // here instead of a memset launch of its own between the two kernels)
    if (active != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *active = 0;
    // grid (K, ceil(len / 256)): one entry per thread (a record is 66 + dim^2 doubles: with one workgroup per signature the
    // 17 MB of partials at c5 were read by 40 workgroups, 52 us per round)
    const int k = blockIdx.x;
    const bool live = all_signatures || state[k] == LS_NEED;
    const int e = first + (int)blockIdx.y * 256 + (int)threadIdx.x;
    if (e < first + len) {
        double t = 0.0;
        if (live && dynB > 0) {
            const LsLive m = ls_live_map(state, K, dynB, N, k / LS_GROUP);
            const int64_t base = (int64_t)(m.ord * LS_GROUP + k % LS_GROUP) * m.Sd;
            int s = 0;
            for (; s + 8 <= m.nchunks; s += 8) {  // (eight loads in flight per round)
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[(base + s + u) * LS_REC + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            for (; s < m.nchunks; ++s) t += part[(base + s) * LS_REC + e];
        } else if (live) {
            int s = 0;
            for (; s + 8 <= S; s += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[((int64_t)k * S + s + u) * LS_REC + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            for (; s < S; ++s) t += part[((int64_t)k * S + s) * LS_REC + e];
        }
        red[(int64_t)k * red_ld + e] = t;  // zero for signatures that asked for nothing: the all-reduce covers all K
    }
}

__global__ void ls_copy_sg_kernel(const double* __restrict__ red, int red_ld, double* __restrict__ sg, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * 64) sg[i] = red[(int64_t)(i / 64) * red_ld + 2 + (i & 63)];
}

// Hl [64][CORR_LD] <- a compact dim x dim Hessian (and, if asked, a copy of it to `copy`): eight loads in flight per lane
// and round (one load per loop iteration is a memory round trip per iteration: 25 of them for dim 40)
__device__ inline void ls_fill_hessian(double* Hl, const double* __restrict__ src, double* __restrict__ copy, int dim, int lane) {
    const int total = dim * dim;
    int row = lane / dim, col = lane - row * dim;  // position of element e = lane + 64 u, advanced by additions
    const int drow = 64 / dim, dcol = 64 - drow * dim;
    for (int e0 = lane; e0 < total + 64 * 7; e0 += 64 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[e0 + 64 * u < total ? e0 + 64 * u : total - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (e0 + 64 * u < total) {
                Hl[row * CORR_LD + col] = v[u];
                if (copy) copy[e0 + 64 * u] = v[u];
            }
            row += drow, col += dcol;
            if (col >= dim) col -= dim, ++row;
        }
        if (e0 - lane + 64 * 8 >= total) break;
    }
    __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations execute in order
}

// The evaluator of the replay: answers from the log of evaluations made so far (by point), asks for the first point it
// does not know and from then on reports "exhausted", which makes ncg::minimize unwind through its bounded loops.
struct ReplayEval {
    const double* ly;  // [n][64]
    const double* lf;  // [n]
    const double* lg;  // [n][64]
    const double* lH;  // [n][dim * dim]
    int n, dim, lane;
    int cursor;          // the log is in request order and the replay asks in the same order: next expected entry
    double variance;
    bool pending;
    double req;          // lane m: component of the point asked for
    double* Hl;          // LDS [64][CORR_LD]: the Hessian sum of the point fixed by prepare_hess
    double* vb;          // LDS [64]: the vector of a Hessian-vector product
    bool have_H;
    // the last LS_WIN entries of the record in LDS (one round trip for all of them at the start of the kernel: looked up in
    // global memory every find costs two dependent round trips, and a resumed replay only asks for recent entries)
    const double* wy;    // LDS [LS_WIN][64]
    const double* wg;    // LDS [LS_WIN][64]
    const double* wf;    // LDS [LS_WIN]
    int w0;              // first entry of the window
    int hl_entry;        // the record entry whose Hessian Hl holds (-1: none)

    __device__ inline bool is(int i, double y) const {
        const double have = i >= w0 ? wy[(i - w0) * 64 + lane] : ly[i * 64 + lane];
        return __all(lane >= dim || have == y);
    }
    __device__ inline double f_at(int i) const { return i >= w0 ? wf[i - w0] : lf[i]; }
    __device__ inline double g_at(int i) const { return i >= w0 ? wg[(i - w0) * 64 + lane] : lg[i * 64 + lane]; }
    __device__ inline int lookup(double y) {
        if (cursor < n && is(cursor, y)) return cursor++;
        if (cursor > 0 && is(cursor - 1, y)) return cursor - 1;  // the point just evaluated, asked for again (fun, then grad)
        // anywhere else in the record (the last match): no early exit, so that the loads of the scan are independent of one
        // another -- with one it is a chain of n memory round trips, paid in full whenever the point is new
        int found = -1;
        for (int i = 0; i < n; ++i) found = is(i, y) ? i : found;
        return found;
    }
    __device__ inline int find_or_ask(double y) {
        if (pending) return -1;
        const int i = lookup(y);
        if (i < 0) {
            pending = true;
            req = y;
        }
        return i;
    }
    __device__ inline double fun(double y) {
        const int i = find_or_ask(y);
        return i < 0 ? 0.0 : f_at(i);
    }
    __device__ inline double grad(double y) {
        const int i = find_or_ask(y);
        return i < 0 ? 0.0 : g_at(i);
    }
    __device__ inline void fun_grad(double y, double& f, double& g) {
        const int i = find_or_ask(y);
        f = i < 0 ? 0.0 : f_at(i);
        g = i < 0 ? 0.0 : g_at(i);
    }
    __device__ inline void prepare_hess(double x) {
        const int i = find_or_ask(x);
        have_H = i >= 0;
        if (!have_H) return;
        if (i == hl_entry) return;  // (the Hessian of the entry just recorded: put there by ls_advance_kernel)
        ls_fill_hessian(Hl, lH + (int64_t)i * dim * dim, nullptr, dim, lane);
        hl_entry = i;
    }
    // (Hessian at the fixed point) . v: the logged sum of w_n U_n U_n^T plus the prior's v / variance, in the summation
    // order of SignatureEmbeddingEval::hessp
    __device__ inline double hessp(double v) const {
        if (!have_H) return 0.0;
        // the vector through LDS, read back as broadcasts (a v_readlane per component stalls the FMA that consumes its
        // scalar result: ~50 cycles per component against ~10 here; same chain, same order)
        vb[lane] = v;
        __builtin_amdgcn_wave_barrier();
        double r = 0.0;
        const double* row = Hl + (lane < dim ? lane : 0) * CORR_LD;
#pragma unroll 8
        for (int j = 0; j < dim; ++j) r = __builtin_fma(row[j], vb[j], r);
        __builtin_amdgcn_wave_barrier();
        return lane < dim ? r + v / variance : 0.0;
    }
    __device__ inline bool exhausted() const { return pending; }
    __device__ inline int tag() const { return cursor; }
    __device__ inline void set_tag(int t) { cursor = t; }
};

// one wavefront per signature: log the evaluation that has just been reduced, replay the solve, ask or finish
__global__ void __launch_bounds__(64) ls_advance_kernel(LockstepParams q) {
    __shared__ double Hl[64 * CORR_LD];
    __shared__ double vbuf[64];
    __shared__ double wy[LS_WIN * 64], wg[LS_WIN * 64], wf[LS_WIN];
    const int k = blockIdx.x, lane = threadIdx.x;
    if (q.state[k] != LS_NEED) return;
#ifdef SALNMF_DEV_PROFILE
    long long at0 = __builtin_amdgcn_s_memtime(), at1 = 0, at2 = 0;
#endif
    const int dim = q.sig.dim;
    const double variance = q.sig.variance;
    double* ly = q.log_y + (int64_t)k * LS_EVAL_MAX * 64;
    double* lf = q.log_f + (int64_t)k * LS_EVAL_MAX;
    double* lg = q.log_g + (int64_t)k * LS_EVAL_MAX * 64;
    double* lH = q.log_H + (int64_t)k * LS_EVAL_MAX * dim * dim;
    const double* red = q.red + (int64_t)k * q.red_ld;
    // ---- the evaluation at the requested point (same arithmetic as SignatureEmbeddingEval::fun_grad)
    const int i = q.n_evals[k];
    const double y = q.req[k * 64 + lane];
    const int w0 = i + 1 > LS_WIN ? i + 1 - LS_WIN : 0;  // the window: entries w0 .. i (entry i is the one logged below)
    {
        double ty[LS_WIN], tg[LS_WIN], tf[LS_WIN];  // (all loads in flight together)
#pragma unroll
        for (int j = 0; j < LS_WIN - 1; ++j) {
            const int ent = w0 + j < i ? w0 + j : (i > 0 ? i - 1 : 0);
            ty[j] = ly[ent * 64 + lane], tg[j] = lg[ent * 64 + lane], tf[j] = lf[ent];
        }
#pragma unroll
        for (int j = 0; j < LS_WIN - 1; ++j)
            if (w0 + j < i) {
                wy[j * 64 + lane] = ty[j], wg[j * 64 + lane] = tg[j];
                if (lane == 0) wf[j] = tf[j];
            }
    }
    {
        // (the linear term sum_n aux[n][k] <U_n, y>: summed over the samples by the evaluation kernel, or -- the same number
        // to rounding -- <y, sg_k> with the solve's constant sg = aux^T U)
        double v = q.lin_from_sg ? ncg::wave_sum(y * q.sg[k * 64 + lane]) : red[0];
        v -= red[1];
        v -= ncg::wave_sum(y * y) / (2 * variance);
        double gg = -red[2 + lane];
        gg += q.sg[k * 64 + lane];
        gg -= y / variance;
        ly[i * 64 + lane] = y;
        lg[i * 64 + lane] = lane < dim ? -gg : 0.0;
        if (lane == 0) lf[i] = -v;
        wy[(i - w0) * 64 + lane] = y;
        wg[(i - w0) * 64 + lane] = lane < dim ? -gg : 0.0;
        if (lane == 0) wf[i - w0] = -v;
        ls_fill_hessian(Hl, red + 66, lH + (int64_t)i * dim * dim, dim, lane);  // (to the record and, for the replay, to LDS)
    }
    __threadfence_block();  // the log entries written above are read back below by other lanes of this wave
#ifdef SALNMF_DEV_PROFILE
    at1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- replay
    ReplayEval ev;
    ev.ly = ly;
    ev.lf = lf;
    ev.lg = lg;
    ev.lH = lH;
    ev.n = i + 1;
    ev.dim = dim;
    ev.lane = lane;
    ev.variance = variance;
    ev.cursor = 0;
    ev.pending = false;
    ev.req = 0.0;
    ev.Hl = Hl;
    ev.vb = vbuf;
    ev.wy = wy, ev.wg = wg, ev.wf = wf, ev.w0 = w0;
    ev.hl_entry = i;
    ev.have_H = false;
    double x = q.x0[k * 64 + lane];
    // resume at the top of the last Newton iteration the previous rounds reached (its CG solve and the line search's
    // earlier trial points are replayed from the record; the iterations before it are not: a round's advance used to cost
    // all of them again, 60 us per round at c5)
    double* cpm = q.cp + (int64_t)k * LS_CP;
    ncg::Checkpoint cp;
    cp.valid = i > 0 && cpm[128 + 7] != 0.0;
    if (cp.valid) {
        cp.xk = cpm[lane], cp.g_next = cpm[64 + lane];
        cp.old_fval = cpm[128], cp.old_old_fval = cpm[129], cp.update_l1norm = cpm[130];
        cp.k = (int)cpm[131], cp.have_old_old = (int)cpm[132], cp.have_g_next = (int)cpm[133], cp.tag = (int)cpm[134];
    }
    int n_newton = 0;
    const int st = ncg::minimize(ev, x, dim, q.sig.maxiter, &n_newton, &cp);
#ifdef SALNMF_DEV_PROFILE
    at2 = __builtin_amdgcn_s_memtime();
    if (q.prof != nullptr && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(q.prof) + 12, (unsigned long long)(at1 - at0));
        atomicAdd(reinterpret_cast<unsigned long long*>(q.prof) + 13, (unsigned long long)(at2 - at1));
        atomicAdd(reinterpret_cast<unsigned long long*>(q.prof) + 14, 1ull);
    }
#endif
    if (ev.pending) {
        if (cp.valid) {
            cpm[lane] = cp.xk, cpm[64 + lane] = cp.g_next;
            if (lane == 0) {
                cpm[128] = cp.old_fval, cpm[129] = cp.old_old_fval, cpm[130] = cp.update_l1norm;
                cpm[131] = cp.k, cpm[132] = cp.have_old_old, cpm[133] = cp.have_g_next, cpm[134] = cp.tag, cpm[135] = 1.0;
            }
        } else if (lane == 0) {
            cpm[135] = 0.0;
        }
        q.req[k * 64 + lane] = lane < dim ? ev.req : 0.0;
        if (lane == 0) {
            q.n_evals[k] = i + 1;
            if (i + 1 >= LS_EVAL_MAX) {
                q.state[k] = LS_FALLBACK;  // a runaway solve: the single-kernel form (with its own budget) takes it over
            } else {
                atomicAdd(q.active, 1);
            }
        }
        return;
    }
    if (x > 0.0 && x < kEps) x = kEps;
    if (x < 0.0 && x > -kEps) x = -kEps;
    if (lane < dim) q.sig.L[k * dim + lane] = x;
    if (lane == 0) {
        q.n_evals[k] = i + 1;
        q.state[k] = LS_DONE;
        if (q.sig.status) q.sig.status[k] = st;
    }
}

}  // namespace salnmf
