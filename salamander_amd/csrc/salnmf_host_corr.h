// Part of salnmf.hip's translation unit (included there, inside its extern "C" block; not a stand-alone header):
// CorrNMF / MultimodalCorrNMF entry points: dense pieces, batched sample solves, lockstep signature solves.
// Split out of salnmf.hip in round 5 for readability only -- one translation unit, the same static helpers and macros.

// ------------------------------------------------------------------------------------ CorrNMF (row f1)

int salnmf_corr_configure(salnmf_engine* e, int dim_embeddings) {
    // feature blocks (n_features > 96): the two passes over X run block by block (corr_compute_aux, corr_poisson_llh);
    // everything else of CorrNMF is K- and dim-sized
    if (e && e->NC > 1) return single_block(e, "CorrNMF");
    if (!e) return fail("null engine");
    if (dim_embeddings < 1 || dim_embeddings > CORR_DMAX) return fail("dim_embeddings must be in [1, %d], got %d", CORR_DMAX, dim_embeddings);
    CK(enter(e));
    HIPCK(hipStreamSynchronize(e->stream));
    double** bufs[] = {&e->alpha, &e->beta, &e->Lemb, &e->Uemb, &e->aux, &e->xrowsum, &e->corrpart};
    for (double** b : bufs) {
        if (*b) HIPCK(hipFree(*b));
        *b = nullptr;
    }
    e->dim = 0;
    const size_t Np = e->Np, K = e->K, d = dim_embeddings;
    e->cgrid = (int)std::min<int64_t>(1024, (e->Np + CORR_TILE - 1) / CORR_TILE);
    HIPCK(hipMalloc(&e->alpha, Np * sizeof(double)));
    HIPCK(hipMalloc(&e->beta, K * sizeof(double)));
    HIPCK(hipMalloc(&e->Lemb, K * d * sizeof(double)));
    HIPCK(hipMalloc(&e->Uemb, (size_t)e->N * d * sizeof(double)));
    HIPCK(hipMalloc(&e->aux, Np * e->KP * sizeof(double)));
    HIPCK(hipMalloc(&e->xrowsum, Np * sizeof(double)));
    HIPCK(hipMalloc(&e->corrpart, (size_t)e->cgrid * 64 * sizeof(double)));
    HIPCK(hipMemsetAsync(e->alpha, 0, Np * sizeof(double), e->stream));
    HIPCK(hipMemsetAsync(e->beta, 0, K * sizeof(double), e->stream));
    HIPCK(hipMemsetAsync(e->Lemb, 0, K * d * sizeof(double), e->stream));
    HIPCK(hipMemsetAsync(e->Uemb, 0, (size_t)e->N * d * sizeof(double), e->stream));
    HIPCK(hipMemsetAsync(e->aux, 0, Np * e->KP * sizeof(double), e->stream));
    e->dim = dim_embeddings;
    e->xrowsum_valid = false;
    return 0;
}

static int corr_ready(salnmf_engine* e) {
    if (!e) return fail("null engine");
    if (e->dim == 0) return fail("salnmf_corr_configure has not been called on this engine");
    CK(enter(e));
    return 0;
}

int salnmf_corr_upload(salnmf_engine* e, int which, const double* src) {
    CK(corr_ready(e));
    if (!src) return fail("null argument");
    switch (which) {
        case SALNMF_CORR_SIGNATURE_SCALINGS: return upload(e, e->beta, src, (size_t)e->K);
        case SALNMF_CORR_SAMPLE_SCALINGS: return upload_padded(e, e->alpha, src, 1, 1, 0.0, 0.0, 0.0);
        case SALNMF_CORR_SIGNATURE_EMBEDDINGS: return upload(e, e->Lemb, src, (size_t)e->K * e->dim);
        case SALNMF_CORR_SAMPLE_EMBEDDINGS: return upload(e, e->Uemb, src, (size_t)e->N * e->dim);
        case SALNMF_CORR_AUX: return upload_padded(e, e->aux, src, e->K, e->KP, 0.0, 0.0, 0.0);
        default: return fail("unknown CorrNMF buffer %d", which);
    }
}

int salnmf_corr_download(salnmf_engine* e, int which, double* dst) {
    CK(corr_ready(e));
    if (!dst) return fail("null argument");
    switch (which) {
        case SALNMF_CORR_SIGNATURE_SCALINGS: return download(e, dst, e->beta, (size_t)e->K);
        case SALNMF_CORR_SAMPLE_SCALINGS: return download(e, dst, e->alpha, (size_t)e->N);
        case SALNMF_CORR_SIGNATURE_EMBEDDINGS: return download(e, dst, e->Lemb, (size_t)e->K * e->dim);
        case SALNMF_CORR_SAMPLE_EMBEDDINGS: return download(e, dst, e->Uemb, (size_t)e->N * e->dim);
        case SALNMF_CORR_AUX: return download_padded(e, dst, e->aux, e->K, e->KP);
        default: return fail("unknown CorrNMF buffer %d", which);
    }
}

static CorrParams corr_params(salnmf_engine* e) {
    CorrParams p;
    p.alpha = e->alpha;
    p.beta = e->beta;
    p.L = e->Lemb;
    p.U = e->Uemb;
    p.xrowsum = e->xrowsum;
    p.out = nullptr;
    p.N = e->N;
    p.Np = e->Np;
    p.K = e->K;
    p.KP = e->KP;
    p.dim = e->dim;
    return p;
}

int salnmf_corr_update_sample_scalings(salnmf_engine* e) {
    CK(corr_ready(e));
    if (!e->xrowsum_valid) {
        for (int b = 0; b < e->NB; ++b)  // (block b's sum joins the earlier blocks')
            hipLaunchKernelGGL(rowsum_X_kernel, dim3(1024), dim3(256), 0, e->stream, e->X + (size_t)b * e->Np * VMAX, e->Np, VMAX, b > 0 ? 1 : 0, e->xrowsum);
        HIPCK(hipGetLastError());
        e->xrowsum_valid = true;
    }
    CorrParams p = corr_params(e);
    p.out = e->alpha;
    p.alpha = nullptr;
    launch_corr_logit<0>(e, p);
    HIPCK(hipGetLastError());
    return 0;
}

int salnmf_corr_compute_exposures(salnmf_engine* e) {
    CK(corr_ready(e));
    CorrParams p = corr_params(e);
    p.out = e->H;
    e->h_pending = false;  // H is overwritten in full
    launch_corr_logit<1>(e, p);
    HIPCK(hipGetLastError());
    return 0;
}

int salnmf_corr_compute_aux(salnmf_engine* e) {
    CK(corr_ready(e));
    if (e->NB > 1) {
        // U = R W^T summed over the feature blocks, aux = H * U unclipped by the last block's launch; the numerators of
        // update_signatures block by block into Gblk (applied by salnmf_corr_update_signatures)
        return blocked_joint_passes(e, e->aux, 0.0, false);  // (one pass per block for both)
    }
    FusedParams p = fused_params(e);
    p.wkl = nullptr;  // CorrNMF is unweighted (corrnmf_det.py:80-85)
    p.wlh = nullptr;
    p.Hout = e->aux;
    p.hfloor = 0.0;
    CK((launch_fused<true, true, false>(e, p)));
    // the same pass produced G = (X/(HW))^T H for update_signatures: reduce it now (all ranks), apply later
    CK(launch_tail(e, e->grid, e->red, 0, 0, 0));
    return allreduce(e, e->red, (size_t)e->K * e->V);
}

int salnmf_corr_update_signatures(salnmf_engine* e, int n_given) {
    CK(corr_ready(e));
    if (n_given < 0 || n_given > e->K) return fail("n_given out of range");
    if (n_given >= e->K) return 0;  // _utils_klnmf.py:204-205
    if (e->NB > 1) return blocked_finish_W(e, n_given, SALNMF_CLIP_NON_GIVEN);
    return launch_tail(e, 0, e->red, n_given, SALNMF_CLIP_NON_GIVEN, 1);
}

int salnmf_corr_update_signature_scalings(salnmf_engine* e) {
    CK(corr_ready(e));
    const int K = e->K;
    // first_k = sum_n aux[n][k]
    const int pgrid = (int)std::min<int64_t>(512, (e->N + 255) / 256);
    CK(ensure_scratch(e, (size_t)512 * 64 + 2 * 64));
    double* part = e->scratch;
    double* first = e->scratch + (size_t)512 * 64;
    double* second = first + 64;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(pgrid), dim3(256), 0, e->stream, e->aux, e->N, e->KP, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(K), dim3(256), 0, e->stream, part, pgrid, e->KP, K, first);
    // second_k = sum_n exp(alpha_n + <L_k, U_n>)
    CorrParams p = corr_params(e);
    p.out = e->corrpart;
    launch_corr_logit<2>(e, p);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(K), dim3(256), 0, e->stream, e->corrpart, corr_logit_grid(e), K, K, second);
    HIPCK(hipGetLastError());
    CK(allreduce(e, first, 128));  // first and second are adjacent
    hipLaunchKernelGGL(corr_log_ratio_kernel, dim3(1), dim3(64), 0, e->stream, first, second, K, e->beta);
    HIPCK(hipGetLastError());
    return 0;
}

static int sample_embeddings_impl(salnmf_engine* const* engines, int n_engines, double variance, int maxiter, int* status_out) {
    if (!engines || n_engines < 1 || n_engines > CORR_MODS) return fail("between 1 and %d engines expected", CORR_MODS);
    salnmf_engine* e0 = engines[0];
    CK(corr_ready(e0));
    if (!(variance > 0.0)) return fail("variance must be positive");
    SampleEmbeddingParams p;
    int terms = 0;
    for (int i = 0; i < CORR_MODS; ++i) {
        salnmf_engine* e = engines[i < n_engines ? i : 0];
        if (i < n_engines) {
            if (!e || e->dim == 0) return fail("engine %d is not configured for CorrNMF", i);
            if (e->device != e0->device || e->N != e0->N || e->dim != e0->dim)
                return fail("engine %d differs from engine 0 in device, n_samples or dim_embeddings", i);
            terms += e->K;
        }
        p.aux[i] = e->aux;
        p.alpha[i] = e->alpha;
        p.beta[i] = e->beta;
        p.L[i] = e->Lemb;
        p.K[i] = i < n_engines ? e->K : 0;
        p.KP[i] = e->KP;
    }
    if (terms > CORR_TERMS) return fail("at most %d signatures over all modalities, got %d", CORR_TERMS, terms);
    p.n_mod = n_engines;
    p.U = e0->Uemb;
    p.status = nullptr;
    p.variance = variance;
    p.N = e0->N;
    p.dim = e0->dim;
    p.maxiter = maxiter > 0 ? maxiter : 200 * e0->dim;  // scipy's default: 200 * len(x0)
    // the modalities' engines have their own streams: everything they queued must be complete first
    for (int i = 1; i < n_engines; ++i) HIPCK(hipStreamSynchronize(engines[i]->stream));
    int* dstatus = nullptr;
    if (status_out) {
        HIPCK(hipMalloc(&dstatus, (size_t)e0->N * sizeof(int)));
        p.status = dstatus;
    }
    // sixteen solves per wavefront on the fp64 MFMA units where an instantiation covers the shape (csrc/salnmf_corr_batched.hip:
    // <= 80 terms, dim <= 48), else one wavefront per sample
    if (!(e0->batched_samples && launch_sample_embeddings_batched(p, terms, e0->stream))) {
        const int grid = (int)std::min<int64_t>((e0->N + 3) / 4, 8192);
        const size_t lds_bytes = (size_t)terms * (e0->dim | 1) * sizeof(double);  // the term matrix (corr_sample_embeddings_kernel)
        if (terms <= 64)
            hipLaunchKernelGGL(corr_sample_embeddings_kernel<1>, dim3(grid), dim3(CORR_BLOCK), lds_bytes, e0->stream, p);
        else
            hipLaunchKernelGGL(corr_sample_embeddings_kernel<2>, dim3(grid), dim3(CORR_BLOCK), lds_bytes, e0->stream, p);
    }
    int rc = 0;
    if (hipGetLastError() != hipSuccess) rc = fail("corr_sample_embeddings_kernel launch failed");
    // the sample embeddings are shared: every modality's engine gets the result
    for (int i = 1; i < n_engines && !rc; ++i)
        if (hipMemcpyAsync(engines[i]->Uemb, e0->Uemb, (size_t)e0->N * e0->dim * sizeof(double), hipMemcpyDeviceToDevice, e0->stream) != hipSuccess)
            rc = fail("copy of the shared sample embeddings failed");
    if (!rc && (status_out || n_engines > 1)) {
        if (status_out && hipMemcpyAsync(status_out, dstatus, (size_t)e0->N * sizeof(int), hipMemcpyDeviceToHost, e0->stream) != hipSuccess)
            rc = fail("status download failed");
        if (!rc && hipStreamSynchronize(e0->stream) != hipSuccess) rc = fail("hipStreamSynchronize failed");
    }
    if (dstatus) (void)hipFree(dstatus);
    return rc;
}

int salnmf_corr_update_sample_embeddings(salnmf_engine* e, double variance, int maxiter, int* status_out) {
    return sample_embeddings_impl(&e, 1, variance, maxiter, status_out);
}

int salnmf_corr_update_sample_embeddings_multi(salnmf_engine* const* engines, int n_engines, double variance, int maxiter,
                                               int* status_out) {
    return sample_embeddings_impl(engines, n_engines, variance, maxiter, status_out);
}

// room for the gathered sample-side inputs of the signature-embedding solves
static int ensure_gathered(salnmf_engine* e, size_t rows) {
    if (e->g_rows >= rows && e->g_dim == e->dim) return 0;
    for (double** b : {&e->gU, &e->galpha, &e->gaux}) {
        if (*b) HIPCK(hipFree(*b));
        *b = nullptr;
    }
    e->g_rows = 0;
    HIPCK(hipMalloc(&e->gU, rows * e->dim * sizeof(double)));
    HIPCK(hipMalloc(&e->galpha, rows * sizeof(double)));
    HIPCK(hipMalloc(&e->gaux, rows * e->KP * sizeof(double)));
    e->g_rows = rows;
    e->g_dim = e->dim;
    return 0;
}

// ---- lockstep form of the signature solves (salnmf_corr_lockstep.h): evaluation rounds over (chunks x signatures)
// workgroups, the solvers replayed from their logs between rounds.  `shard`: the rows are this rank's shard and the
// reduced sums of every round are all-reduced (objective, gradient and Hessian are sums over samples).
// below: the single-kernel form (one workgroup per signature passes over all samples for every evaluation; a lockstep solve
// costs ~0.4 ms of launches and read-backs whatever the size).  Measured crossover 1 500 - 2 000 samples at 10 signatures,
// lower with more (profiles/r04/corr_sizes.txt: 5 000 x 10: 1.70 -> 0.72 ms per update, 12 000 x 30: 4.73 -> 1.23 ms)
constexpr int64_t LS_MIN_ROWS = 2048;

static int lockstep_signature_solves(salnmf_engine* e, const double* U, const double* alpha, const double* aux, int64_t n_rows, double variance,
                                     int maxiter, int* status_out, bool shard) {
    const int K = e->K, dim = e->dim;
    const int n_cus = e->cus;  // (cached at salnmf_create: a properties query per solve costs as much as a small solve)
    const int64_t max_chunks = (n_rows + SIGT - 1) / SIGT;
    // dim <= 48: LS_GROUP signatures share a staged tile of U (salnmf_corr_lockstep.h), so there are fewer, longer rows of
    // workgroups and more chunks; otherwise one signature per workgroup
    const bool multi = dim <= 48;
    const int groups = multi ? (K + LS_GROUP - 1) / LS_GROUP : K;
    // chunks per signature (group): as many as it takes to fill the chip, within 64 MB of partial records
    const int64_t s_cap = std::max<int64_t>(1, (int64_t)(64u << 20) / (int64_t)((size_t)LS_GROUP * groups * LS_REC * sizeof(double)));
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>({multi ? 128 : 16, n_cus / groups, max_chunks, s_cap}));
    const int64_t chunk = ((n_rows + S - 1) / S + SIGT - 1) / SIGT * SIGT;
    // one allocation: [x0 | req | sg] (K x 64 each), part (K S REC), red (K REC), log_y, log_g (K EVAL 64), log_f (K EVAL), log_H (K EVAL dim^2)
    const size_t n_part = (size_t)(multi ? LS_GROUP * groups : K) * S * LS_REC;  // (records by (group ordinal, slot, chunk) under the live-group map)
    const size_t nd = (size_t)3 * K * 64 + n_part + (size_t)K * LS_REC + (size_t)2 * K * LS_EVAL_MAX * 64 +
                      (size_t)K * LS_EVAL_MAX + (size_t)K * LS_EVAL_MAX * dim * dim + (size_t)K * LS_CP;
    if (e->ls_doubles < nd || e->ls_S != S || e->ls_dim != dim) {
        if (e->ls_buf) HIPCK(hipFree(e->ls_buf));
        e->ls_buf = nullptr;
        e->ls_doubles = 0;
        HIPCK(hipMalloc(&e->ls_buf, nd * sizeof(double)));
        e->ls_doubles = nd;
        e->ls_S = S;
        e->ls_dim = dim;
    }
    if (!e->ls_int) HIPCK(hipMalloc(&e->ls_int, (size_t)(3 * 64 + 8) * sizeof(int)));
    LockstepParams q;
    q.sig.aux = aux;
    q.sig.alpha = alpha;
    q.sig.beta = e->beta;
    q.sig.U = U;
    q.sig.L = e->Lemb;
    q.sig.only = nullptr;
    q.sig.variance = variance;
    q.sig.N = n_rows;
    q.sig.Np = n_rows;
    q.sig.K = K;
    q.sig.KP = e->KP;
    q.sig.dim = dim;
    q.sig.maxiter = maxiter > 0 ? maxiter : 200 * dim;
    q.S = S;
    q.chunk = chunk;
    double* b = e->ls_buf;
    q.x0 = b; b += (size_t)K * 64;
    q.req = b; b += (size_t)K * 64;
    q.sg = b; b += (size_t)K * 64;
    q.part = b; b += n_part;
    q.red = b; b += (size_t)K * LS_REC;
    q.red_ld = 66 + dim * dim;  // (the live part of a record: what a sharded solve sends through the all-reduce -- 130 doubles per signature at dim 8, 1 666 at dim 40, of the 4 162 a record is laid out for)
    q.log_y = b; b += (size_t)K * LS_EVAL_MAX * 64;
    q.log_g = b; b += (size_t)K * LS_EVAL_MAX * 64;
    q.log_f = b; b += (size_t)K * LS_EVAL_MAX;
    q.log_H = b; b += (size_t)K * LS_EVAL_MAX * dim * dim;
    q.cp = b;
    q.lin_from_sg = multi && dim % 16 != 0;  // (ls_eval_packed_kernel)
    q.dyn = q.lin_from_sg;
    q.prof = nullptr;
#ifdef SALNMF_DEV_PROFILE
    static long long* ls_prof = nullptr;  // (development aid: one buffer per process, printed after every solve)
    if (!ls_prof) HIPCK(hipMalloc(&ls_prof, 16 * sizeof(long long)));
    HIPCK(hipMemsetAsync(ls_prof, 0, 16 * sizeof(long long), e->stream));
    q.prof = ls_prof;
#endif
    q.state = e->ls_int;
    q.n_evals = e->ls_int + 64;
    q.sig.status = e->ls_int + 128;
    q.active = e->ls_int + 192;
    int* hactive = reinterpret_cast<int*>(e->hpin);
    const dim3 grid(S, groups);
    // start: sg = sum_n aux[n][k] U[n][:] and the first requests (the start points)
    if (multi) {
        // sg = aux^T U as one MFMA product, four workgroups per CU; their partial sums [wg][K][64] borrow
        // the front of q.part (K S LS_REC doubles)
        const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)4 * n_cus, (n_rows + 255) / 256, (int64_t)S * LS_REC / 64}));
        const int64_t rows_per_wg = ((n_rows + nwg - 1) / nwg + 255) / 256 * 256;
        hipLaunchKernelGGL(ls_begin_mfma_kernel, dim3(nwg), dim3(256), 0, e->stream, q, q.part, rows_per_wg);
        hipLaunchKernelGGL(ls_reduce_sg_kernel, dim3(K), dim3(1024), 0, e->stream, q.part, q.red, q.red_ld, nwg, K, dim);
    } else {
        hipLaunchKernelGGL(ls_begin_kernel, grid, dim3(SIGT), 0, e->stream, q);
        hipLaunchKernelGGL(ls_reduce_kernel, dim3(K, 1), dim3(256), 0, e->stream, q.part, q.red, q.red_ld, q.state, S, 2, 64, 1);
    }
    HIPCK(hipGetLastError());
    if (shard) CK(allreduce(e, q.red, (size_t)K * q.red_ld));
    hipLaunchKernelGGL(ls_copy_sg_kernel, dim3((K * 64 + 255) / 256), dim3(256), 0, e->stream, q.red, q.red_ld, q.sg, K);
    HIPCK(hipGetLastError());
    const int rec = 66 + dim * dim;
    for (hipEvent_t& ev : e->ls_ev)
        if (!ev) HIPCK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // one round: evaluation at every live signature's requested point, fixed-order sum of the partials, the solvers advance;
    // the number of signatures that still want an evaluation travels to the host behind it (slot and event r & 1)
    auto launch_round = [&](int r) -> int {
        if (multi && dim % 16 != 0) {
            // the last block column's live columns of the group's signatures side by side (ls_eval_packed_kernel)
            const int tail_cols = dim + 1 - 16 * ((dim + 15) / 16 - 1);
            const int npk = (LS_GROUP * tail_cols + 15) / 16;
            // <3, 40, ..>: three row tiles and three packed tiles as compile-time constants, the next tile of U prefetched
            // into 40 registers per lane (c5's shape: dim 33 .. 40)
            // (even dim: the tile by LDS-DMA, per wave, in double-buffered halves -- no staging registers, no barrier in the loop)
            if (npk == 3 && dim > 32 && dim <= 40 && dim % 2 == 0) hipLaunchKernelGGL((ls_eval_packed_kernel<3, 0, true, true, true>), grid, dim3(SIGT), 0, e->stream, q);
            else if (npk == 3 && dim > 32 && dim <= 40) hipLaunchKernelGGL((ls_eval_packed_kernel<3, 40, true, true>), grid, dim3(SIGT), 0, e->stream, q);
            else if (npk <= 3) hipLaunchKernelGGL((ls_eval_packed_kernel<3, 0, false, false>), grid, dim3(SIGT), 0, e->stream, q);
            else hipLaunchKernelGGL((ls_eval_packed_kernel<5, 0, false, false>), grid, dim3(SIGT), 0, e->stream, q);
        } else if (multi) hipLaunchKernelGGL(ls_eval_multi_kernel, grid, dim3(SIGT), 0, e->stream, q);
        else hipLaunchKernelGGL(ls_eval_kernel, grid, dim3(SIGT), 0, e->stream, q);
        hipLaunchKernelGGL(ls_reduce_kernel, dim3(K, (rec + 255) / 256), dim3(256), 0, e->stream, q.part, q.red, q.red_ld, q.state, S, 0, rec, 0, q.dyn ? S * groups : 0, K,
                           (int64_t)n_rows, q.active);
        HIPCK(hipGetLastError());
        if (shard) CK(allreduce(e, q.red, (size_t)K * q.red_ld));
        hipLaunchKernelGGL(ls_advance_kernel, dim3(K), dim3(64), 0, e->stream, q);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(hactive + (r & 1), q.active, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipEventRecord(e->ls_ev[r & 1], e->stream));
        return 0;
    };
    // The host only decides whether another round is needed.  Round r + 1 is queued BEFORE round r's count is read: if
    // everything had finished it is a no-op on the device (the kernels return for signatures that want nothing), and
    // otherwise the round trip of the count hides behind it (25 us per round before).
    bool finished = false;
    CK(launch_round(0));
    for (int round = 0; round < LS_EVAL_MAX + 2; ++round) {
        if (round + 1 < LS_EVAL_MAX + 2) CK(launch_round(round + 1));
        HIPCK(hipEventSynchronize(e->ls_ev[round & 1]));
        if (hactive[round & 1] == 0) {
            finished = true;
            break;
        }
    }
    if (!finished) return fail("lockstep signature solves did not terminate");
#ifdef SALNMF_DEV_PROFILE
    {
        long long h[16];
        HIPCK(hipMemcpyAsync(h, q.prof, sizeof h, hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipStreamSynchronize(e->stream));
        const double w = h[7] > 0 ? (double)h[7] : 1.0;
        fprintf(stderr, "[ls_eval_packed K=%d dim=%d] shader-clock ticks per wave and launch: tile to LDS %.0f + barrier %.0f, load issue %.0f, logit products %.0f, "
                        "weights %.0f, dense %.0f, packed %.0f, end-of-tile barrier %.0f, finish %.0f (%.0f wave-launches)\n", K, dim, h[8] / w, h[0] / w, h[9] / w, h[1] / w,
                h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, w);
        const double wa = h[14] > 0 ? (double)h[14] : 1.0;
        fprintf(stderr, "[ls_advance] shader-clock ticks per wave and launch: record + window %.0f, replay %.0f (%.0f wave-launches)\n", h[12] / wa, h[13] / wa, wa);
    }
#endif
    // runaway solves (log full) are finished by the single-kernel form with its own evaluation budget -- on the rows at hand
    // (a sharded engine reaches this point on every rank alike, the decisions being identical)
    std::vector<int> st(64 * 3);
    HIPCK(hipMemcpyAsync(st.data(), e->ls_int, st.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    bool fallback = false;
    for (int k = 0; k < K; ++k) fallback |= st[k] == LS_FALLBACK;
    if (fallback) {
        if (shard) return fail("a signature-embedding solve exceeded %d evaluations on a sample-sharded engine", LS_EVAL_MAX);
        SignatureEmbeddingParams p = q.sig;
        p.only = q.state;
        hipLaunchKernelGGL(corr_signature_embeddings_kernel, dim3(K), dim3(SIGT), 0, e->stream, p);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(st.data() + 128, q.sig.status, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipStreamSynchronize(e->stream));
    }
    if (status_out)
        for (int k = 0; k < K; ++k) status_out[k] = st[128 + k];
    return 0;
}

// One Newton-CG solve per signature over n_rows samples whose embeddings / scalings / aux rows are at U, alpha, aux
static int launch_signature_solves(salnmf_engine* e, const double* U, const double* alpha, const double* aux, int64_t n_rows,
                                   double variance, int maxiter, int* status_out) {
    SignatureEmbeddingParams p;
    p.aux = aux;
    p.alpha = alpha;
    p.beta = e->beta;
    p.U = U;
    p.L = e->Lemb;
    p.status = nullptr;
    p.only = nullptr;
    p.variance = variance;
    p.N = n_rows;
    p.Np = n_rows;
    p.K = e->K;
    p.KP = e->KP;
    p.dim = e->dim;
    p.maxiter = maxiter > 0 ? maxiter : 200 * e->dim;
    int* dstatus = nullptr;
    if (status_out) {
        HIPCK(hipMalloc(&dstatus, (size_t)e->K * sizeof(int)));
        p.status = dstatus;
    }
    hipLaunchKernelGGL(corr_signature_embeddings_kernel, dim3(e->K), dim3(SIGT), 0, e->stream, p);
    int rc = 0;
    if (hipGetLastError() != hipSuccess) rc = fail("corr_signature_embeddings_kernel launch failed");
    if (!rc && status_out) {
        if (hipMemcpyAsync(status_out, dstatus, (size_t)e->K * sizeof(int), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess)
            rc = fail("status download failed");
    }
    if (dstatus) (void)hipFree(dstatus);
    return rc;
}

// A signature embedding depends on ALL samples (its objective, gradient and Hessian are sums over samples), and
// its Newton-CG solve takes data-dependent decisions after every one of them.  With the sample axis sharded there are two
// forms (salnmf_corr_update_signature_embeddings picks):
//   * from LS_MIN_ROWS samples per rank on (every BASELINE-sized cohort) the K solvers advance in LOCKSTEP rounds on the
//     local rows and the 1 + dim + dim^2 sums per signature of every round are all-reduced (lockstep_signature_solves):
//     identical sums, hence identical decisions and embeddings, on every rank;
//   * smaller cohorts gather the sample-side inputs -- U (N x dim), alpha (N) and aux (N x K) -- ONCE per update
//     (all-gather with per-rank counts = one broadcast per rank inside an RCCL group, below), after which every rank runs
//     all K solves on identical inputs in the sample order of an unsharded engine.
static int gather_sample_side(salnmf_engine* e) {
    CK(ensure_gathered(e, (size_t)e->N_total));
    NCCLCK(ncclGroupStart());
    int64_t off = 0;
    for (int r = 0; r < e->n_ranks; ++r) {
        const size_t n = (size_t)e->shard_N[r];
        NCCLCK(ncclBroadcast(e->Uemb, e->gU + off * e->dim, n * e->dim, ncclDouble, r, e->comm, e->stream));
        NCCLCK(ncclBroadcast(e->alpha, e->galpha + off, n, ncclDouble, r, e->comm, e->stream));
        NCCLCK(ncclBroadcast(e->aux, e->gaux + off * e->KP, n * e->KP, ncclDouble, r, e->comm, e->stream));
        off += (int64_t)n;
    }
    NCCLCK(ncclGroupEnd());
    return 0;
}

int salnmf_corr_update_signature_embeddings(salnmf_engine* e, double variance, int maxiter, int* status_out) {
    CK(corr_ready(e));
    if (!(variance > 0.0)) return fail("variance must be positive");
    if (sharded(e)) {
        // sample-sharded: with enough samples the solves run in lockstep on the local rows and the sums of every
        // evaluation are all-reduced (66 + dim^2 per signature: through the peer exchange where that fits its inbox, else
        // through RCCL); small problems gather the sample side once (RCCL) and solve on identical inputs
        if (e->lockstep && e->N_total >= LS_MIN_ROWS * e->n_ranks)
            return lockstep_signature_solves(e, e->Uemb, e->alpha, e->aux, e->N, variance, maxiter, status_out, true);
        if (!e->comm) return fail("the sharded signature-embedding solves of a small cohort gather through the RCCL communicator: call salnmf_comm_init");
        CK(gather_sample_side(e));
        return launch_signature_solves(e, e->gU, e->galpha, e->gaux, e->N_total, variance, maxiter, status_out);
    }
    if (e->lockstep && e->N >= LS_MIN_ROWS) return lockstep_signature_solves(e, e->Uemb, e->alpha, e->aux, e->N, variance, maxiter, status_out, false);
    return launch_signature_solves(e, e->Uemb, e->alpha, e->aux, e->N, variance, maxiter, status_out);
}

int salnmf_corr_update_signature_embeddings_from(salnmf_engine* e, int64_t n_all, const double* U_all, const double* alpha_all,
                                                 const double* aux_all, double variance, int maxiter, int* status_out) {
    CK(corr_ready(e));
    if (!U_all || !alpha_all || !aux_all) return fail("null argument");
    if (n_all < 1) return fail("n_all must be positive");
    if (!(variance > 0.0)) return fail("variance must be positive");
    CK(ensure_gathered(e, (size_t)n_all));
    CK(upload(e, e->gU, U_all, (size_t)n_all * e->dim));
    CK(upload(e, e->galpha, alpha_all, (size_t)n_all));
    CK(ensure_scratch(e, (size_t)n_all * e->K));
    CK(upload(e, e->scratch, aux_all, (size_t)n_all * e->K));
    hipLaunchKernelGGL(pad_kernel, dim3(2048), dim3(256), 0, e->stream, e->gaux, e->scratch, n_all, e->K, n_all, e->KP, 0.0, 0.0, 0.0);
    HIPCK(hipGetLastError());
    // (the form an engine that holds all n_all samples would use: the same bits as its result)
    if (e->lockstep && n_all >= LS_MIN_ROWS) return lockstep_signature_solves(e, e->gU, e->galpha, e->gaux, n_all, variance, maxiter, status_out, false);
    return launch_signature_solves(e, e->gU, e->galpha, e->gaux, n_all, variance, maxiter, status_out);
}

int salnmf_corr_embedding_sumsq(salnmf_engine* e, double* out2) {
    CK(corr_ready(e));
    if (!out2) return fail("null argument");
    const int g = 256;
    CK(ensure_scratch(e, (size_t)2 * g));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(g), dim3(256), 0, e->stream, e->Lemb, (int64_t)e->K * e->dim, e->scratch);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(g), dim3(256), 0, e->stream, e->Uemb, e->N * (int64_t)e->dim, e->scratch + g);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->scratch, g, 1, 1, e->scal + 5);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->scratch + g, g, 1, 1, e->scal + 6);
    HIPCK(hipGetLastError());
    CK(allreduce(e, e->scal + 6, 1));  // the sample embeddings are sharded, the signature embeddings replicated
    return read_scalars(e, 5, 2, out2);
}

int salnmf_corr_poisson_llh(salnmf_engine* e, double* out) {
    if (!e || !out) return fail("null argument");
    CK(enter(e));
    if (!e->lgam_valid) {
        const int g = 1024;
        CK(ensure_scratch(e, (size_t)g + 1));
        CK(ensure_scratch(e, (size_t)g * e->NB + 1));
        for (int b = 0; b < e->NB; ++b)
            hipLaunchKernelGGL(lgamma_partial_kernel, dim3(g), dim3(256), 0, e->stream, e->X + (size_t)b * e->Np * VMAX, e->N, block_width(e, b), VMAX,
                               e->scratch + (size_t)b * g);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->scratch, g * e->NB, 1, 1, e->scal + 5);
        HIPCK(hipGetLastError());
        CK(allreduce(e, e->scal + 5, 1));
        double v;
        CK(read_scalars(e, 5, 1, &v));
        e->lgam_sum = v;
        e->lgam_valid = true;
    }
    FwdParams p;
    CK(fwd_params(e, p));
    p.wkl = nullptr;
    p.wlh = nullptr;
    for (int b = 0; b < e->NB; ++b) {  // (a sum over the features: one pass per feature block)
        FwdParams pb = p;
        pb.X = e->X + (size_t)b * e->Np * VMAX;
        pb.W = e->W + (size_t)VMAX * b;
        pb.V = block_width(e, b);
        pb.out = e->objpart + (size_t)b * e->fgrid;
        CK(launch_forward<3>(e, pb));
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->objpart, e->fgrid * e->NB, 1, 1, e->scal + 6);
    HIPCK(hipGetLastError());
    CK(allreduce(e, e->scal + 6, 1));
    double v;
    CK(read_scalars(e, 6, 1, &v));
    *out = v - e->lgam_sum;
    return 0;
}
