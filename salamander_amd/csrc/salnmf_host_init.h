// Part of salnmf.hip's translation unit (included there, inside its extern "C" block; not a stand-alone header):
// device-side initialisation entry points: Gram matrix, projection, flat, separableNMF selection.
// Split out of salnmf.hip in round 5 for readability only -- one translation unit, the same static helpers and macros.

// ------------------------------------------------------------------------------------ initialisation (row f3)

// one 96 x 96 diagonal block of X^T X (and the block's sum of X) -> host[GRAM_PART + 1], all-reduced over the shards
static int gram_diagonal_block(salnmf_engine* e, const double* Xb, std::vector<double>& host) {
    const int nparts = e->grid * WAVES;
    CK(ensure_scratch(e, (size_t)nparts * GRAM_PART + nparts + GRAM_PART + 2));
    double* part = e->scratch;
    double* xpart = part + (size_t)nparts * GRAM_PART;
    double* red = xpart + nparts;  // [GRAM_PART | 1]
    hipLaunchKernelGGL(gram_kernel, dim3(e->grid), dim3(BLOCK), 0, e->stream, Xb, e->ntiles, part, xpart);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(GRAM_PART), dim3(256), 0, e->stream, part, nparts, GRAM_PART, GRAM_PART, red);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, xpart, nparts, 1, 1, red + GRAM_PART);
    HIPCK(hipGetLastError());
    CK(allreduce(e, red, (size_t)GRAM_PART + 1));  // sample-sharded engines: the Gram matrix of ALL samples
    host.resize((size_t)GRAM_PART + 1);
    return download(e, host.data(), red, host.size());
}

int salnmf_init_gram(salnmf_engine* e, double* gram_out, double* xsum_out) {
    if (!e || !gram_out) return fail("null argument");
    CK(enter(e));
    const int V = e->V;
    std::vector<double> host;
    double xsum = 0.0;
    for (int b = 0; b < e->NB; ++b) {  // the diagonal blocks (one block: the whole matrix)
        CK(gram_diagonal_block(e, e->X + (size_t)b * e->Np * VMAX, host));
        const int o = VMAX * b;
        int idx = 0;
        for (int vt = 0; vt < VT; ++vt)
            for (int wt = vt; wt < VT; ++wt, ++idx)
                for (int r = 0; r < 4; ++r)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = o + 16 * vt + (lane >> 4) + 4 * r, j = o + 16 * wt + (lane & 15);
                        if (i < V && j < V) {
                            const double g = host[((size_t)idx * 4 + r) * 64 + lane];
                            // a diagonal tile holds both triangles; off-diagonal tiles are mirrored
                            gram_out[(size_t)i * V + j] = g;
                            if (vt != wt) gram_out[(size_t)j * V + i] = g;
                        }
                    }
        xsum += host[GRAM_PART];
    }
    // n_features > 96: the off-diagonal blocks Xa^T Xb, a < b, half a block (18 of its 36 tiles) per launch
    if (e->NB > 1) {
        const int nparts = e->grid * WAVES;
        CK(ensure_scratch(e, (size_t)nparts * GRAMX_PART + GRAMX_PART));
        double* part = e->scratch;
        double* red = part + (size_t)nparts * GRAMX_PART;
        host.resize(GRAMX_PART);
        for (int a = 0; a < e->NB; ++a)
            for (int b = a + 1; b < e->NB; ++b)
                for (int half = 0; half < 2; ++half) {
                    const double* Xa = e->X + (size_t)a * e->Np * VMAX;
                    const double* Xb = e->X + (size_t)b * e->Np * VMAX;
                    if (half == 0)
                        hipLaunchKernelGGL(gram_cross_kernel<0>, dim3(e->grid), dim3(BLOCK), 0, e->stream, Xa, Xb, e->ntiles, part);
                    else
                        hipLaunchKernelGGL(gram_cross_kernel<1>, dim3(e->grid), dim3(BLOCK), 0, e->stream, Xa, Xb, e->ntiles, part);
                    hipLaunchKernelGGL(sum_partials_kernel, dim3(GRAMX_PART), dim3(256), 0, e->stream, part, nparts, GRAMX_PART, GRAMX_PART, red);
                    HIPCK(hipGetLastError());
                    CK(allreduce(e, red, (size_t)GRAMX_PART));
                    CK(download(e, host.data(), red, host.size()));
                    for (int vt = 0; vt < 3; ++vt)
                        for (int wt = 0; wt < VT; ++wt)
                            for (int r = 0; r < 4; ++r)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int i = VMAX * a + 48 * half + 16 * vt + (lane >> 4) + 4 * r, j = VMAX * b + 16 * wt + (lane & 15);
                                    if (i < V && j < V) {
                                        const double g = host[((size_t)(VT * vt + wt) * 4 + r) * 64 + lane];
                                        gram_out[(size_t)i * V + j] = g;
                                        gram_out[(size_t)j * V + i] = g;
                                    }
                                }
                }
    }
    if (xsum_out) *xsum_out = xsum;
    return 0;
}

int salnmf_init_project(salnmf_engine* e, const double* B, double* posneg_out) {
    if (!e || !B || !posneg_out) return fail("null argument");
    CK(enter(e));
    // (more than 64 signatures: chunk by chunk -- the chunk's rows of B into the chunk's columns of H, its norms behind the
    // earlier chunks')
    const int K = e->K, V = e->V, KP = e->KP, NC = e->NC;
    const int pgrid = (int)std::min<int64_t>(1024, e->ntiles);
    CK(ensure_scratch(e, (size_t)K * V + (size_t)pgrid * 2 * KP + (size_t)NC * 2 * KP));
    double* dB = e->scratch;
    double* part = dB + (size_t)K * V;
    double* red = part + (size_t)pgrid * 2 * KP;
    CK(upload(e, dB, B, (size_t)K * V));
    e->h_pending = false;  // H is overwritten in full
    const size_t lds = ((size_t)KP * PROJ_LD + 16 * PROJ_LD + 256) * sizeof(double);
    for (int ci = 0; ci < NC; ++ci) {
        const auto& c = e->kc[(size_t)ci];
        for (int b = 0; b < e->NB; ++b)  // (feature blocks: the projection is a sum over them, accumulated in H)
            hipLaunchKernelGGL(init_project_kernel, dim3(pgrid), dim3(256), lds, e->stream, e->X + (size_t)b * e->Np * VMAX,
                               dB + (size_t)c.k0 * V + (size_t)VMAX * b, e->H + (size_t)ci * e->Np * KP, e->N, e->ntiles, block_width(e, b), V, c.K, KP, part,
                               b == 0 ? 1 : 0, b == e->NB - 1 ? 1 : 0);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(2 * KP), dim3(256), 0, e->stream, part, pgrid, 2 * KP, 2 * KP, red + (size_t)ci * 2 * KP);
        HIPCK(hipGetLastError());
    }
    CK(allreduce(e, red, (size_t)NC * 2 * KP));
    std::vector<double> host((size_t)NC * 2 * KP);
    CK(download(e, host.data(), red, host.size()));
    for (int ci = 0; ci < NC; ++ci) {
        const auto& c = e->kc[(size_t)ci];
        for (int j = 0; j < c.K; ++j) {
            posneg_out[c.k0 + j] = host[(size_t)ci * 2 * KP + j];
            posneg_out[K + c.k0 + j] = host[(size_t)ci * 2 * KP + KP + j];
        }
    }
    return 0;
}

int salnmf_init_finish(salnmf_engine* e, const double* scale, const int* take_neg, const double* post, double zero_below, double fill) {
    if (!e || !scale || !take_neg || !post) return fail("null argument");
    CK(enter(e));
    const int K = e->K;
    CK(ensure_scratch(e, (size_t)3 * K + 8));
    double* dscale = e->scratch;
    double* dpost = dscale + K;
    int* dneg = reinterpret_cast<int*>(dpost + K);
    CK(upload(e, dscale, scale, (size_t)K));
    CK(upload(e, dpost, post, (size_t)K));
    HIPCK(hipMemcpyAsync(dneg, take_neg, (size_t)K * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    for (int ci = 0; ci < e->NC; ++ci) {  // (signature chunks: the chunk's entries of the operands, its columns of H)
        const auto& c = e->kc[(size_t)ci];
        InitFinishParams p;
        p.H = e->H + (size_t)ci * e->Np * e->KP;
        p.scale = dscale + c.k0;
        p.take_neg = dneg + c.k0;
        p.post = dpost + c.k0;
        p.zero_below = zero_below;
        p.fill = fill;
        p.N = e->N;
        p.Np = e->Np;
        p.K = c.K;
        p.KP = e->KP;
        p.first_component = c.k0 == 0 ? 1 : 0;
        hipLaunchKernelGGL(init_finish_kernel, dim3(2048), dim3(256), 0, e->stream, p);
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(e->stream));  // the scratch operands may be reused by the next call
    return 0;
}

int salnmf_init_flat(salnmf_engine* e, const double* post) {
    if (!e || !post) return fail("null argument");
    CK(enter(e));
    e->h_pending = false;
    CK(ensure_scratch(e, (size_t)e->K));
    CK(upload(e, e->scratch, post, (size_t)e->K));
    for (int ci = 0; ci < e->NC; ++ci) {
        const auto& c = e->kc[(size_t)ci];
        hipLaunchKernelGGL(init_flat_kernel, dim3(1024), dim3(256), 0, e->stream, e->X, e->H + (size_t)ci * e->Np * e->KP, e->N, e->Np, c.K, e->KP,
                           e->scratch + c.k0, e->NB, e->K);
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(e->stream));  // the scratch operand may be reused by the next call
    return 0;
}

int salnmf_init_separable(salnmf_engine* e, int n_select, int64_t* chosen_out, double* norms_out) {
    if (!e || !chosen_out) return fail("null argument");
    if (n_select < 1 || (int64_t)n_select > e->N) return fail("n_select must be in [1, n_samples]");
    if (sharded(e)) return fail("the separableNMF selection needs all samples on one engine: not available on a sharded engine");
    CK(enter(e));
    const int grid = (int)std::min<int64_t>(1024, (e->N + 15) / 16);
    const bool wide = e->NB > 1;  // rows of R over all feature blocks (sep_pass_wide_kernel)
    const int ldr = e->NB * VMAX;
    const size_t nR = (size_t)e->Np * ldr, nstate = wide ? (size_t)ldr + 2 : (size_t)SEP_STATE;
    CK(ensure_scratch(e, nR + nstate + 2 * (size_t)grid + 2 * (size_t)n_select));
    double* R = e->scratch;
    double* state = R + nR;
    double* pval = state + nstate;
    long long* pidx = reinterpret_cast<long long*>(pval + grid);
    long long* chosen = pidx + grid;
    double* norms = reinterpret_cast<double*>(chosen + n_select);
    auto pass = [&](bool init) {
        if (wide) {
            if (init) hipLaunchKernelGGL(sep_pass_wide_kernel<true>, dim3(grid), dim3(SEP_BLOCK), 0, e->stream, e->X, R, e->N, e->Np, e->V, ldr, state, pval, pidx);
            else hipLaunchKernelGGL(sep_pass_wide_kernel<false>, dim3(grid), dim3(SEP_BLOCK), 0, e->stream, e->X, R, e->N, e->Np, e->V, ldr, state, pval, pidx);
        } else {
            if (init) hipLaunchKernelGGL(sep_pass_kernel<true>, dim3(grid), dim3(SEP_BLOCK), 0, e->stream, e->X, R, e->N, e->V, state, pval, pidx);
            else hipLaunchKernelGGL(sep_pass_kernel<false>, dim3(grid), dim3(SEP_BLOCK), 0, e->stream, e->X, R, e->N, e->V, state, pval, pidx);
        }
    };
    pass(true);
    HIPCK(hipGetLastError());
    for (int k = 0; k < n_select; ++k) {
        hipLaunchKernelGGL(sep_select_kernel, dim3(1), dim3(256), 0, e->stream, R, pval, pidx, grid, state, chosen, norms, k, ldr);
        HIPCK(hipGetLastError());
        if (k + 1 < n_select) {
            pass(false);
            HIPCK(hipGetLastError());
        }
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "index type");
    HIPCK(hipMemcpyAsync(chosen_out, chosen, (size_t)n_select * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
    if (norms_out) HIPCK(hipMemcpyAsync(norms_out, norms, (size_t)n_select * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    return check_abort(e);
}
