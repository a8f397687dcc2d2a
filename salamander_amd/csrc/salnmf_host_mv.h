// Part of salnmf.hip's translation unit (included there, inside its extern "C" block; not a stand-alone header):
// MvNMF entry points and their host-side logic (mvnmf.py:19-92, 149-210): classic, queued and sharded-queued steps, the wide / chunked plain form.
// Split out of salnmf.hip in round 5 for readability only -- one translation unit, the same static helpers and macros.

// ------------------------------------------------------------------------------------ MvNMF

static int mv_logdet_to_slot(salnmf_engine* e, const double* W, double delta, int slot) {
    hipLaunchKernelGGL(mv_logdet_kernel, dim3(1), dim3(MV_BLOCK), 0, e->stream, W, e->K, e->V, delta, e->scal + slot);
    HIPCK(hipGetLastError());
    return 0;
}

// ---- MvNMF on more than 96 features (feature blocks), on more than 64 signatures (signature chunks; round 5) or both: the step of mvnmf.py:197-210 in its plain form -- update_H over the blocks / chunks, the
// numerator passes, the W-only algebra, root, and a host-driven line search whose objectives are the KLNMF path's forward
// passes.  No speculation: such a problem spends its time in the passes over the samples
// (csrc/salnmf_mv_wide_kernels.h has the kernels).
static inline bool mv_wide(const salnmf_engine* e) { return split(e); }
static int mv_wide_check(const salnmf_engine*) { return 0; }  // (round 5: every split engine runs MvNMF, sample shards included)
// (signature chunks) the K x 2K scratch of the global-memory elimination
static int ensure_mv_scratch(salnmf_engine* e) {
    if (e->mvS) return 0;
    if (e->K > MVM_KMAX) return fail("MvNMF supports up to %d signatures (this engine has %d)", MVM_KMAX, e->K);
    HIPCK(hipMalloc(&e->mvS, (size_t)2 * e->K * e->K * sizeof(double)));
    return 0;
}
static int mv_wide_logdet(salnmf_engine* e, const double* W, double delta, int slot) {
    if (e->NC > 1) {
        CK(ensure_mv_scratch(e));
        hipLaunchKernelGGL(mv_many_gram_kernel<false>, dim3(e->K), dim3(256), 0, e->stream, W, e->K, e->V, delta, e->mvS);
        HIPCK(hipGetLastError());
        hipLaunchKernelGGL(mv_many_eliminate_kernel<false>, dim3(1), dim3(MVM_BLOCK), 0, e->stream, e->mvS, e->K, e->scal + slot);
        HIPCK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(mv_logdet_wide_kernel, dim3(1), dim3(MV_BLOCK), 0, e->stream, W, e->K, e->V, delta, e->scal + slot);
    HIPCK(hipGetLastError());
    return 0;
}
// update_H of an MvNMF step (MvNMF._update_H, mvnmf.py:162-165: in place, unweighted) on a split engine
static int mv_wide_update_H(salnmf_engine* e) {
    if (grid_split(e)) {
        CK(flush_H_scale(e));
        return grid_passes(e, false, true, 0, e->H, false);
    }
    if (e->NC > 1) {
        CK(flush_H_scale(e));
        CK(chunk_ratio(e));
        return chunk_passes(e, false, true, 0, 0, false);
    }
    CK(blocked_update_H(e, e->H, kEps, false));
    e->h_pending = false;
    return 0;
}
// out[0 .. Kc) = sum over the samples of H[n][k] for one [Np][KP] block of H: partial sums of contiguous row ranges read in
// whole rows (colsum_partial_kernel), then the partials in order -- the one-workgroup-per-column kernel of the narrow path reads
// a column with a stride of KP doubles, 156 us per chunk at 10^5 samples
static int rowsums_H_block(salnmf_engine* e, const double* H, int Kc, double* out) {
    const int pgrid = (int)std::min<int64_t>(512, (e->N + 255) / 256);
    CK(ensure_scratch(e, (size_t)512 * 64));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(pgrid), dim3(256), 0, e->stream, H, e->N, e->KP, e->scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(Kc), dim3(256), 0, e->stream, e->scratch, pgrid, e->KP, Kc, out);
    HIPCK(hipGetLastError());
    return 0;
}
// numerator of (W, H) -> Gblk (feature blocks) / red (signature chunks), rowsums_H -> red + K V, A = W Y_minus, B = W |Y|
// -> mvA, mvB, log det(W) -> scal[3]
static int mv_wide_prepare(salnmf_engine* e, double delta) {
    CK(flush_H_scale(e));  // (the column sums below read H as it is)
    if (e->NC > 1) {
        CK(ensure_mv_scratch(e));
        if (grid_split(e)) {
            CK(grid_passes(e, true, false, 0, nullptr, false));  // every (chunk, block) pair's numerator -> Gblk
        } else {
            CK(chunk_ratio(e));
            CK(chunk_passes(e, true, false, 0, 0, false, true));  // every row's numerator, W untouched
        }
        for (int ci = 0; ci < e->NC; ++ci) {
            const auto& c = e->kc[(size_t)ci];
            CK(rowsums_H_block(e, e->H + (size_t)ci * e->Np * e->KP, c.K, e->red + (size_t)e->K * e->V + c.k0));
        }
        if (sharded(e)) CK(allreduce(e, e->red + (size_t)e->K * e->V, (size_t)e->K));  // rowsums_H over all shards
        hipLaunchKernelGGL(mv_many_gram_kernel<true>, dim3(e->K), dim3(256), 0, e->stream, e->W, e->K, e->V, delta, e->mvS);
        HIPCK(hipGetLastError());
        hipLaunchKernelGGL(mv_many_eliminate_kernel<true>, dim3(1), dim3(MVM_BLOCK), 0, e->stream, e->mvS, e->K, e->scal + 3);
        HIPCK(hipGetLastError());
        hipLaunchKernelGGL(mv_many_AB_kernel, dim3(e->K), dim3(128), 0, e->stream, e->mvS, e->W, e->K, e->V, e->mvA, e->mvB);
        HIPCK(hipGetLastError());
        return 0;
    }
    CK(blocked_numerators(e, false));  // update_W_unconstrained takes no weights (mvnmf.py:37-66): as the narrow path
    CK(rowsums_H_block(e, e->H, e->K, e->red + (size_t)e->K * e->V));
    if (sharded(e)) CK(allreduce(e, e->red + (size_t)e->K * e->V, (size_t)e->K));  // rowsums_H over all shards
    hipLaunchKernelGGL(mv_prepare_W_wide_kernel, dim3(1), dim3(MV_BLOCK), 0, e->stream, e->W, e->K, e->V, delta, e->mvA, e->mvB, e->scal + 3);
    HIPCK(hipGetLastError());
    return 0;
}
static int mv_wide_root(salnmf_engine* e, double lam, int n_given) {
    hipLaunchKernelGGL(mv_trial_row_wide_kernel<true>, dim3(e->K), dim3(256), 0, e->stream, e->W, e->Wunc, 1.0, 0, e->K, e->V, e->Wtrial, e->cs, e->mvA,
                       e->mvB, (e->NC > 1 && e->NB == 1) ? e->red : e->Gblk, e->red + (size_t)e->K * e->V, lam, n_given);
    HIPCK(hipGetLastError());
    return 0;
}
// line_search (mvnmf.py:69-92) from the resident (W, H) with W_unconstrained in Wunc.  trial_ready: the first trial (the
// normalised, clipped W_unconstrained) and its column sums are in Wtrial / cs already; have_logdet: scal[3] = log det(W).
static int mv_wide_line_search(salnmf_engine* e, double lam, double delta, double* gamma, bool trial_ready, bool have_logdet, double* f_accepted) {
    CK(flush_H_scale(e));
    CK(objective_to_slot(e, e->W, nullptr, false, 0));  // KL(X || W H), unweighted (mvnmf.py:27-34)
    if (!have_logdet) CK(mv_wide_logdet(e, e->W, delta, 3));
    double g = *gamma;
    bool blend = false;
    for (;;) {
        if (blend || !trial_ready) {
            hipLaunchKernelGGL(mv_trial_row_wide_kernel<false>, dim3(e->K), dim3(256), 0, e->stream, e->W, e->Wunc, g, blend ? 1 : 0, e->K, e->V, e->Wtrial,
                               e->cs, nullptr, nullptr, nullptr, nullptr, 0.0, 0);
            HIPCK(hipGetLastError());
        }
        CK(mv_wide_logdet(e, e->Wtrial, delta, 4));
        CK(objective_to_slot(e, e->Wtrial, e->cs, false, 2));  // KL(X || W_trial clip(H * colsum))
        double v[5];
        CK(read_scalars(e, 0, 5, v));
        const double f0 = v[0] + lam * v[3], f1 = v[2] + lam * v[4];
        if (f_accepted) *f_accepted = f1;
        if (f1 > f0 && g > 1e-16) {  // mvnmf.py:84
            g *= 0.8;
            blend = true;
            continue;
        }
        break;
    }
    *gamma = std::min(1.0, 1.2 * g);  // mvnmf.py:91
    std::swap(e->W, e->Wtrial);
    e->h_pending = true;  // H <- clip(H * colsum), applied by the readers until the next update_H pass writes H in full
    if (e->NC > 1) CK(flush_H_scale(e));  // (the chunked passes read H as it is: the rescale is a pass of its own there)
    return 0;
}
static int mv_wide_update_W(salnmf_engine* e, int n_given, double lam, double delta, double* gamma, double* f_accepted) {
    if (n_given >= e->K) return 0;
    CK(mv_wide_prepare(e, delta));
    CK(mv_wide_root(e, lam, n_given));
    return mv_wide_line_search(e, lam, delta, gamma, true, true, f_accepted);
}

int salnmf_mv_objective(salnmf_engine* e, double lam, double delta, double* out) {
    if (e && split(e)) CK(mv_wide_check(e));
    if (!e || !out) return fail("null argument");
    CK(enter(e));
    CK(objective_to_slot(e, e->W, nullptr, false, 0));
    if (mv_wide(e))
        CK(mv_wide_logdet(e, e->W, delta, 3));
    else
        CK(mv_logdet_to_slot(e, e->W, delta, 3));
    double v[4];
    CK(read_scalars(e, 0, 4, v));
    *out = v[0] + lam * v[3];
    return 0;
}

// Launch with `ev` bound to the kernel's own completion signal: a hipEventRecord behind the launch would put a
// barrier packet of its own into the queue, which costs the stream ~7 us per record (profiles/r02/mv_timeline_*.txt).
#define LAUNCH_WITH_EVENT(kernel, grid, block, stream, ev, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, nullptr, ev, 0, __VA_ARGS__)

// MvNMF._update_W (mvnmf.py:190-195) on the current (W, H).
// Two streams: everything that depends on W alone -- Gram matrix, Cholesky, inverse, A = W Y_minus, B = W |Y|,
// log det, and later the log det of a trial W -- is single-workgroup latency-bound work and runs on stream2
// while the passes over the samples (which leave one CU free, mv_grid / mv_fgrid) run on the main stream.
//   w_ready: the caller recorded evW on the main stream after the last write of W and already started
//            mv_prepare_W on stream2 (mv_step does, so that it also overlaps the update_H pass)
// The MvNMF side streams exist only in engines that run MvNMF steps (HIP multiplexes a process's streams onto a few
// hardware queues: an engine that only ever runs KL or CorrNMF steps should not hold three of them).
static int ensure_side_streams(salnmf_engine* e) {
    if (!e->stream2) HIPCK(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    if (!e->stream3) HIPCK(hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking));
    return 0;
}

// An MvNMF update_H pass whose last workgroup runs the W-only algebra (fused_kernel<!G, U, STATS>: sideW): `total`
// workgroups are launched, `total - 1` of them process tiles.  One per CU at most, so that the side workgroup starts at once.
static inline int mv_side_total(const salnmf_engine* e) { return std::min(e->cus, e->grid + 1); }

static void mv_side_params(salnmf_engine* e, FusedParams& p, const double* W, double delta) {
    p.sideW = W;
    p.sideDelta = delta;
    p.sideA = e->mvA;
    p.sideB = e->mvB;
    p.sideLogdet = e->scal + 3;
}

static int mv_start_prepare_W(salnmf_engine* e, double delta, bool record_w_event) {
    CK(ensure_side_streams(e));
    if (record_w_event) HIPCK(hipEventRecord(e->evW, e->stream));  // else: recorded when W was last written
    HIPCK(hipStreamWaitEvent(e->stream2, e->evW, 0));
    LAUNCH_WITH_EVENT(mv_prepare_W_kernel, dim3(1), dim3(MV_BLOCK), e->stream2, e->evPrepW, e->W, e->K, e->V, delta, e->mvA, e->mvB,
                      e->scal + 3);
    HIPCK(hipGetLastError());
    return 0;
}

// The numerator pass of one MvNMF W update on (W, H): G = (X/(WH)) @ H.T partials and the KL partial, reduced
// together with the row sums of H (from the preceding update_H pass) by one tail launch and all-reduced.
//   grid: e->mv_grid leaves one CU per XCD to the side stream's kernels; the steady state of mv_step has nothing on the
//   side stream and uses the whole chip (e->grid)
//   hsum_parts: workgroups of the preceding update_H pass (rows of Hsumpart)
static int mv_numerator_pass(salnmf_engine* e, const double* W, const double* H, const double* hscale, int grid, int hsum_parts) {
    CK(ensure_xlogx(e));
    FusedParams p = fused_params(e);
    p.wkl = nullptr;  // the MvNMF path is unweighted (mvnmf.py:56)
    p.wlh = nullptr;
    p.W = W;
    p.H = const_cast<double*>(H);
    p.Hout = const_cast<double*>(H);
    p.hscale = hscale;
    CK((launch_fused<true, false, true>(e, p, grid)));
    e->mv_slabs = grid;        // what the tail that follows (now or after the line-search decision) has to reduce
    e->mv_hparts = hsum_parts;
    return 0;
}

// The tail of a numerator pass: G slabs, row sums of H and KL partials reduced in one launch (-> e->red).  with_root: the
// same launch also evaluates the closed-form root and the first trial of the line search (tail_kernel: rootA; needs the
// row sums from the preceding update_H pass and an unsharded engine); ev: bound to the launch's completion.
static int mv_tail(salnmf_engine* e, bool with_root, double lam, int n_given, hipEvent_t ev) {
    TailParams t = tail_params(e, e->mv_slabs, e->red, n_given, 0, 0, true, e->mv_hparts);
    if (with_root) {
        t.rootA = e->mvA;
        t.rootB = e->mvB;
        t.rootLogdet = e->scal + 3;
        t.rootF0 = e->scal + 1;
        t.rootWunc = e->Wunc;
        t.rootWtrial = e->Wtrial;
        t.rootCs = e->cs;
        t.rootLam = lam;
    }
    if (ev)
        hipExtLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, nullptr, ev, 0, t);
    else
        hipLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, t);
    HIPCK(hipGetLastError());
    return 0;
}

//   speculate: the caller will run another step right after this one.  Everything the NEXT step can do before this
//            step's line search is decided is queued behind the first trial, assuming it will be accepted (the common
//            case):
//              * that step's update_H pass with the trial as W, reading H as clip(H * colsum) on the fly and writing a
//                second H buffer.  Its P = H' W_trial is exactly what KL(W_trial, H') -- the trial's objective --
//                needs, so the pass evaluates the trial as well (KL partials from P before the division) and the
//                separate forward pass over the samples is not run at all;
//              * its W-only algebra on stream2, and its numerator pass + tail on the main stream.
//            The scalars come back on a side stream while the numerator pass runs; on acceptance the buffers are
//            swapped and *speculated = true tells the caller that the next step starts at its closed-form root, otherwise
//            everything speculative is dropped and the backtracking loop evaluates its trials with the forward kernel.
//   w_ready:  A, B and the log det of the current W are produced on the MAIN stream already (the side workgroup of the
//            preceding update_H pass); otherwise mv_prepare_W_kernel is started on stream2 here and waited for
//   g_ready:  (in) the numerator pass of THIS step was queued by the previous call's speculation (its tail was not)
static int mv_update_W_impl(salnmf_engine* e, int n_given, double lam, double delta, double* gamma, bool have_hsum, bool w_ready,
                            bool speculate = false, bool* speculated = nullptr, bool g_ready = false, double* f_accepted = nullptr,
                            double* wunc_out = nullptr, const double* wunc_given = nullptr) {
    // wunc_out (salnmf_mv_update_W_unconstrained): stop behind the closed-form root and hand W_unconstrained back; the
    // resident state is not changed.  wunc_given (salnmf_mv_line_search): the line search of mvnmf.py:69-92 from the
    // resident (W, H) with the caller's W_unconstrained instead of the root.
    // f_accepted: the line search's value at the accepted point (mvnmf.py:82,89), which IS the model's objective of the
    // state this call leaves behind (kl_divergence_penalized of the normalised W and the rescaled H, mvnmf.py:27-34,149-156)
    if (speculated) *speculated = false;
    if (n_given >= e->K) return 0;
    CK(ensure_side_streams(e));
    const int K = e->K, V = e->V;
    if (wunc_given) {
        CK(flush_H_scale(e));
        HIPCK(hipMemcpyAsync(e->Wunc, wunc_given, (size_t)K * V * sizeof(double), hipMemcpyHostToDevice, e->stream));
        // f0 = KL(X || W H) + lam log det(W W^T + delta I) -> scal[1]  (mvnmf.py:79)
        CK(objective_to_slot(e, e->W, nullptr, false, 0));
        CK(mv_logdet_to_slot(e, e->W, delta, 3));
        hipLaunchKernelGGL(combine_scalar_kernel, dim3(1), dim3(1), 0, e->stream, e->scal + 1, (const double*)e->scal, lam, (const double*)(e->scal + 3));
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(e->stream));  // (the caller's array is free again)
    }
    if (!g_ready && !wunc_given) {
        CK(flush_H_scale(e));  // a stand-alone call after an earlier step; inside mv_step the update_H pass consumed it
        if (!w_ready) CK(mv_start_prepare_W(e, delta, true));
        // the rowsums_H partials come from the preceding update_H pass (have_hsum) or, for a stand-alone _update_W,
        // from a column-sum kernel over the current H.  Inside mv_step (w_ready: the W-only algebra ran in that pass's
        // side workgroup, on this stream) the numerator pass has the whole chip, as the speculative one has: the same
        // slab order, hence the same bits of W, whether the steps come in one call or one by one; a stand-alone call
        // leaves one CU per XCD to the W-only kernel on stream2.
        if (w_ready)  // (on as many tile workgroups as a pass with the side workgroup: one slab order for every form of the step)
            CK(mv_numerator_pass(e, e->W, e->H, nullptr, mv_side_total(e) - 1, mv_side_total(e) - 1));
        else
            CK(mv_numerator_pass(e, e->W, e->H, nullptr, e->mv_grid, e->mv_grid));
    }
    // The tail of the numerator pass -- which the previous call's speculation queued WITHOUT its tail (g_ready), so that
    // the tail runs after the line-search decision and can carry the closed-form root and the first trial of THIS step
    // in the same launch (inside mv_step on an unsharded engine: one kernel and one boundary less per step).  A, B and
    // the log det it reads come from the side workgroup of the preceding update_H pass (same stream), or from stream2.
    const bool root_in_tail = have_hsum && !sharded(e) && !wunc_given;
    if (!w_ready && !wunc_given) HIPCK(hipStreamWaitEvent(e->stream, e->evPrepW, 0));
    // (a non-speculative first trial is followed by the log det of the trial on stream2: the tail's completion is its event)
    if (!wunc_given) CK(mv_tail(e, root_in_tail, lam, n_given, root_in_tail && !speculate ? e->evTrial : nullptr));
    if (!root_in_tail && !wunc_given) {
        // the rowsums_H partials came from the preceding update_H pass (have_hsum) or, for a stand-alone _update_W,
        // come from a column-sum kernel over the current H
        if (!have_hsum) {
            hipLaunchKernelGGL(colsum_kernel, dim3(K), dim3(256), 0, e->stream, e->H, e->N, e->KP, e->red + K * V);
            HIPCK(hipGetLastError());
        }
        CK(allreduce(e, e->red, (size_t)K * V + K + 1));
    }
    // otherwise: W_unconstrained from A, B and the reduced sums, f0 = KL + lam * logdet(W) -> scal[1], inside the first
    // trial kernel below
    const MvRootParams root{e->mvA, e->mvB, e->red, e->red + K * V, e->red + K * V + K, e->scal + 3, e->scal + 1, lam, n_given};
    double g = *gamma;
    bool blend = false;
    for (;;) {
        const bool spec = speculate && !blend;
        // trial W: normalise + clip and the column sums for H on the main stream
        if (!blend && root_in_tail)
            ;  // (done by the tail launch above)
        else if (spec)
            hipLaunchKernelGGL(mv_trial_light_kernel<true>, dim3(1), dim3(MV_BLOCK), 0, e->stream, e->W, e->Wunc, 1.0, 0, K, V, e->Wtrial, e->cs, root);
        else if (!blend && wunc_given)  // the first trial is the caller's W_unconstrained, normalised and clipped (mvnmf.py:80-81)
            LAUNCH_WITH_EVENT(mv_trial_light_kernel<false>, dim3(1), dim3(MV_BLOCK), e->stream, e->evTrial, e->W, e->Wunc, 1.0, 0, K,
                              V, e->Wtrial, e->cs, root);
        else if (!blend)
            LAUNCH_WITH_EVENT(mv_trial_light_kernel<true>, dim3(1), dim3(MV_BLOCK), e->stream, e->evTrial, e->W, e->Wunc, 1.0, 0, K,
                              V, e->Wtrial, e->cs, root);
        else
            LAUNCH_WITH_EVENT(mv_trial_light_kernel<false>, dim3(1), dim3(MV_BLOCK), e->stream, e->evTrial, e->W, e->Wunc, g, 1, K, V,
                              e->Wtrial, e->cs, root);
        HIPCK(hipGetLastError());
        if (wunc_out) return download(e, wunc_out, e->Wunc, (size_t)K * V);  // (W, H untouched; the trial buffers are scratch)
        double v[5];
        if (spec) {
            CK(ensure_halt(e));
            if (!e->KLpart2) HIPCK(hipMalloc(&e->KLpart2, (size_t)e->grid * sizeof(double)));
            // The next step's update_H (+ row sums of the new H), which also evaluates this trial: KL(W_trial, H') -> scal[2],
            // summed inside the launch by the workgroup that finishes last.  The grid's last workgroup does no tiles: it
            // runs the next step's W-only algebra on the trial (A, B; its log det -> scal[3] is this trial's log det as
            // well) -- nothing of the steady state is left on a second stream, so nothing here waits for another queue.
            const int total = mv_side_total(e), nwg = total - 1;
            FusedParams sp = fused_params(e);
            sp.wkl = nullptr;
            sp.wlh = nullptr;
            sp.W = e->Wtrial;
            sp.hscale = e->cs;
            sp.Hout = e->Halt;
            sp.KLpart = e->KLpart2;
            mv_side_params(e, sp, e->Wtrial, delta);
            if (!sharded(e)) {
                sp.kl_out = e->scal + 2;
                sp.kl_counter = e->klcnt;
                CK((launch_fused<false, true, true>(e, sp, total, nullptr, e->evObj)));  // evObj = the pass's own completion signal
            } else {
                CK((launch_fused<false, true, true>(e, sp, total)));
                hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->KLpart2, nwg, 1, 1, e->scal + 2, nullptr);
                HIPCK(hipGetLastError());
                CK(allreduce(e, e->scal + 2, 1));
                HIPCK(hipEventRecord(e->evObj, e->stream));
            }
            // ... and its numerator pass (W_trial as W, the new H) on the whole chip: it runs while the scalars travel to the
            // host.  Its tail waits for the decision (see above).
            CK(mv_numerator_pass(e, e->Wtrial, e->Halt, nullptr, nwg, nwg));
            HIPCK(hipStreamWaitEvent(e->stream3, e->evObj, 0));
            HIPCK(hipMemcpyAsync(e->hpin, e->scal, 5 * sizeof(double), hipMemcpyDeviceToHost, e->stream3));
            HIPCK(hipStreamSynchronize(e->stream3));
            for (int i = 0; i < 5; ++i) v[i] = e->hpin[i];
            v[4] = v[3];  // the trial's log det came from the side workgroup
        } else {
            // the trial's log det -> scal[4] on stream2, beside KL(W_trial, clip(H * colsum)) by the forward pass
            HIPCK(hipStreamWaitEvent(e->stream2, e->evTrial, 0));
            LAUNCH_WITH_EVENT(mv_logdet_kernel, dim3(1), dim3(MV_BLOCK), e->stream2, e->evLogdet, e->Wtrial, K, V, delta, e->scal + 4);
            HIPCK(hipGetLastError());
            CK(objective_to_slot(e, e->Wtrial, e->cs, false, 2, e->mv_fgrid));
            HIPCK(hipStreamWaitEvent(e->stream, e->evLogdet, 0));
            CK(read_scalars(e, 0, 5, v));
        }
        const double f0 = v[1], f1 = v[2] + lam * v[4];
        if (f_accepted) *f_accepted = f1;  // (of the last trial: the accepted one when the loop ends)
        if (f1 > f0 && g > 1e-16) {  // mvnmf.py:84
            // (a rejected speculation is simply dropped: it wrote scratch buffers only -- Halt, the numerator slabs, A, B
            // and scal[3], which the next non-speculative step recomputes for the W it starts from -- and everything that
            // read the trial buffer ran on this stream, ahead of the blend that overwrites it)
            g *= 0.8;
            blend = true;
            continue;
        }
        if (spec) {
            // accepted at the first trial: the speculative passes are the first half of the next step
            *gamma = std::min(1.0, 1.2 * g);
            std::swap(e->W, e->Wtrial);
            std::swap(e->H, e->Halt);  // written in full from clip(H * cs): nothing pending
            // (no evW: the next step's W-only algebra ran already; a later stand-alone start records its own)
            e->h_pending = false;
            if (speculated) *speculated = true;
            return 0;
        }
        break;
    }
    *gamma = std::min(1.0, 1.2 * g);  // mvnmf.py:91
    // accept: W <- W_trial, H <- clip(H * colsum).  The rescale of H is not a pass of its own: every reader of H
    // applies clip(H * cs) on the fly until the next update_H pass writes H in full (flush_H_scale otherwise)
    std::swap(e->W, e->Wtrial);  // no copy: the trial buffer becomes W
    HIPCK(hipEventRecord(e->evW, e->stream));  // W is final for the next step's W-only kernels
    e->h_pending = true;
    return 0;
}

// ---- MvNMF steps queued ahead of the host (unsharded engines, at least one free signature).
// Per step TWO launches: the tail of the previous numerator half (reduce, closed-form root, first trial, f0) and the MVJ
// pass (fused_kernel<.., MVJ>: update_H with the trial -- which evaluates the trial -- and the numerator half on the new H,
// the next step's W-only algebra in its last workgroup).  The line-search decision of step i is taken ON THE DEVICE, in the
// prologue of step i + 1's tail (TailParams::dec_*): accepted -> go on; rejected -> the flag is set and everything queued
// behind returns at once.  The host queues a whole call's steps with the buffer roles alternating as if every first trial
// were accepted (the common case), reads the flag and the scalars once at the end, and resolves a rejected step on the
// classic path (blends evaluated by the forward kernel, mvnmf.py:84-90), then queues the rest.  The decision compares the
// same doubles with the same operations as the host's, so the result is the classic form's bit for bit.
static inline int mv_f0_slot(int step) { return (step & 1) ? 9 : 1; }  // f0 by step parity: a tail reads the previous step's while it writes its own

// the backtracking part of line_search (mvnmf.py:84-90) from the resident (W, H), W_unconstrained in Wunc and f0 given;
// g: the gamma the rejected first trial left (already multiplied by 0.8).  Accepts: W <- the blend's trial, H pending.
static int mv_backtrack(salnmf_engine* e, double lam, double delta, double f0, double* g_io, double* f_accepted) {
    const int K = e->K, V = e->V;
    const MvRootParams root{e->mvA, e->mvB, e->red, e->red + K * V, e->red + K * V + K, e->scal + 3, e->scal + 1, lam, 0};
    double g = *g_io;
    for (;;) {
        LAUNCH_WITH_EVENT(mv_trial_light_kernel<false>, dim3(1), dim3(MV_BLOCK), e->stream, e->evTrial, e->W, e->Wunc, g, 1, K, V, e->Wtrial, e->cs, root);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamWaitEvent(e->stream2, e->evTrial, 0));
        LAUNCH_WITH_EVENT(mv_logdet_kernel, dim3(1), dim3(MV_BLOCK), e->stream2, e->evLogdet, e->Wtrial, K, V, delta, e->scal + 4);
        HIPCK(hipGetLastError());
        CK(objective_to_slot(e, e->Wtrial, e->cs, false, 2, e->mv_fgrid));
        HIPCK(hipStreamWaitEvent(e->stream, e->evLogdet, 0));
        double v[5];
        CK(read_scalars(e, 0, 5, v));
        const double f1 = v[2] + lam * v[4];
        if (f_accepted) *f_accepted = f1;
        if (f1 > f0 && g > 1e-16) {
            g *= 0.8;
            continue;
        }
        break;
    }
    *g_io = g;
    std::swap(e->W, e->Wtrial);
    HIPCK(hipEventRecord(e->evW, e->stream));
    e->h_pending = true;
    return 0;
}

static int mv_steps_queued(salnmf_engine* e, int n_steps, int n_given, double lam, double delta, double* gamma, double* f_out, bool more_follows,
                           bool resume) {
    CK(ensure_side_streams(e));
    CK(ensure_halt(e));
    if (!e->KLpart2) HIPCK(hipMalloc(&e->KLpart2, (size_t)e->grid * sizeof(double)));
    if (!e->mvflag) HIPCK(hipMalloc(&e->mvflag, 16));
    const int total = mv_side_total(e), nwg = total - 1;
    // A sample-sharded engine (round 5) runs the same queue with the sums over the samples all-reduced: per step the tail
    // (local reduction only), ONE exchange of [G | rowsums_H | KL of the numerator half | KL of the previous step's trial]
    // (K V + K + 2 doubles: the peer-to-peer kernel, or RCCL), the root / first trial / f0 kernel -- which takes the
    // device-side decision from the all-reduced sums, the same bits on every rank -- and the MVJ pass.  Every rank queues
    // the same launches and exchanges whatever the flag says, so the exchanges stay in step.
    const bool sh = sharded(e);
    const int K = e->K, V = e->V;
    const size_t nred = (size_t)K * V + K + 2;
    double* const trial_kl = sh ? e->red + (size_t)K * V + K + 1 : e->scal + 2;  // where an MVJ pass leaves its trial's KL
    double g = *gamma, f_last = 0.0;
    bool ahead = resume;
    int done = 0;
    while (done < n_steps) {
        HIPCK(hipMemsetAsync(e->mvflag, 0, sizeof(unsigned), e->stream));
        if (!ahead) {
            // the first half of step `done`: update_H (+ the W-only algebra of W in the side workgroup), numerator pass
            FusedParams p = fused_params(e);
            p.wkl = nullptr;  // MvNMF._update_H passes no weights (mvnmf.py:162-165)
            p.wlh = nullptr;
            p.KLpart = nullptr;
            mv_side_params(e, p, e->W, delta);
            CK((launch_fused<false, true, true>(e, p, total)));
            e->h_pending = false;
            CK(mv_numerator_pass(e, e->W, e->H, nullptr, nwg, nwg));
        }
        // queue: tail_i (decides step i - 1, root + trial of step i), MVJ pass_i (evaluates trial i, first half of step i + 1)
        const int first = done;
        int queued = 0;       // steps whose MVJ pass is queued (their trial's decision is pending or on the device)
        bool classic_last = false;
        for (int i = first; i < n_steps; ++i) {
            const bool spec = (i + 1 < n_steps) || more_follows;
            TailParams t = tail_params(e, e->mv_slabs, e->red, n_given, 0, 0, true, e->mv_hparts);
            t.rootA = e->mvA;
            t.rootB = e->mvB;
            t.rootLogdet = e->scal + 3;
            t.rootF0 = e->scal + mv_f0_slot(i);
            t.rootWunc = e->Wunc;
            t.rootWtrial = e->Wtrial;
            t.rootCs = e->cs;
            t.rootLam = lam;
            t.mv_flag = e->mvflag;
            if (sh) {
                // local sums only; the root, the trial and the decision follow the exchange
                t.rootA = nullptr;
                hipLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, t);
                HIPCK(hipGetLastError());
                CK(allreduce(e, e->red, nred));
                MvRootParams root{e->mvA, e->mvB, e->red, e->red + K * V, e->red + K * V + K, e->scal + 3, e->scal + mv_f0_slot(i), lam, n_given};
                root.mv_flag = e->mvflag;
                if (i > first) {
                    root.dec_f0 = e->scal + mv_f0_slot(i - 1);
                    root.dec_kl = trial_kl;
                    root.dec_logdet = e->scal + 3;
                    root.dec_lam = lam;
                    root.dec_code = (unsigned)(i - first);
                }
                if (!spec)
                    LAUNCH_WITH_EVENT(mv_trial_light_kernel<true>, dim3(1), dim3(MV_BLOCK), e->stream, e->evTrial, e->W, e->Wunc, 1.0, 0, K, V, e->Wtrial,
                                      e->cs, root);
                else
                    hipLaunchKernelGGL(mv_trial_light_kernel<true>, dim3(1), dim3(MV_BLOCK), 0, e->stream, e->W, e->Wunc, 1.0, 0, K, V, e->Wtrial, e->cs, root);
                HIPCK(hipGetLastError());
            } else {
            if (i > first) {  // (gamma of a step behind accepted first trials is >= the call's gamma: the `gamma > 1e-16` half of mvnmf.py:84 holds)
                t.dec_f0 = e->scal + mv_f0_slot(i - 1);
                t.dec_kl = e->scal + 2;
                t.dec_logdet = e->scal + 3;
                t.dec_lam = lam;
                t.dec_code = (unsigned)(i - first);  // 1 + index within this batch of the step whose trial is rejected
            }
            if (!spec)
                hipExtLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, nullptr, e->evTrial, 0, t);
            else
                hipLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, t);
            HIPCK(hipGetLastError());
            }
            if (!spec) {
                classic_last = true;
                break;
            }
            FusedParams sp = fused_params(e);
            sp.wkl = nullptr;
            sp.wlh = nullptr;
            sp.W = e->Wtrial;
            sp.hscale = e->cs;
            sp.Hout = e->Halt;
            sp.KLpart = e->KLpart2;  // the update_H half's partials: the trial's KL, summed inside the launch -> scal[2]
            sp.KLpartB = e->KLpart;  // the numerator half's: f0 of the next step, reduced by its tail
            sp.kl_out = trial_kl;  // (sharded: this rank's share, all-reduced with the next tail's sums)
            sp.kl_counter = e->klcnt;
            sp.skip_flag = e->mvflag;
            mv_side_params(e, sp, e->Wtrial, delta);
            const FusedSel sel{e->KS, e->KTM, e->KR, true, true, true, false, false, false, false, true};
            if (launch_fused_inst(sel, sp, total, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", e->KS, e->KTM, e->KR);
            HIPCK(hipGetLastError());
            e->mv_slabs = nwg;
            e->mv_hparts = nwg;
            std::swap(e->W, e->Wtrial);  // as if accepted (undone below if it was not)
            std::swap(e->H, e->Halt);
            e->h_pending = false;
            ++queued;
        }
        if (sh && queued > 0 && !classic_last) {
            // the last queued trial's KL has not been through an exchange yet (the next step's would have carried it): one
            // scalar all-reduce, into the slot the host reads (after a rejection further up it is a stale value nobody uses)
            HIPCK(hipMemcpyAsync(e->scal + 2, trial_kl, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
            CK(allreduce(e, e->scal + 2, 1));
        }
        // one read for the whole batch: the flag and the scalars
        unsigned code = 0;
        double v[16];
        HIPCK(hipMemcpyAsync(e->hpin, e->scal, 16 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipMemcpyAsync(reinterpret_cast<char*>(e->hpin) + 16 * sizeof(double), e->mvflag, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipStreamSynchronize(e->stream));
        CK(check_abort(e));
        for (int i = 0; i < 16; ++i) v[i] = e->hpin[i];
        memcpy(&code, reinterpret_cast<char*>(e->hpin) + 16 * sizeof(double), sizeof code);
        // which queued step (if any) is open: rejected on the device, or the last queued one (nobody has compared its f1
        // with its f0 yet).  The steps before it were accepted on the device.
        int open = -1;
        if (code != 0) open = first + (int)code - 1;
        else if (queued > 0 && !classic_last) open = first + queued - 1;  // (a classic last step's tail decided the last queued one)
        for (int i = first; i < (open >= 0 ? open : first + queued); ++i) g = std::min(1.0, 1.2 * g);
        if (open >= 0) {
            // the open step's MVJ pass was the last one to write scal[2], scal[3] (everything behind a rejection returned at once)
            const double f0 = v[mv_f0_slot(open)], f1 = v[2] + lam * v[3];
            const bool rejected = (code != 0 || f1 > f0) && g > 1e-16;  // mvnmf.py:84
            // the pointer swaps of the steps behind the open one (and, if rejected, its own) are undone
            const int undo = first + queued - open - (rejected ? 0 : 1);
            if (undo & 1) {
                std::swap(e->W, e->Wtrial);
                std::swap(e->H, e->Halt);
            }
            if (rejected) {
                // (W, H) = the state the open step started its line search from: H unscaled with its scale in cs, W_unconstrained
                // in Wunc; the numerator slabs, A, B, log det and the second H buffer hold the dropped speculation
                g *= 0.8;
                CK(mv_backtrack(e, lam, delta, f0, &g, &f_last));
                ahead = false;
            } else {
                f_last = f1;
                ahead = true;  // its MVJ pass is the first half of the next step
            }
            g = std::min(1.0, 1.2 * g);
            done = open + 1;
            if (done < n_steps || !classic_last) continue;
        }
        if (classic_last) {
            // the call's last step without a continuation: its trial is evaluated by the forward kernel (no pass follows)
            const int i = n_steps - 1;
            HIPCK(hipStreamWaitEvent(e->stream2, e->evTrial, 0));
            LAUNCH_WITH_EVENT(mv_logdet_kernel, dim3(1), dim3(MV_BLOCK), e->stream2, e->evLogdet, e->Wtrial, e->K, e->V, delta, e->scal + 4);
            HIPCK(hipGetLastError());
            CK(objective_to_slot(e, e->Wtrial, e->cs, false, 2, e->mv_fgrid));
            HIPCK(hipStreamWaitEvent(e->stream, e->evLogdet, 0));
            double w[16];
            CK(read_scalars(e, 0, 16, w));
            const double f0 = w[mv_f0_slot(i)], f1 = w[2] + lam * w[4];
            f_last = f1;
            if (f1 > f0 && g > 1e-16) {
                g *= 0.8;
                CK(mv_backtrack(e, lam, delta, f0, &g, &f_last));
            } else {
                std::swap(e->W, e->Wtrial);
                HIPCK(hipEventRecord(e->evW, e->stream));
                e->h_pending = true;
            }
            g = std::min(1.0, 1.2 * g);
            done = n_steps;
            ahead = false;
        }
    }
    *gamma = g;
    if (f_out) *f_out = f_last;
    if (ahead) {  // (only with more_follows)
        e->mv_ahead = true;
        e->mv_ahead_delta = delta;
        e->mv_ahead_given = n_given;
    }
    return 0;
}

int salnmf_mv_update_W(salnmf_engine* e, int n_given, double lam, double delta, double* gamma_inout) {
    if (e && split(e)) CK(mv_wide_check(e));
    if (!e || !gamma_inout) return fail("null argument");
    e->keep_valid = false;  // (the MvNMF steps use the second H buffer themselves)
    CK(enter(e));
    if (mv_wide(e)) return mv_wide_update_W(e, n_given, lam, delta, gamma_inout, nullptr);
    return mv_update_W_impl(e, n_given, lam, delta, gamma_inout, false, false);
}

int salnmf_mv_logdet(salnmf_engine* e, double delta, double* out) {
    if (e && split(e)) CK(mv_wide_check(e));
    if (!e || !out) return fail("null argument");
    CK(enter(e));
    if (mv_wide(e))
        CK(mv_wide_logdet(e, e->W, delta, 3));
    else
        CK(mv_logdet_to_slot(e, e->W, delta, 3));
    return read_scalars(e, 3, 1, out);
}

int salnmf_mv_update_W_unconstrained(salnmf_engine* e, int n_given, double lam, double delta, double* Wunc_out) {
    if (e && split(e)) CK(mv_wide_check(e));
    if (!e || !Wunc_out) return fail("null argument");
    if (n_given < 0 || n_given > e->K) return fail("n_given out of range");
    CK(enter(e));
    if (n_given >= e->K) return download(e, Wunc_out, e->W, (size_t)e->K * e->V);  // every column given: W itself (mvnmf.py:61)
    if (mv_wide(e)) {
        CK(mv_wide_prepare(e, delta));
        CK(mv_wide_root(e, lam, n_given));
        return download(e, Wunc_out, e->Wunc, (size_t)e->K * e->V);  // (W, H untouched; the trial buffers are scratch)
    }
    double gamma = 1.0;
    return mv_update_W_impl(e, n_given, lam, delta, &gamma, false, false, false, nullptr, false, nullptr, Wunc_out, nullptr);
}

int salnmf_mv_line_search(salnmf_engine* e, double lam, double delta, double* gamma_inout, const double* Wunc) {
    if (e && split(e)) CK(mv_wide_check(e));
    if (!e || !gamma_inout || !Wunc) return fail("null argument");
    e->keep_valid = false;
    CK(enter(e));
    if (mv_wide(e)) {
        HIPCK(hipMemcpyAsync(e->Wunc, Wunc, (size_t)e->K * e->V * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPCK(hipStreamSynchronize(e->stream));  // (the caller's array is free again)
        return mv_wide_line_search(e, lam, delta, gamma_inout, false, false, nullptr);
    }
    return mv_update_W_impl(e, 0, lam, delta, gamma_inout, false, false, false, nullptr, false, nullptr, nullptr, Wunc);
}

int salnmf_mv_step(salnmf_engine* e, int n_steps, int n_given, double lam, double delta, double* gamma_inout) {
    return salnmf_mv_step_objective(e, n_steps, n_given, lam, delta, gamma_inout, nullptr, 0);
}

int salnmf_mv_step_objective(salnmf_engine* e, int n_steps, int n_given, double lam, double delta, double* gamma_inout, double* objective_out,
                             int more_follows) {
    if (e && split(e)) CK(mv_wide_check(e));
    if (!e || !gamma_inout) return fail("null argument");
    if (n_steps < 1 && objective_out) return fail("n_steps must be positive");
    if (n_given < 0 || n_given > e->K) return fail("n_given out of range");
    HIPCK(hipSetDevice(e->device));
    if (mv_wide(e)) {
        CK(enter(e));
        e->keep_valid = false;
        double f = 0.0;
        for (int i = 0; i < n_steps; ++i) {
            CK(mv_wide_update_H(e));
            CK(mv_wide_update_W(e, n_given, lam, delta, gamma_inout, &f));
        }
        if (objective_out) {
            if (n_given >= e->K) return salnmf_mv_objective(e, lam, delta, objective_out);
            *objective_out = f;  // the line search's value at the accepted point = the objective of the state left behind
        }
        return 0;
    }
    // an engine left ahead by the previous call continues from there if this call is the continuation it speculated on
    const bool resume = e->mv_ahead && n_steps > 0 && e->mv_ahead_delta == delta && e->mv_ahead_given == n_given;
    if (!resume) CK(mv_settle(e));
    e->mv_ahead = false;
    e->keep_valid = false;  // (the MvNMF steps use the second H buffer themselves)
    if (e->mv_queued && n_given < e->K && n_steps > 0)
        return mv_steps_queued(e, n_steps, n_given, lam, delta, gamma_inout, objective_out, more_follows != 0, resume);
    bool ahead = resume;  // this step's update_H pass, W-only algebra and numerator pass already ran during the previous step
    for (int i = 0; i < n_steps; ++i) {
        const bool update_W = n_given < e->K;
        if (!ahead) {
            FusedParams p = fused_params(e);
            p.wkl = nullptr;  // MvNMF._update_H passes no weights (mvnmf.py:162-165)
            p.wlh = nullptr;
            p.KLpart = nullptr;  // row sums of the new H only
            // update_H + row sums of the new H; the pass's last workgroup runs the W-only algebra of the W step beside
            // it (an accepted speculation ran both already, for exactly this W)
            if (update_W) {
                mv_side_params(e, p, e->W, delta);
                CK((launch_fused<false, true, true>(e, p, mv_side_total(e))));
            } else {
                CK((launch_fused<false, true, true>(e, p)));
            }
            e->h_pending = false;
        }
        const bool was_ahead = ahead;
        CK(mv_update_W_impl(e, n_given, lam, delta, gamma_inout, true, true, i + 1 < n_steps || more_follows != 0, &ahead, was_ahead, objective_out));
    }
    if (ahead) {  // (only with more_follows: the last step's speculation was accepted)
        e->mv_ahead = true;
        e->mv_ahead_delta = delta;
        e->mv_ahead_given = n_given;
    }
    // (all signatures given: no line search ran -- the objective as a pass of its own)
    if (objective_out && n_given >= e->K) return salnmf_mv_objective(e, lam, delta, objective_out);
    return 0;
}
