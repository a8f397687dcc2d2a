// Part of salnmf.hip's translation unit (included there, inside its extern "C" block; not a stand-alone header):
// measurement entry points: kernel durations by HIP events bound to the dispatches, the sharded step timeline.
// Split out of salnmf.hip in round 5 for readability only -- one translation unit, the same static helpers and macros.

// ------------------------------------------------------------------------------------ measurement

static int ensure_events(salnmf_engine* e, size_t n) {
    while (e->events.size() < n) {
        hipEvent_t ev;
        HIPCK(hipEventCreate(&ev));
        e->events.push_back(ev);
    }
    return 0;
}

int salnmf_profile_kl_steps(salnmf_engine* e, int n_steps, int n_given, int sample_stride, double* total_ms,
                            double* fused_avg_ms, double* tail_avg_ms) {
    if (e && split(e)) return single_block(e, "the profiling entry points");
    if (!e) return fail("null engine");
    if (n_steps < 1 || n_steps > 1000000) return fail("n_steps out of range");
    if (sample_stride < 1) sample_stride = 1;
    CK(enter(e));
    const int n_samples = (n_steps + sample_stride - 1) / sample_stride;
    CK(ensure_events(e, (size_t)4 * n_samples + 2));
    hipEvent_t first = e->events[4 * (size_t)n_samples], last = e->events[4 * (size_t)n_samples + 1];
    HIPCK(hipEventRecord(first, e->stream));
    for (int i = 0; i < n_steps; ++i) {
        // every sample_stride-th step carries events (bound to its two dispatches: their own durations)
        hipEvent_t* ev = (i % sample_stride == 0) ? &e->events[4 * (size_t)(i / sample_stride)] : nullptr;
        CK(kl_step_once(e, n_given, ev));
    }
    HIPCK(hipEventRecord(last, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    double fused = 0, tail = 0;
    for (int i = 0; i < n_samples; ++i) {
        float a = 0, b = 0;
        HIPCK(hipEventElapsedTime(&a, e->events[4 * (size_t)i], e->events[4 * (size_t)i + 1]));
        HIPCK(hipEventElapsedTime(&b, e->events[4 * (size_t)i + 2], e->events[4 * (size_t)i + 3]));
        fused += a;
        tail += b;
    }
    float tot = 0;
    HIPCK(hipEventElapsedTime(&tot, first, last));
    if (total_ms) *total_ms = tot;
    if (fused_avg_ms) *fused_avg_ms = fused / n_samples;
    if (tail_avg_ms) *tail_avg_ms = tail / n_samples;
    return 0;
}

// n_steps sharded joint steps with HIP events bound to every step's two dispatches and the in-kernel stamps of the
// tail's exchange (salnmf_p2p_kernels.h: tail_p2p_kernel): where a sharded step's microseconds go, rank by rank.
//   out[0] step (wall clock of the stream / n_steps)   out[1] fused pass   out[2] tail + exchange launch   (events)
//   per row workgroup of the tail, averaged over rows and steps (s_memrealtime, 100 MHz):
//   out[3] local slab reduction  out[4] stores to the peers + flags  out[5] wait for the peers' flags
//   out[6] read + sum of the peers' rows  out[7] W row finish   out[8] the longest wait of any row and step
// all in microseconds.  Needs the peer-to-peer exchange (a world of one rank included: the rehearsal).
int salnmf_profile_sharded_steps(salnmf_engine* e, int n_steps, int n_given, double* out9) {
    if (!e || !out9) return fail("null argument");
    if (split(e)) return single_block(e, "the profiling entry points");
    if (n_steps < 1 || n_steps > 4096) return fail("n_steps out of range");
    CK(enter(e));
    const size_t count = (size_t)e->K * e->V;
    if (!(p2p_usable(e, count) && e->K <= P2P_MAX_WG)) return fail("the sharded timeline needs the peer-to-peer exchange (salnmf_p2p_connect, salnmf_set_p2p)");
    if (e->wkl || e->wlh) return fail("the sharded timeline profiles the unweighted step");
    // (with every signature given the step has no W update, hence no exchange launch: there would be nothing to report but zeros)
    if (n_given < 0 || n_given >= e->K) return fail("the sharded timeline needs a step with an exchange: n_given must be in 0..K-1");
    CK(ensure_events(e, (size_t)4 * n_steps + 2));
    unsigned long long* dstamps = nullptr;
    const size_t n_stamps = (size_t)n_steps * 6 * P2P_MAX_WG;
    HIPCK(hipMalloc(&dstamps, n_stamps * sizeof(unsigned long long)));
    hipEvent_t first = e->events[4 * (size_t)n_steps], last = e->events[4 * (size_t)n_steps + 1];
    int rc = 0;  // (from here on every exit path frees dstamps)
    if (hipMemsetAsync(dstamps, 0, n_stamps * sizeof(unsigned long long), e->stream) != hipSuccess) rc = fail("hipMemsetAsync failed");
    if (!rc && hipEventRecord(first, e->stream) != hipSuccess) rc = fail("hipEventRecord failed");
    e->p2p.stamps = dstamps;
    for (int i = 0; i < n_steps && !rc; ++i) rc = kl_step_once(e, n_given, &e->events[4 * (size_t)i]);
    e->p2p.stamps = nullptr;
    if (!rc && hipEventRecord(last, e->stream) != hipSuccess) rc = fail("hipEventRecord failed");
    // (also after a failed launch: nothing may still be writing stamps when the buffer is freed)
    if (hipStreamSynchronize(e->stream) != hipSuccess && !rc) rc = fail("hipStreamSynchronize failed");
    std::vector<unsigned long long> st(n_stamps);
    if (!rc && hipMemcpy(st.data(), dstamps, n_stamps * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = fail("stamp download failed");
    (void)hipFree(dstamps);
    if (rc) return rc;
    CK(check_abort(e));
    if (st[0] == 0 || st[5] == 0) return fail("no exchange launch wrote its stamps: the steps did not take the peer-to-peer tail");
    double fused = 0, tail = 0, seg[5] = {0, 0, 0, 0, 0}, wait_max = 0;
    for (int i = 0; i < n_steps; ++i) {
        float a = 0, b = 0;
        HIPCK(hipEventElapsedTime(&a, e->events[4 * (size_t)i], e->events[4 * (size_t)i + 1]));
        HIPCK(hipEventElapsedTime(&b, e->events[4 * (size_t)i + 2], e->events[4 * (size_t)i + 3]));
        fused += a;
        tail += b;
        for (int k = 0; k < e->K; ++k) {
            const unsigned long long* s6 = st.data() + ((size_t)i * P2P_MAX_WG + k) * 6;
            for (int j = 0; j < 5; ++j) seg[j] += (double)(s6[j + 1] - s6[j]) * 0.01;  // 100 MHz ticks -> us
            wait_max = std::max(wait_max, (double)(s6[3] - s6[2]) * 0.01);
        }
    }
    float tot = 0;
    HIPCK(hipEventElapsedTime(&tot, first, last));
    out9[0] = tot * 1e3 / n_steps;
    out9[1] = fused * 1e3 / n_steps;
    out9[2] = tail * 1e3 / n_steps;
    for (int j = 0; j < 5; ++j) out9[3 + j] = seg[j] / ((double)n_steps * e->K);
    out9[8] = wait_max;
    return 0;
}

// average duration of the forward kernel: mode 0 = W@H + objective terms, mode 2 = W@H alone (written
// to a scratch reconstruction buffer)
static int profile_forward(salnmf_engine* e, int mode, int n_calls, double* avg_ms) {
    if (!e) return fail("null engine");
    if (split(e)) return single_block(e, "the profiling entry points");
    if (n_calls < 1 || n_calls > 100000) return fail("n_calls out of range");
    CK(enter(e));
    CK(ensure_events(e, (size_t)2 * n_calls));
    FwdParams p;
    CK(fwd_params(e, p));
    double* recon = nullptr;
    if (mode == 2) {
        HIPCK(hipMalloc(&recon, (size_t)e->Np * VMAX * sizeof(double)));
        p.out = recon;
    }
    int rc = 0;
    for (int i = 0; i < n_calls && !rc; ++i) {
        // (events bound to the dispatch: the kernel's own duration)
        hipEvent_t a = e->events[2 * (size_t)i], b = e->events[2 * (size_t)i + 1];
        rc = (mode == 2) ? launch_forward<2>(e, p, 0, a, b) : launch_forward<0>(e, p, 0, a, b);
    }
    if (hipStreamSynchronize(e->stream) != hipSuccess && !rc) rc = fail("hipStreamSynchronize failed");
    double s = 0;
    for (int i = 0; i < n_calls && !rc; ++i) {
        float a = 0;
        if (hipEventElapsedTime(&a, e->events[2 * (size_t)i], e->events[2 * (size_t)i + 1]) != hipSuccess) rc = fail("hipEventElapsedTime failed");
        s += a;
    }
    if (recon) (void)hipFree(recon);
    if (!rc && avg_ms) *avg_ms = s / n_calls;
    return rc;
}

int salnmf_profile_objective(salnmf_engine* e, int n_calls, double* avg_ms) { return profile_forward(e, 0, n_calls, avg_ms); }
int salnmf_profile_reconstruct(salnmf_engine* e, int n_calls, double* avg_ms) { return profile_forward(e, 2, n_calls, avg_ms); }
