// Part of salnmf.hip's translation unit (included there, inside its extern "C" block; not a stand-alone header):
// multi-GPU entry points: RCCL communicator, peer-to-peer exchange, what the exchange layers report.
// Split out of salnmf.hip in round 5 for readability only -- one translation unit, the same static helpers and macros.

// ------------------------------------------------------------------------------------ multi-GPU

int salnmf_comm_unique_id(char* out_id) {
    if (!out_id) return fail("null argument");
    static_assert(sizeof(ncclUniqueId) <= SALNMF_UNIQUE_ID_BYTES, "id size");
    CK(rccl_bind());
    ncclUniqueId id;
    NCCLCK(ncclGetUniqueId(&id));
    memset(out_id, 0, SALNMF_UNIQUE_ID_BYTES);
    memcpy(out_id, &id, sizeof id);
    return 0;
}

int salnmf_comm_init(salnmf_engine* e, const char* id_bytes, int n_ranks, int rank) {
    if (!e || !id_bytes) return fail("null argument");
    if (e->comm) return fail("communicator already attached");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail("rank %d out of range for %d ranks", rank, n_ranks);
    CK(enter(e));
    CK(rccl_bind());
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    NCCLCK(ncclCommInitRank(&e->comm, n_ranks, id, rank));
    e->n_ranks = n_ranks;
    e->rank = rank;
    // every rank learns every shard's size (the gathers of the CorrNMF signature solves need the counts)
    int64_t* dn = nullptr;
    HIPCK(hipMalloc(&dn, (size_t)(n_ranks + 1) * sizeof(int64_t)));
    int rc = 0;
    if (hipMemcpyAsync(dn + n_ranks, &e->N, sizeof(int64_t), hipMemcpyHostToDevice, e->stream) != hipSuccess) rc = fail("hipMemcpy failed");
    if (!rc && ncclAllGather(dn + n_ranks, dn, 1, ncclInt64, e->comm, e->stream) != ncclSuccess) rc = fail("ncclAllGather of the shard sizes failed");
    e->shard_N.assign(n_ranks, 0);
    if (!rc && (hipMemcpyAsync(e->shard_N.data(), dn, (size_t)n_ranks * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
                hipStreamSynchronize(e->stream) != hipSuccess))
        rc = fail("download of the shard sizes failed");
    (void)hipFree(dn);
    if (rc) return rc;
    e->N_total = 0;
    for (int64_t n : e->shard_N) e->N_total += n;
    return 0;
}

int salnmf_comm_info(salnmf_engine* e, int* n_ranks, int* rank, int64_t* n_samples_total) {
    if (!e) return fail("null engine");
    if (n_ranks) *n_ranks = e->n_ranks;
    if (rank) *rank = e->rank;
    if (n_samples_total) *n_samples_total = sharded(e) ? e->N_total : e->N;
    return 0;
}

int salnmf_comm_observed(salnmf_engine* e, int* rccl_nranks, int* rccl_rank, int* rccl_device, int* p2p_nranks, int* p2p_inboxes_mapped,
                         int* peer_devices, int* device, char* pci_bus_id) {
    if (!e) return fail("null engine");
    HIPCK(hipSetDevice(e->device));
    int cnt = -1, urank = -1, cdev = -1;
    if (e->comm) {  // asked of the communicator, not echoed from salnmf_comm_init's arguments
        if (g_rccl.CommCount && g_rccl.CommCount(e->comm, &cnt) != ncclSuccess) cnt = -1;
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(e->comm, &urank) != ncclSuccess) urank = -1;
        if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(e->comm, &cdev) != ncclSuccess) cdev = -1;
    }
    if (rccl_nranks) *rccl_nranks = cnt;
    if (rccl_rank) *rccl_rank = urank;
    if (rccl_device) *rccl_device = cdev;
    int mapped = 0;
    for (int r = 0; r < P2P_MAX_RANKS; ++r) {
        int dev = -1;
        if (e->p2p.connected && r < e->p2p.n_ranks && e->p2p.inbox[r]) {
            ++mapped;
            hipPointerAttribute_t attr;
            // (the device an inbox lives on as THIS process numbers it; an IPC mapping of a peer process's memory may not
            // resolve, which leaves -1)
            if (hipPointerGetAttributes(&attr, e->p2p.inbox[r]) == hipSuccess)
                dev = attr.device;
            else
                (void)hipGetLastError();
        }
        if (peer_devices) peer_devices[r] = dev;
    }
    if (p2p_nranks) *p2p_nranks = e->p2p.connected ? e->p2p.n_ranks : 0;
    if (p2p_inboxes_mapped) *p2p_inboxes_mapped = mapped;
    if (device) *device = e->device;
    if (pci_bus_id) {
        memset(pci_bus_id, 0, SALNMF_PCI_BUS_ID_BYTES);
        if (hipDeviceGetPCIBusId(pci_bus_id, SALNMF_PCI_BUS_ID_BYTES - 1, e->device) != hipSuccess) {
            (void)hipGetLastError();
            pci_bus_id[0] = 0;
        }
    }
    return 0;
}

int salnmf_p2p_export(salnmf_engine* e, int n_ranks, int64_t max_count, char* handle_out) {
    if (!e || !handle_out) return fail("null argument");
    if (e->p2p.local) return fail("the peer-to-peer inbox is exported already");
    if (n_ranks < 1 || n_ranks > P2P_MAX_RANKS) return fail("peer-to-peer exchange supports 1..%d ranks, not %d", P2P_MAX_RANKS, n_ranks);
    if (max_count < 1 || max_count > (int64_t)P2P_MAX_WG * P2P_BLOCK) return fail("max_count must be in 1..%d", P2P_MAX_WG * P2P_BLOCK);
    static_assert(sizeof(hipIpcMemHandle_t) == SALNMF_P2P_HANDLE_BYTES, "handle size");
    CK(enter(e));
    e->p2p.max_count = (size_t)max_count;
    e->p2p.slot = 2 * (size_t)max_count;  // (salnmf_p2p_kernels.h: two tagged words per double)
    e->p2p.n_ranks = n_ranks;
    const size_t bytes = 2 * (size_t)n_ranks * e->p2p.slot * sizeof(double);
    HIPCK(hipExtMallocWithFlags((void**)&e->p2p.local, bytes, hipDeviceMallocUncached));
    HIPCK(hipMemset(e->p2p.local, 0, bytes));
    HIPCK(hipMalloc(&e->p2p.abort_dev, sizeof(unsigned)));
    HIPCK(hipMemset(e->p2p.abort_dev, 0, sizeof(unsigned)));
    HIPCK(hipDeviceSynchronize());  // the flags are zero before any peer can learn the handle
    hipIpcMemHandle_t h;
    HIPCK(hipIpcGetMemHandle(&h, e->p2p.local));
    memcpy(handle_out, &h, sizeof h);
    return 0;
}

int salnmf_p2p_connect(salnmf_engine* e, int rank, int n_ranks, const char* handles, int64_t n_samples_total) {
    if (!e || !handles) return fail("null argument");
    if (!e->p2p.local) return fail("salnmf_p2p_export first");
    if (e->p2p.connected) return fail("peer-to-peer exchange already connected");
    if (n_ranks != e->p2p.n_ranks || rank < 0 || rank >= n_ranks) return fail("rank %d of %d does not match the exported inbox (%d ranks)", rank, n_ranks, e->p2p.n_ranks);
    if (e->comm && (e->n_ranks != n_ranks || e->rank != rank)) return fail("rank %d of %d contradicts the RCCL communicator (%d of %d)", rank, n_ranks, e->rank, e->n_ranks);
    if (n_samples_total < e->N) return fail("n_samples_total %lld is smaller than this shard (%lld)", (long long)n_samples_total, (long long)e->N);
    CK(enter(e));
    for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) {
            e->p2p.inbox[r] = e->p2p.local;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * sizeof h, sizeof h);
        void* ptr = nullptr;
        hipError_t rc = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (rc != hipSuccess) {
            for (int q = 0; q < r; ++q)
                if (q != rank && e->p2p.inbox[q]) (void)hipIpcCloseMemHandle(e->p2p.inbox[q]);
            for (double*& b : e->p2p.inbox) b = nullptr;
            return fail("hipIpcOpenMemHandle of rank %d's inbox failed: %s", r, hipGetErrorString(rc));
        }
        e->p2p.inbox[r] = static_cast<double*>(ptr);
    }
    e->n_ranks = n_ranks;
    e->rank = rank;
    if (!e->comm) e->N_total = n_samples_total;
    e->p2p.connected = true;
    e->p2p.on = true;
    return 0;
}

int salnmf_set_p2p_timeout_ms(salnmf_engine* e, int64_t timeout_ms) {
    if (!e) return fail("null engine");
    if (timeout_ms < 1) return fail("timeout_ms must be positive");
    e->p2p.timeout_ticks = (unsigned long long)timeout_ms * 100000ull;  // the exchange kernels count the 100 MHz clock
    return 0;
}

int salnmf_set_p2p(salnmf_engine* e, int on) {
    if (!e) return fail("null engine");
    if (on && !e->p2p.connected) return fail("peer-to-peer exchange is not connected");
    if (on && e->p2p.abort_dev) {
        unsigned gave_up = 0;
        HIPCK(hipMemcpy(&gave_up, e->p2p.abort_dev, sizeof gave_up, hipMemcpyDeviceToHost));
        if (gave_up) return fail("the peer-to-peer exchange gave up earlier on this engine and cannot be switched on again");
    }
    if (!on && e->p2p.connected && !e->comm) return fail("without an RCCL communicator the peer-to-peer exchange cannot be switched off");
    e->p2p.on = on != 0;
    if (!on && e->pabort && *e->pabort == 2u) {
        // an exchange gave up: RCCL takes over and the caller uploads W and H again.  (The exchange stays unusable on every
        // rank after that: the ranks' sequence numbers are no longer known to agree.)
        HIPCK(hipStreamSynchronize(e->stream));
        *e->pabort = 0;
    }
    return 0;
}

void* salnmf_device_ptr(salnmf_engine* e, int which) {
    if (!e) return nullptr;
    if (enter(e)) return nullptr;
    switch (which) {
        case SALNMF_BUF_G: return e->red;
        case SALNMF_BUF_W: return e->W;
        case SALNMF_BUF_H:
            if (flush_H_scale(e)) return nullptr;
            return e->H;
        case SALNMF_BUF_X: return e->X;
        case SALNMF_BUF_OBJ: return e->scal;
        case SALNMF_BUF_RED: return e->red;
        default: return nullptr;
    }
}

void* salnmf_stream(salnmf_engine* e) { return e ? (void*)e->stream : nullptr; }

int salnmf_sync(salnmf_engine* e) {
    if (!e) return fail("null engine");
    CK(enter(e));
    HIPCK(hipStreamSynchronize(e->stream));
    return check_abort(e);
}
