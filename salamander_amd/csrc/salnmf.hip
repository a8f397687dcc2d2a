// Host-side engine and C ABI (include/salnmf.h) of the MI355X KL-NMF update path.
// One engine = one GPU = one shard of the sample axis.  All launches go to the
// engine's own HIP stream; nothing inside a step synchronises with the host.
#include "../../include/salnmf.h"
#include "salnmf_launch.h"
#include "salnmf_mv_kernels.h"
#include "salnmf_mv_wide_kernels.h"
#include "salnmf_corr_kernels.h"
#include "salnmf_corr_lockstep.h"
#include "salnmf_init_kernels.h"
#include "salnmf_p2p_kernels.h"

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace salnmf;

#ifdef SALNMF_DEV_POISON
// development builds: every device allocation starts as NaNs (all bits set), so that a read of a buffer nobody has
// written yet shows up in the results deterministically instead of depending on what the allocator hands back
static hipError_t salnmf_poison_malloc(void** p, size_t n) {
    hipError_t rc = (hipMalloc)(p, n);
    if (rc == hipSuccess) rc = hipMemset(*p, 0xFF, n);
    if (rc == hipSuccess) rc = hipDeviceSynchronize();  // (the fill runs on the null stream, the engine's streams do not wait for it)
    return rc;
}
#define hipMalloc(p, n) salnmf_poison_malloc((void**)(p), (n))
#endif

static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

// ---- RCCL is bound at first use, not at load time.  The PyTorch-ROCm wheel ships its own librccl.so (SONAME
// librccl.so.1, requested by libtorch_hip.so under the unversioned name), the system has /opt/rocm/lib/librccl.so.1.
// A DT_NEEDED entry would pick whichever is visible when THIS library is loaded: loaded before torch, it would pull in
// the system copy and a later `import torch` a second one -- two RCCLs (built against different HIP runtimes) in one
// process.  Bound lazily, the choice is made when a communicator is first needed, and by then every torch.distributed
// program has imported torch: dlopen(RTLD_NOLOAD) finds the copy torch loaded (glibc matches loaded objects by SONAME)
// and both sides share it.  Only a process without torch falls through to the system library.  Nothing here opens
// torch's copy by path (doing so from a process that never initialises torch aborted in that copy's static destructors at
// exit, round 1).
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // optional (salnmf_comm_observed): what the communicator itself says about its size, this rank and its device
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;

static int rccl_bind() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);  // engines of different host threads may bind concurrently
    if (g_rccl.handle) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // the copy already in the process (torch's)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("RCCL is not available: %s", dlerror());
    RcclApi api;
    api.handle = h;
    bool ok = true;
#define SALNMF_BIND(field, name)                                      \
    *(void**)(&api.field) = dlsym(h, name);                           \
    ok = ok && api.field != nullptr;
    SALNMF_BIND(GetUniqueId, "ncclGetUniqueId")
    SALNMF_BIND(CommInitRank, "ncclCommInitRank")
    SALNMF_BIND(CommDestroy, "ncclCommDestroy")
    SALNMF_BIND(AllReduce, "ncclAllReduce")
    SALNMF_BIND(AllGather, "ncclAllGather")
    SALNMF_BIND(Broadcast, "ncclBroadcast")
    SALNMF_BIND(GroupStart, "ncclGroupStart")
    SALNMF_BIND(GroupEnd, "ncclGroupEnd")
    SALNMF_BIND(GetErrorString, "ncclGetErrorString")
#undef SALNMF_BIND
    if (!ok) return fail("the RCCL library in this process lacks a symbol this engine needs");
    *(void**)(&api.CommCount) = dlsym(h, "ncclCommCount");
    *(void**)(&api.CommUserRank) = dlsym(h, "ncclCommUserRank");
    *(void**)(&api.CommCuDevice) = dlsym(h, "ncclCommCuDevice");
    g_rccl = api;
    return 0;
}
// every use below goes through the bound table (a communicator exists only after rccl_bind succeeded)
#define ncclGetUniqueId g_rccl.GetUniqueId
#define ncclCommInitRank g_rccl.CommInitRank
#define ncclCommDestroy g_rccl.CommDestroy
#define ncclAllReduce g_rccl.AllReduce
#define ncclAllGather g_rccl.AllGather
#define ncclBroadcast g_rccl.Broadcast
#define ncclGroupStart g_rccl.GroupStart
#define ncclGroupEnd g_rccl.GroupEnd
#define ncclGetErrorString g_rccl.GetErrorString

#define HIPCK(call)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NCCLCK(call)                                                                             \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
#define CK(call)               \
    do {                       \
        int rc_ = (call);      \
        if (rc_) return rc_;   \
    } while (0)

// up to this many tiles (16 samples each) the one-workgroup multi-step kernel (salnmf_small.hip) beats two launches per
// step: 5.2 against 10.9 us per step at 64 samples, 7.3 against 10.9 at 128; from 12 tiles on (c1: 11.1 against 10.9) one
// CU's four matrix pipes are the bound -- 60 MFMAs per tile whatever K <= 16 is -- and the per-step path, which spreads
// the tiles over one CU per four of them, is as fast or faster (profiles/r04/small_cohorts.md)
constexpr int SMALL_TILES_DEFAULT = 8;

struct salnmf_engine {
    int device = 0;
    int V = 0, K = 0;
    int NB = 1;    // 96-feature blocks of X and W: 1 unless n_features > 96 (then only the KLNMF entry points are available)
    // n_signatures > 64: chunks of <= 64 signatures (rows k0 .. k0 + K - 1 of W; H is stored chunk-major [NC][Np][64]),
    // each with the kernel geometry of an engine of that many signatures; only the KLNMF entry points are available
    struct Chunk { int k0, K, KS, KTM, KR; };
    int NC = 1;
    std::vector<Chunk> kc;
    double* PR = nullptr;    // [Np][96] the product H W accumulated over the chunks, then the ratio X / (H W) (NC > 1)
    double* Gblk = nullptr;  // [NB][K][96] reduced numerators of the feature blocks (NB > 1)
    double* Uacc = nullptr;  // [Np][KP] running sum of U = R W^T over the feature blocks (NB > 1)
    int64_t N = 0, Np = 0, ntiles = 0;  // Np = 16 * ntiles: rows of the padded device layout
    int KS = 0;    // instantiated contraction depth (k-steps of 4) covering K
    int KP = 0;    // leading dimension of H on the device = 16 * ceil(KS / 4)
    int KTM = 0;   // 16-wide signature tiles done on MFMA on the output side of G and U
    int KR = 0;    // remainder columns (K - 16*KTM, <= 4) done on the VALU; 0 = none
    double* scratch = nullptr;  // compact staging buffer for layout conversion (lazily sized)
    size_t scratch_n = 0;
    int cus = 0;    // compute units of the device
    int grid = 0;   // workgroups of the fused kernel (one per CU)
    int fgrid = 0;  // workgroups of the forward kernels (two per CU)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // MvNMF: the small W-only kernels run here, beside the passes over the samples
    hipStream_t stream3 = nullptr;  // MvNMF: read-back of the line-search scalars past a speculative pass on the main stream
    hipEvent_t evW = nullptr, evPrepW = nullptr, evTrial = nullptr, evLogdet = nullptr, evObj = nullptr;
    double* Halt = nullptr;      // [Np][KP] second H buffer: the speculative update_H pass of MvNMF; the state kept by salnmf_kl_step_keep (lazily allocated)
    double* Wkeep = nullptr;     // [K][V] W of the state kept by salnmf_kl_step_keep (lazily allocated)
    double* Hkeep = nullptr;     // [Np][KP] H of the kept state of an engine with feature blocks (its steps use Halt themselves)
    double* Wdst = nullptr;      // where the next W tail writes its result (null = in place); set for one step by a kept block
    bool keep_valid = false, keep_has_W = false;
    double* objring = nullptr;   // [SALNMF_OBJECTIVE_SLOTS] device copies of the queued objectives (sharded engines: all-reduced in place)
    double* objpin = nullptr;    // [SALNMF_OBJECTIVE_SLOTS] pinned host ring the queued objectives land in (salnmf_objective_async)
    std::vector<hipEvent_t> objev;  // per slot: completion of the kernel that wrote it
    int mv_grid = 0, mv_fgrid = 0;  // grids that leave one CU free for stream2
    int mv_slabs = 0, mv_hparts = 0;  // workgroups of the last MvNMF numerator pass / of the update_H pass before it (what the tail reduces)
    double *X = nullptr, *H = nullptr, *W = nullptr, *wkl = nullptr, *wlh = nullptr;
    double *wones = nullptr, *wzeros = nullptr;  // [Np] fillers the weighted kernel loads from where a weight vector is absent
    double* Gpart = nullptr;     // [grid][K][VMAX]
    double* Hsumpart = nullptr;  // [grid][K]
    double* KLpart = nullptr;    // [grid]
    double* KLpart2 = nullptr;   // [grid] KL partials of a speculative update_H pass (the trial's objective)
    double* red = nullptr;       // [K*V | K | 1 | pad]  G, rowsums_H, KL of the local shard (then all-reduced)
    double* objpart = nullptr;   // [grid]
    double* scal = nullptr;      // device scalars: [0]=objective, [1]=f0, [2]=f1, [3]=logdet0, [4]=logdet1, [5..7] CorrNMF sums
    unsigned* klcnt = nullptr;   // arrival counter of the in-launch KL sum of the MvNMF update_H pass (zero between launches)
    double* xlx = nullptr;       // [NB][Np][16] x-only constants of the KL terms per (sample, lane column) (tile_kl), lazily computed
    bool xlx_valid = false;
    double* Wunc = nullptr;      // MvNMF scratch [K][V]
    double* Wtrial = nullptr;    // [K][V]
    double* mvA = nullptr;       // [K][V]  W @ Y_minus
    double* mvB = nullptr;       // [K][V]  W @ |Y|
    double* cs = nullptr;        // [KP], filler 1
    double* hpin = nullptr;      // pinned host scalars
    // correlated NMF (row f1): scalings, embeddings, aux; allocated by salnmf_corr_configure
    int dim = 0;                 // dim_embeddings, 0 = not configured
    double* alpha = nullptr;     // [Np] sample scalings
    double* beta = nullptr;      // [K]  signature scalings
    double* Lemb = nullptr;      // [K][dim]
    double* Uemb = nullptr;      // [N][dim]
    double* aux = nullptr;       // [Np][KP], padded like H
    double* xrowsum = nullptr;   // [Np]
    double* corrpart = nullptr;  // [cgrid][64] partial sums
    int cgrid = 0;
    bool xrowsum_valid = false, lgam_valid = false;
    bool h_pending = false;      // H is to be read as clip(H * cs): the rescale of an accepted MvNMF trial, applied by the next reader
    // salnmf_mv_step_objective(more_follows): the engine was left AHEAD -- the accepted trial is W, but H already holds the
    // next step's update (the pre-update H, unscaled, sits in Halt with its scale in cs) and that step's W-only algebra
    // and numerator slabs are done.  The next mv_step with the same (delta, n_given) continues from there; every other
    // entry point first steps back to the plain accepted state (mv_settle: a swap, no kernel)
    bool mv_ahead = false;
    double mv_ahead_delta = 0.0;
    int mv_ahead_given = 0;
    double lgam_sum = 0.0;       // sum gammaln(1 + X) over the local shard
    // ingest: two pinned host buffers + two device buffers of STAGE_BYTES each, reused by every upload
    void* stage_host[2] = {nullptr, nullptr};
    void* stage_dev[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    unsigned* psync = nullptr;   // persistent kernel: device sync words (SYNC_WORDS), zeroed before every launch
    unsigned* pabort = nullptr;  // pinned host word the persistent kernel sets when a wait gives up
    bool persistent = false;     // multi-step kl_step calls run as one persistent launch (opt-in: salnmf_set_persistent)
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    // opt-in fp32 fast mode of the KL step (salnmf_kernels_f32.h): fp32 shadow copies of X and H, made when needed
    bool fast32 = false, x32_valid = false;
    float *X32 = nullptr, *H32 = nullptr;
    // peer-to-peer exchange of the small all-reduces (salnmf_p2p_kernels.h); RCCL stays for everything larger
    struct {
        bool connected = false, on = false;
        double* local = nullptr;                // this rank's inbox (uncached device memory, exported over hipIpc)
        double* inbox[P2P_MAX_RANKS] = {};      // every rank's inbox as mapped into this process
        size_t max_count = 0, slot = 0;
        int n_ranks = 0;                        // as exported
        unsigned long long seq = 0;
        unsigned* abort_dev = nullptr;          // device word: an exchange gave up (salnmf_p2p_kernels.h)
        unsigned long long* stamps = nullptr;  // salnmf_profile_sharded_steps: where the next tail_p2p launch writes its stamps
        unsigned long long timeout_ticks = P2P_TIMEOUT_TICKS;
    } p2p;
    std::vector<int64_t> shard_N;  // n_samples of every rank's shard (filled by salnmf_comm_init)
    int64_t N_total = 0;           // sum of shard_N
    // gathered inputs of the signature-embedding solves (all samples of all shards, compact rows)
    // lockstep signature solves (salnmf_corr_lockstep.h): per-signature logs and per-round partial sums, lazily sized
    double* ls_buf = nullptr;
    int* ls_int = nullptr;
    size_t ls_doubles = 0;
    int ls_S = 0, ls_dim = 0;
    bool lockstep = true;  // salnmf_set_lockstep(e, 0) forces the single-kernel form
    bool batched_samples = true;  // salnmf_set_batched_sample_solves(e, 0) forces one wavefront per sample
    double* mvS = nullptr;       // (MvNMF on signature chunks) [K][2K] scratch of the global-memory elimination
    unsigned long long* fk_prof = nullptr;  // (SALNMF_DEV_PROFILE builds) section clocks of the fused passes, printed by salnmf_destroy
    bool w_dma = true;           // salnmf_set_w_dma(e, 0): the update passes stage W through registers instead of by LDS-DMA
    bool mv_queued = true;       // salnmf_set_mv_queued(e, 0): MvNMF steps with the host's line-search decision per step (the classic form)
    unsigned* mvflag = nullptr;  // device word of the queued MvNMF steps: non-zero = a trial was rejected on the device
    int small_max_tiles = SMALL_TILES_DEFAULT;  // salnmf_set_small_cohort_tiles: up to this many tiles salnmf_kl_step runs as one workgroup
    hipEvent_t ls_ev[2] = {nullptr, nullptr};  // lockstep rounds: the count of live solves has reached the host
    double *gU = nullptr, *galpha = nullptr, *gaux = nullptr;
    size_t g_rows = 0;
    int g_dim = 0;
    std::vector<hipEvent_t> events;
};

static int check_abort(salnmf_engine* e);
constexpr size_t SMALL_PINNED_BYTES = 4096;  // per-engine pinned block: read-back scalars, the abort word at its middle
static void release_pinned(void* p, int small_block);
static hipError_t acquire_pinned(void** out, int small_block);

constexpr int NB_MAX = 32;  // feature blocks of 96: n_features <= 3072 (SBS-1536 needs 16)
constexpr int KC = 64;      // signatures per chunk (the widest accumulator geometry of the fused pass)
constexpr int NC_MAX = 8;   // signature chunks: n_signatures <= 512

static const int kKS[] = {1, 2, 4, 8, 10, 13, 16};

static int pick_ks(int K) {
    int need = (K + 3) / 4;
    for (int ks : kKS)
        if (ks >= need) return ks;
    return -1;
}

// ------------------------------------------------------------------------------------ launches

// an engine left ahead by salnmf_mv_step_objective(more_follows) steps back to the plain accepted state of its last step:
// W = the accepted trial, H = the pre-update exposures read as clip(H * cs) -- exactly what a non-speculative accept leaves
static int mv_settle(salnmf_engine* e) {
    if (!e->mv_ahead) return 0;
    std::swap(e->H, e->Halt);
    e->h_pending = true;
    e->mv_ahead = false;
    HIPCK(hipEventRecord(e->evW, e->stream));  // W is final for a later stand-alone W-only kernel on stream2
    return 0;
}
// every entry point that takes an engine: its device current, no half-finished MvNMF step
static int enter(salnmf_engine* e) {
    HIPCK(hipSetDevice(e->device));
    return mv_settle(e);
}

// the always-valid weight arrays of a weighted launch (salnmf_kernels.h: FusedParams::wkl_eff): the vectors themselves, or
// fillers of ones / zeros where one of the two is absent
static int weight_arrays(salnmf_engine* e, FusedParams& p) {
    auto filler = [&](double*& buf, double value) -> int {
        if (buf) return 0;
        HIPCK(hipMalloc(&buf, (size_t)e->Np * sizeof(double)));
        std::vector<double> host((size_t)e->Np, value);
        HIPCK(hipMemcpyAsync(buf, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPCK(hipStreamSynchronize(e->stream));
        return 0;
    };
    if (!p.wkl) CK(filler(e->wones, 1.0));
    if (!p.wlh) CK(filler(e->wzeros, 0.0));
    p.wkl_eff = p.wkl ? p.wkl : e->wones;
    p.wlh_eff = p.wlh ? p.wlh : e->wzeros;
    return 0;
}

// ev_start / ev_stop (profiling only): bound to the dispatch itself, so that their elapsed time is the kernel's own
// duration, as rocprofv3 reports it -- events recorded around the launch add their barrier packets to it.
// The instantiations live in salnmf_fused_inst.hip / salnmf_forward_inst.hip (salnmf_launch.h).
static int flush_H_scale(salnmf_engine* e);
template <bool DO_G, bool DO_U, bool DO_STATS>
static int launch_fused(salnmf_engine* e, const FusedParams& p, int grid = 0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    // per-sample weights select the WTS instantiation (KLNMF only: the MvNMF / CorrNMF passes, which are the
    // ones that collect statistics, are unweighted)
    const bool wts = p.wkl || p.wlh;
    if (wts && DO_STATS) return fail("internal: weighted pass with statistics is not instantiated");
    if (DO_STATS && (DO_G || p.KLpart) && (!p.xlx || !e->xlx_valid)) return fail("internal: KL pass without the x-only constants (ensure_xlogx)");
    const FusedSel sel{e->KS, e->KTM, e->KR, DO_G, DO_U, DO_STATS, wts, false};
    FusedParams pw = p;
    if (wts) CK(weight_arrays(e, pw));
    if (DO_G && DO_U && pw.hscale != nullptr) {
        // the joint steps read their H tiles by LDS-DMA (fused_kernel: HDMA), i.e. as they are in memory: a pending
        // rescale (an accepted MvNMF trial, salnmf_set_H_scale) is applied as a pass of its own first
        if (pw.hscale != e->cs || pw.H != e->H || !e->h_pending) return fail("internal: the joint step cannot apply a foreign exposure scale on the fly");
        CK(flush_H_scale(e));
        pw.hscale = nullptr;
    }
    if (launch_fused_inst(sel, pw, grid > 0 ? grid : e->grid, e->stream, ev_start, ev_stop))
        return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", e->KS, e->KTM, e->KR);
    HIPCK(hipGetLastError());
    return 0;
}

template <int MODE>
static int launch_forward(salnmf_engine* e, const FwdParams& p, int grid = 0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    if (launch_forward_inst(e->KS, MODE, p, grid > 0 ? grid : e->fgrid, e->stream, ev_start, ev_stop))
        return fail("no kernel instantiation for KS=%d", e->KS);
    HIPCK(hipGetLastError());
    return 0;
}

constexpr size_t FK_PROF_ROWS = 4096;  // (SALNMF_DEV_PROFILE) waves of the largest grid
static FusedParams fused_params(salnmf_engine* e) {
    FusedParams p{};  // (the optional parts -- MvNMF side workgroup, in-launch KL sum, persistent mode -- are off)
    p.X = e->X;
    p.H = e->H;
    p.Hout = e->H;
    p.hfloor = kEps;
    p.W = e->W;
    p.wkl = e->wkl;
    p.wlh = e->wlh;
    p.hscale = e->h_pending ? e->cs : nullptr;
    p.Gpart = e->Gpart;
    p.Hsumpart = e->Hsumpart;
    p.KLpart = e->KLpart;
    p.xlx = e->xlx;  // (valid whenever a DO_STATS pass runs: ensure_xlogx)
    p.N = e->N;
    p.V = e->V;
    p.ldw = e->V;
    p.K = e->K;
    p.ntiles = e->ntiles;
    p.wdma = e->w_dma ? 1 : 0;
#ifdef SALNMF_DEV_PROFILE
    if (!e->fk_prof && hipMalloc(&e->fk_prof, FK_PROF_ROWS * (FK_NSEC + 1) * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(e->fk_prof, 0, FK_PROF_ROWS * (FK_NSEC + 1) * sizeof(unsigned long long));
    p.prof = e->fk_prof;
#endif
    return p;
}

// ---- feature blocks (n_features > 96): block b of X, the matching 96 columns of W
static inline int block_width(const salnmf_engine* e, int b) { return std::min(VMAX, e->V - VMAX * b); }
static void to_block(const salnmf_engine* e, FusedParams& p, int b) {
    p.X = e->X + (size_t)b * e->Np * VMAX;
    p.W = e->W + (size_t)VMAX * b;
    p.V = block_width(e, b);
    p.ldw = e->V;
}
static int single_block(const salnmf_engine* e, const char* what) {
    if (e->NB > 1) return fail("%s is not available for n_features > %d (this engine has %d): only the KLNMF entry points are", what, VMAX, e->V);
    if (e->NC > 1) return fail("%s is not available for n_signatures > %d (this engine has %d): only the KLNMF entry points are", what, KC, e->K);
    return 0;
}
static inline bool split(const salnmf_engine* e) { return e->NB > 1 || e->NC > 1; }  // feature blocks or signature chunks
static inline size_t h_doubles(const salnmf_engine* e) { return (size_t)e->NC * e->Np * e->KP; }

static TailParams tail_params(salnmf_engine* e, int nslabs, double* G, int n_given, int clip_mode, int do_tail, bool with_stats, int hsum_parts = 0) {
    TailParams t{};
    t.Gpart = e->Gpart;
    t.G = G;
    t.W = e->W;
    t.Wout = e->Wdst ? e->Wdst : e->W;
    t.nslabs = nslabs;
    t.V = e->V;
    t.K = e->K;
    t.n_given = n_given;
    t.clip_mode = clip_mode;
    t.do_tail = do_tail;
    t.hsum_part = with_stats ? e->Hsumpart : nullptr;
    t.hsum_out = e->red + (size_t)e->K * e->V;
    t.kl_part = with_stats ? e->KLpart : nullptr;
    t.kl_out = e->red + (size_t)e->K * e->V + e->K;
    t.nparts = nslabs;
    t.nparts_h = hsum_parts > 0 ? hsum_parts : nslabs;
    return t;
}

static int launch_tail(salnmf_engine* e, int nslabs, double* G, int n_given, int clip_mode, int do_tail, bool with_stats = false,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int hsum_parts = 0) {
    if (ev_stop)
        hipExtLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, ev_start, ev_stop, 0,
                              tail_params(e, nslabs, G, n_given, clip_mode, do_tail, with_stats, hsum_parts));
    else
        hipLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, tail_params(e, nslabs, G, n_given, clip_mode, do_tail, with_stats, hsum_parts));
    HIPCK(hipGetLastError());
    return 0;
}

// the engine holds one shard of the samples: sums over the samples are all-reduced
static inline bool sharded(const salnmf_engine* e) { return e->comm != nullptr || e->p2p.connected; }

// parameters of the next peer-to-peer exchange (every rank issues the same sequence of exchanges)
static P2PParams next_exchange(salnmf_engine* e, double* buf, size_t count) {
    P2PParams q{};
    q.buf = buf;
    q.count = (int)count;
    q.rank = e->rank;
    q.n_ranks = e->n_ranks;
    q.seq = ++e->p2p.seq;
    q.tag = ((q.seq % 0xFFFFFFFFull) + 1ull) << 32;  // (salnmf_p2p_kernels.h: never 0, differs between exchanges s and s + 2)
    q.parity = (int)(q.seq & 1);
    q.slot = e->p2p.slot;
    q.max_count = e->p2p.max_count;
    for (int r = 0; r < e->n_ranks; ++r) q.inbox[r] = e->p2p.inbox[r];
    q.abort_host = e->pabort;
    q.abort_dev = e->p2p.abort_dev;
    q.timeout_ticks = e->p2p.timeout_ticks;
    q.stamps = e->p2p.stamps;  // (null outside salnmf_profile_sharded_steps)
    if (e->p2p.stamps) e->p2p.stamps += (size_t)6 * P2P_MAX_WG;
    return q;
}

static inline bool p2p_usable(const salnmf_engine* e, size_t count) { return e->p2p.connected && e->p2p.on && count <= e->p2p.max_count; }

static int allreduce(salnmf_engine* e, double* buf, size_t count) {
    if (p2p_usable(e, count)) {
        hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(((int)count + P2P_BLOCK - 1) / P2P_BLOCK), dim3(P2P_BLOCK), 0, e->stream,
                           next_exchange(e, buf, count));
        HIPCK(hipGetLastError());
        return 0;
    }
    if (!e->comm) {
        if (e->p2p.connected) return fail("an all-reduce of %zu doubles needs the RCCL communicator (peer-to-peer exchange: %s, limit %zu)", count,
                                          e->p2p.on ? "on" : "off", e->p2p.max_count);
        return 0;
    }
    NCCLCK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, e->comm, e->stream));
    return 0;
}

// W tail of a sharded step: local reduction of the numerator slabs, all-reduce over the ranks, W update.
// With the peer-to-peer exchange all of it is one launch (tail_p2p_kernel), otherwise reduce launch + RCCL + finish launch.
static int sharded_tail(salnmf_engine* e, int n_given, int clip_mode) {
    const size_t count = (size_t)e->K * e->V;
    if (p2p_usable(e, count) && e->K <= P2P_MAX_WG) {
        TailP2PParams tp;
        tp.t = tail_params(e, e->grid, e->red, n_given, clip_mode, 1, false);
        tp.x = next_exchange(e, e->red, count);
        hipLaunchKernelGGL(tail_p2p_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, tp);
        HIPCK(hipGetLastError());
        return 0;
    }
    CK(launch_tail(e, e->grid, e->red, 0, 0, 0));
    CK(allreduce(e, e->red, count));
    return launch_tail(e, 0, e->red, n_given, clip_mode, 1);
}

// The second H buffer (kept steps, MvNMF speculation): allocated at first use and ZEROED.  A pass that writes it covers
// every row, but only the columns it computes (16 KTM + KR of the KP padded ones): the pad columns must hold finite
// filler before the buffer is ever read as H -- they meet zero rows of W in the P product, and 0 * NaN is NaN.  (Found
// with SALNMF_DEV_POISON: with K = 17, 33, 42-50 an MvNMF step that continued from an accepted speculation read
// whatever hipMalloc had handed back.)
static int ensure_halt(salnmf_engine* e) {
    if (e->Halt) return 0;
    const size_t bytes = (size_t)e->NC * e->Np * e->KP * sizeof(double);
    HIPCK(hipMalloc(&e->Halt, bytes));
    HIPCK(hipMemsetAsync(e->Halt, 0, bytes, e->stream));
    return 0;
}

// the x-only part of the KL divergence (salnmf_kernels.h: tile_kl), once per upload of X: the constants per (sample, lane column)
static int ensure_xlogx(salnmf_engine* e) {
    if (e->xlx_valid) return 0;
    if (!e->xlx) HIPCK(hipMalloc(&e->xlx, (size_t)e->NB * e->Np * 16 * sizeof(double)));
    for (int b = 0; b < e->NB; ++b) {  // (feature blocks: the constants of every block's own features)
        hipLaunchKernelGGL(xlogx_lane_kernel, dim3((unsigned)((e->Np + 15) / 16)), dim3(256), 0, e->stream, e->X + (size_t)b * e->Np * VMAX, e->Np,
                           block_width(e, b), VMAX, e->xlx + (size_t)b * e->Np * 16);
        HIPCK(hipGetLastError());
    }
    e->xlx_valid = true;
    return 0;
}

// materialise a pending rescale of H (needed only by readers that cannot apply it on the fly)
static int flush_H_scale(salnmf_engine* e) {
    if (!e->h_pending) return 0;
    for (int ci = 0; ci < e->NC; ++ci) {  // (signature chunks: chunk ci's columns are the signatures from kc[ci].k0 on)
        hipLaunchKernelGGL(scale_H_kernel, dim3(2048), dim3(256), 0, e->stream, e->H + (size_t)ci * e->Np * e->KP, e->cs + e->kc[(size_t)ci].k0,
                           (int64_t)e->Np * e->KP, e->KP);
        HIPCK(hipGetLastError());
    }
    e->h_pending = false;
    return 0;
}

// one joint step; ev != nullptr: {fused start, fused stop, tail start, tail stop} are bound to the two dispatches
// (a weighted or sharded step, whose tail is more than one launch, falls back to records around the launches)
//   keep: this step leaves the state it starts from untouched -- the new H goes to the second H buffer, the new W to
//   Wkeep -- and the buffers change roles afterwards (salnmf_kl_step_keep)
//   obj_slot >= 0: the step also evaluates the objective of the state it starts from (the fused pass has P = H W of that
//   state in registers anyway: one logarithm per entry on top, instead of a forward pass of its own) -- its KL partials are
//   reduced by an extra workgroup of the tail launch straight into slot obj_slot of the pinned ring, and the slot's event
//   is that launch's completion signal.  Unweighted, unsharded, n_given < K (salnmf_kl_step_objective checks).
static int kl_step_once(salnmf_engine* e, int n_given, hipEvent_t* ev, bool keep = false, int obj_slot = -1) {
    struct ResetWdst {  // (a kept step redirects the W tail for the duration of this call only, also when a launch fails)
        salnmf_engine* e;
        ~ResetWdst() { e->Wdst = nullptr; }
    } reset_wdst{e};
    FusedParams p = fused_params(e);
    const bool all_given = n_given >= e->K;  // _utils_klnmf.py:330-331: W untouched
    if (obj_slot >= 0) {
        if (keep) {
            p.Hout = e->Halt;
            e->Wdst = e->Wkeep;
        }
        CK((launch_fused<true, true, true>(e, p)));
        e->h_pending = false;
        TailParams t = tail_params(e, e->grid, e->red, n_given, SALNMF_CLIP_ALL, 1, false);
        t.kl_part = e->KLpart;
        t.kl_out = e->objpin + obj_slot;
        t.nparts = e->grid;
        t.kl_extra = 1;
        hipExtLaunchKernelGGL(tail_kernel, dim3(e->K + 1), dim3(TAIL_BLOCK), 0, e->stream, nullptr, e->objev[obj_slot], 0, t);
        HIPCK(hipGetLastError());
        if (keep) {
            std::swap(e->H, e->Halt);
            std::swap(e->W, e->Wkeep);
            e->Wdst = nullptr;
            e->keep_has_W = true;
        }
        return 0;
    }
    if (keep) {
        p.Hout = e->Halt;
        e->Wdst = all_given ? nullptr : e->Wkeep;
    }
    const bool bound = ev && !(p.wkl || p.wlh);
    if (ev && !bound) HIPCK(hipEventRecord(ev[0], e->stream));
    if (all_given)
        CK((launch_fused<false, true, false>(e, p, 0, bound ? ev[0] : nullptr, bound ? ev[1] : nullptr)));
    else
        CK((launch_fused<true, true, false>(e, p, 0, bound ? ev[0] : nullptr, bound ? ev[1] : nullptr)));
    e->h_pending = false;  // the pass wrote H in full
    if (ev && !bound) HIPCK(hipEventRecord(ev[1], e->stream));
    const bool tail_bound = ev && !all_given && !sharded(e);
    if (ev && !tail_bound) HIPCK(hipEventRecord(ev[2], e->stream));
    if (!all_given) {
        if (sharded(e)) {
            CK(sharded_tail(e, n_given, SALNMF_CLIP_ALL));
        } else {
            CK(launch_tail(e, e->grid, e->red, n_given, SALNMF_CLIP_ALL, 1, false, tail_bound ? ev[2] : nullptr, tail_bound ? ev[3] : nullptr));
        }
    }
    if (ev && !tail_bound) HIPCK(hipEventRecord(ev[3], e->stream));
    if (keep) {
        std::swap(e->H, e->Halt);
        if (!all_given) std::swap(e->W, e->Wkeep);
        e->Wdst = nullptr;
        e->keep_has_W = !all_given;
    }
    return 0;
}

template <typename T>
static void launch_pad_rows(salnmf_engine* e, double* dst, const void* src, int64_t rows, int cols, int ld, double fill_cols, double clip_lo) {
    const int grid = (int)std::min<int64_t>(2048, (rows * ld + 255) / 256);
    hipLaunchKernelGGL(pad_rows_kernel<T>, dim3(std::max(grid, 1)), dim3(256), 0, e->stream, dst, static_cast<const T*>(src), rows, cols, ld, fill_cols, clip_lo);
}

// workgroups of a logit pass: two per CU at most (each opens with the staging of L^T)
static inline int corr_logit_grid(const salnmf_engine* e) { return std::min(e->cgrid, 2 * e->cus); }
// the logit passes of CorrNMF (corr_logit_mfma_kernel): the k-step count is a compile-time constant, dim rounded up to 16
template <int MODE>
static void launch_corr_logit(salnmf_engine* e, const CorrParams& p) {
    const dim3 g(corr_logit_grid(e)), b(CORR_BLOCK);
    switch ((e->dim + 15) / 16) {
        case 1: hipLaunchKernelGGL((corr_logit_mfma_kernel<MODE, 4>), g, b, 0, e->stream, p); break;
        case 2: hipLaunchKernelGGL((corr_logit_mfma_kernel<MODE, 8>), g, b, 0, e->stream, p); break;
        case 3: hipLaunchKernelGGL((corr_logit_mfma_kernel<MODE, 12>), g, b, 0, e->stream, p); break;
        default: hipLaunchKernelGGL((corr_logit_mfma_kernel<MODE, 16>), g, b, 0, e->stream, p); break;
    }
}

// ------------------------------------------------------------------------------------ C ABI

extern "C" {

const char* salnmf_last_error(void) { return g_err.c_str(); }
int salnmf_version(void) { return 100; }
int salnmf_build_flags(void) { return built_with_persistent() ? SALNMF_BUILD_PERSISTENT : 0; }

int salnmf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void salnmf_destroy(salnmf_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
#ifdef SALNMF_DEV_PROFILE
    if (e->fk_prof) {  // section clocks of every fused pass this engine launched (salnmf_fused_kernel.h: FK_TICK)
        std::vector<unsigned long long> rows(FK_PROF_ROWS * (FK_NSEC + 1));
        unsigned long long t[FK_NSEC + 1] = {};
        size_t nw = 0;
        if (hipMemcpy(rows.data(), e->fk_prof, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            for (size_t w = 0; w < FK_PROF_ROWS; ++w) {
                if (rows[w * (FK_NSEC + 1) + FK_NSEC] == 0 && rows[w * (FK_NSEC + 1)] == 0) continue;
                ++nw;
                for (int i = 0; i <= FK_NSEC; ++i) t[i] += rows[w * (FK_NSEC + 1) + i];
            }
        }
        if (t[FK_NSEC] > 0) {
            static const char* names[FK_NSEC] = {"prologue", "a.stage_H", "a.P_product", "a.KL_terms", "a.divisions", "a.prefetch+R_transpose", "a.G_phase",
                                                 "a.U_phase+H_update", "b.stage_H|-", "b.P_product|coop.load+stage", "b.KL_terms|coop.phase_A", "b.divisions|epi.wait_slowest_wave", "b.prefetch+R_tr|coop.phase_B",
                                                 "b.G_phase|epi.park+barrier", "b.U_phase|epi.sum+slab_stores", "epilogue"};
            unsigned long long tot = 0;
            for (int i = 0; i < FK_NSEC; ++i) tot += t[i];
            fprintf(stderr, "[salnmf dev profile] fused passes of engine N=%lld K=%d: %zu waves, %llu (wave, tile) pairs, %.0f shader-clock cycles per wave in all\n",
                    (long long)e->N, e->K, nw, t[FK_NSEC], (double)tot / nw);
            for (int i = 0; i < FK_NSEC; ++i)
                if (t[i]) {
                    const bool per_tile = i != 0 && i != FK_NSEC - 1;
                    fprintf(stderr, "  %-24s %6.2f %%  %9.0f cycles per %s\n", names[i], 100.0 * t[i] / tot, (double)t[i] / (per_tile ? (double)t[FK_NSEC] : (double)nw),
                            per_tile ? "(wave, tile)" : "wave and launch x launches");
                }
        }
        (void)hipFree(e->fk_prof);
    }
#endif
    if (e->comm) ncclCommDestroy(e->comm);
    for (int r = 0; r < P2P_MAX_RANKS; ++r)
        if (e->p2p.inbox[r] && e->p2p.inbox[r] != e->p2p.local) (void)hipIpcCloseMemHandle(e->p2p.inbox[r]);
    if (e->p2p.local) (void)hipFree(e->p2p.local);
    if (e->X32) (void)hipFree(e->X32);
    if (e->H32) (void)hipFree(e->H32);
    if (e->p2p.abort_dev) (void)hipFree(e->p2p.abort_dev);
    double* bufs[] = {e->PR, e->wones, e->wzeros, e->Gblk, e->Uacc, e->xlx, e->X, e->H, e->W, e->wkl, e->wlh, e->Gpart, e->Hsumpart, e->KLpart, e->red,
                      e->objpart, e->scal, e->Wunc, e->Wtrial, e->mvA, e->mvB, e->cs, e->scratch, e->Halt, e->KLpart2, e->Wkeep, e->Hkeep, e->objring,
                      e->alpha, e->beta, e->Lemb, e->Uemb, e->aux, e->xrowsum, e->corrpart, e->gU, e->galpha, e->gaux, e->mvS};
    for (double* b : bufs)
        if (b) (void)hipFree(b);
    if (e->hpin) release_pinned(e->hpin, 1);  // (one block: hpin, and the abort word behind it)
    if (e->objpin) release_pinned(e->objpin, 1);
    for (hipEvent_t ev : e->objev)
        if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < 2; ++i) {
        if (e->stage_host[i]) release_pinned(e->stage_host[i], 0);  // (the engine's streams are idle: synchronised above)
        if (e->stage_dev[i]) (void)hipFree(e->stage_dev[i]);
        if (e->stage_done[i]) (void)hipEventDestroy(e->stage_done[i]);
    }
    if (e->ls_buf) (void)hipFree(e->ls_buf);
    if (e->ls_int) (void)hipFree(e->ls_int);
    if (e->psync) (void)hipFree(e->psync);
    if (e->klcnt) (void)hipFree(e->klcnt);
    if (e->mvflag) (void)hipFree(e->mvflag);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {e->evW, e->evPrepW, e->evTrial, e->evLogdet, e->evObj, e->ls_ev[0], e->ls_ev[1]})
        if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t st : {e->stream2, e->stream3})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int salnmf_create(int device, int n_features, int64_t n_samples, int n_signatures, salnmf_engine** out) {
    if (!out) return fail("out is null");
    *out = nullptr;
    if (n_features < 1 || n_features > VMAX * NB_MAX) return fail("n_features must be in [1, %d], got %d", VMAX * NB_MAX, n_features);
    if (n_signatures < 1 || n_signatures > KC * NC_MAX) return fail("n_signatures must be in [1, %d], got %d", KC * NC_MAX, n_signatures);
    if (n_samples < 1) return fail("n_samples must be positive");
    int ndev = 0;
    HIPCK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
    HIPCK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail("this build targets gfx950 only; device %d is %s", device, prop.gcnArchName);

    salnmf_engine* e = new salnmf_engine();
    e->device = device;
    e->V = n_features;
    e->K = n_signatures;
    e->N = n_samples;
    e->ntiles = (n_samples + 15) / 16;
    e->Np = e->ntiles * 16;
    e->NB = (n_features + VMAX - 1) / VMAX;
    // output-side split: full MFMA tiles + up to 4 remainder columns on the VALU (K >= 17 only)
    auto geometry = [](int k0, int k) {
        salnmf_engine::Chunk c{k0, k, pick_ks(k), (k + 15) / 16, 0};
        if (k > 16 && k % 16 >= 1 && k % 16 <= 4) {
            c.KTM = k / 16;
            c.KR = k % 16;
        }
        return c;
    };
    e->NC = (n_signatures + KC - 1) / KC;
    const int ck = (n_signatures + e->NC - 1) / e->NC;  // chunks of equal size, the last one takes what is left
    for (int c = 0, k0 = 0; c < e->NC; ++c) {
        const int k = std::min(ck, n_signatures - k0);
        salnmf_engine::Chunk ch = geometry(k0, k);
        if (e->NC > 1) {
            // all chunks share the 64-column layout of H: the two contraction depths whose padded width is 64, with the
            // output side padded to whole tiles below 48 signatures (pad columns of H are 0, pad rows of W staged as 0)
            if (k <= 48) ch = {k0, k, 13, 3, 0};
            else if (k <= 52) ch = {k0, k, 13, 3, k - 48};
            else ch = {k0, k, 16, 4, 0};
        }
        e->kc.push_back(ch);
        k0 += k;
    }
    e->KS = e->kc[0].KS;
    e->KTM = e->kc[0].KTM;
    e->KR = e->kc[0].KR;
    e->KP = e->NC > 1 ? KC : 16 * ((e->KS + 3) / 4);
    int64_t wg_needed = (e->ntiles + WAVES - 1) / WAVES;
    e->grid = (int)std::min<int64_t>(prop.multiProcessorCount, wg_needed);
    e->fgrid = (int)std::min<int64_t>(2 * prop.multiProcessorCount, wg_needed);
    // MvNMF overlaps single-workgroup kernels on a second stream with the passes over the samples.  Workgroups are
    // dealt round-robin to the 8 XCDs and a one-workgroup kernel lands on the first XCD, whichever kernel is
    // dispatched first: leaving one CU per XCD free (3 % of the pass) guarantees it a place
    const int cus = e->cus = prop.multiProcessorCount;
    const int spare = (cus % 8 == 0 && cus >= 64) ? 8 : 1;
    e->mv_grid = (e->grid >= cus && e->grid > spare) ? e->grid - spare : e->grid;
    e->mv_fgrid = (e->fgrid >= 2 * cus && e->fgrid > 2 * spare) ? e->fgrid - 2 * spare : e->fgrid;
    const size_t K = e->K, V = e->V, Np = e->Np, KP = e->KP;
    auto cleanup = [&](int rc) {
        salnmf_destroy(e);
        return rc;
    };
#define ALLOC(ptr, n)                                                      \
    if (hipMalloc(&(ptr), (n) * sizeof(double)) != hipSuccess) return cleanup(fail("hipMalloc of %zu doubles failed", (size_t)(n)));
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(fail("stream create failed"));
    // (stream2 / stream3 are created by the first MvNMF call: ensure_side_streams)
    // cross-stream dependencies on ONE device (the host sees results only through explicit copies behind them): no
    // system-scope fence at the record -- 2 % of an MvNMF step (profiles/r02/ab_step_variants.txt)
    for (hipEvent_t* ev : {&e->evW, &e->evPrepW, &e->evTrial, &e->evLogdet, &e->evObj})
        if (hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return cleanup(fail("event create failed"));
    ALLOC(e->X, (size_t)e->NB * Np * VMAX);
    ALLOC(e->H, (size_t)e->NC * Np * KP);
    if (e->NC > 1) ALLOC(e->PR, Np * VMAX);
    ALLOC(e->W, K * V);
    ALLOC(e->Gpart, (size_t)e->grid * K * VMAX);
    ALLOC(e->Hsumpart, (size_t)e->grid * K);
    ALLOC(e->KLpart, (size_t)e->grid);
    ALLOC(e->red, K * V + K + 2);
    ALLOC(e->objpart, (size_t)(e->NB > 1 && e->NC > 1 ? e->NB + e->NC : std::max(e->NB, e->NC)) * e->fgrid);
    if (e->NB > 1) {
        ALLOC(e->Gblk, (size_t)e->NB * K * VMAX);
        ALLOC(e->Uacc, (size_t)e->NC * Np * KP);
    }
    ALLOC(e->scal, 16);
    ALLOC(e->Wunc, K * V);
    ALLOC(e->Wtrial, K * V);
    ALLOC(e->mvA, K * V);
    ALLOC(e->mvB, K * V);
    // (signature chunks: one entry per signature, compact, + a chunk's width of filler behind the last one -- a chunk's
    // passes read KP entries from its first signature on)
    const size_t ncs = e->NC > 1 ? K + KP : KP;
    ALLOC(e->cs, ncs);
#undef ALLOC
    {
        std::vector<double> ones(ncs, 1.0);
        if (hipMemcpy(e->cs, ones.data(), ncs * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail("hipMemcpy failed"));
    }
    if (acquire_pinned((void**)&e->hpin, 1) != hipSuccess) return cleanup(fail("hipHostMalloc failed"));
    e->pabort = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e->hpin) + SMALL_PINNED_BYTES / 2);
    if (hipMalloc(&e->psync, SYNC_WORDS * sizeof(unsigned)) != hipSuccess) return cleanup(fail("hipMalloc failed"));
    if (hipMalloc(&e->klcnt, 16) != hipSuccess || hipMemset(e->klcnt, 0, 16) != hipSuccess) return cleanup(fail("hipMalloc failed"));
    *e->pabort = 0;
    *out = e;
    return 0;
}

static int upload(salnmf_engine* e, double* dst, const double* src, size_t n) {
    if (!e || !src) return fail("null argument");
    CK(enter(e));
    HIPCK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    return 0;
}

static int ensure_scratch(salnmf_engine* e, size_t n) {
    if (e->scratch_n >= n) return 0;
    if (e->scratch) HIPCK(hipFree(e->scratch));
    e->scratch = nullptr;
    e->scratch_n = 0;
    HIPCK(hipMalloc(&e->scratch, n * sizeof(double)));
    e->scratch_n = n;
    return 0;
}

// ---- ingest (row f4): pinned, chunked, overlapped.  The caller's (pageable) array is copied chunk by chunk into
// one of two pinned staging buffers by a few host threads, each chunk goes to the device with an asynchronous
// DMA and is converted / clipped / padded into the engine's layout by a kernel on the engine's stream, while the
// host threads already fill the other buffer.  Element types: the reference hands float64 (after
// _setup_adata, signature_nmf.py:269-281); raw count matrices may come as float32 / int32 / int64 / uint16 and are
// converted on the device, which also cuts the bytes that cross PCIe.
constexpr size_t STAGE_BYTES = (size_t)32 << 20;
constexpr int STAGE_THREADS = 8;

static size_t dtype_size(int dtype) {
    switch (dtype) {
        case SALNMF_F64: return 8;
        case SALNMF_F32: return 4;
        case SALNMF_I32: return 4;
        case SALNMF_I64: return 8;
        case SALNMF_U16: return 2;
        default: return 0;
    }
}

// Pinned staging buffers are kept for the life of the process and handed from engine to engine: pinning 32 MB costs
// 5-80 ms per hipHostMalloc on this platform (rocprofv3 --hip-trace of tools/time_init.py), more than the transfer it serves,
// and a fit() creates a fresh engine.  At most four are cached (128 MB); never freed at exit (the runtime may be gone).
// The same for the small per-engine block (scalars read back + the abort word: SMALL_PINNED_BYTES).
static std::mutex g_pinned_mutex;
static std::vector<void*> g_pinned_cache[2];  // [0] staging buffers (STAGE_BYTES), [1] small blocks

static hipError_t acquire_pinned(void** out, int small_block) {
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        auto& c = g_pinned_cache[small_block];
        if (!c.empty()) {
            *out = c.back();
            c.pop_back();
            return hipSuccess;
        }
    }
    // (portable: a cached block may be handed to an engine on another device of this process)
    if (!small_block) return hipHostMalloc(out, STAGE_BYTES, hipHostMallocPortable);
    // small blocks are carved out of one pinned slab of 16 (never returned to the runtime)
    void* slab = nullptr;
    const hipError_t rc = hipHostMalloc(&slab, 16 * SMALL_PINNED_BYTES, hipHostMallocPortable);
    if (rc != hipSuccess) return rc;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (int i = 1; i < 16; ++i) g_pinned_cache[1].push_back(static_cast<char*>(slab) + (size_t)i * SMALL_PINNED_BYTES);
    *out = slab;
    return hipSuccess;
}

static void release_pinned(void* p, int small_block) {
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        auto& c = g_pinned_cache[small_block];
        if (small_block || c.size() < 4u) {  // (a small block is part of a slab: it always goes back to the list)
            c.push_back(p);
            return;
        }
    }
    (void)hipHostFree(p);
}

// the pinned halves of the staging pairs are borrowed for the duration of one upload / download (both return with the
// stream idle), so that engines that are alive at the same time share them
static void release_staging(salnmf_engine* e) {
    for (int i = 0; i < 2; ++i)
        if (e->stage_host[i]) {
            release_pinned(e->stage_host[i], 0);
            e->stage_host[i] = nullptr;
        }
}

static int ensure_staging(salnmf_engine* e) {
    for (int i = 0; i < 2; ++i) {
        if (!e->stage_host[i]) HIPCK(acquire_pinned(&e->stage_host[i], 0));
        if (!e->stage_dev[i]) HIPCK(hipMalloc(&e->stage_dev[i], STAGE_BYTES));
        if (!e->stage_done[i]) HIPCK(hipEventCreateWithFlags(&e->stage_done[i], hipEventDisableTiming));
    }
    return 0;
}

static void parallel_copy(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)4 << 20)) {
        memcpy(dst, src, bytes);
        return;
    }
    std::thread workers[STAGE_THREADS - 1];
    const size_t piece = (bytes / STAGE_THREADS + 63) & ~(size_t)63;
    for (int t = 1; t < STAGE_THREADS; ++t) {
        const size_t off = std::min(bytes, piece * t), len = std::min(bytes, piece * (t + 1)) - off;
        workers[t - 1] = std::thread([=] { if (len) memcpy((char*)dst + off, (const char*)src + off, len); });
    }
    memcpy(dst, src, std::min(bytes, piece));
    for (auto& w : workers) w.join();
}

// host compact [N][cols] of element type `dtype` -> device padded double [Np][ld]
//   nb > 1 (X of an engine with more than 96 features): the rows are scattered into nb blocks of 96 columns, block b at
//   dst + b * Np * 96 (pad_rows_blocked_kernel); ld, fill_cols and fill_rows are then 96, 0, 0
static int upload_rows_staged(salnmf_engine* e, double* dst, const void* src, int dtype, int cols, int ld, double fill_cols, double fill_rows,
                              double clip_lo, int nb = 1, int bw = VMAX) {
    if (!e || !src) return fail("null argument");
    const size_t esz = dtype_size(dtype);
    if (!esz) return fail("unknown element type %d", dtype);
    CK(enter(e));
    CK(ensure_staging(e));
    const size_t row_bytes = (size_t)cols * esz;
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / row_bytes));
    int slot = 0;
    for (int64_t r0 = 0; r0 < e->N; r0 += chunk_rows, slot ^= 1) {
        const int64_t rows = std::min<int64_t>(chunk_rows, e->N - r0);
        const size_t bytes = (size_t)rows * row_bytes;
        HIPCK(hipEventSynchronize(e->stage_done[slot]));  // the chunk that used this slot two rounds ago has landed
        parallel_copy(e->stage_host[slot], (const char*)src + (size_t)r0 * row_bytes, bytes);
        HIPCK(hipMemcpyAsync(e->stage_dev[slot], e->stage_host[slot], bytes, hipMemcpyHostToDevice, e->stream));
        double* out = dst + r0 * ld;
        if (nb > 1) {
            const int g = (int)std::min<int64_t>(2048, (rows * nb * ld + 255) / 256);
            const int64_t bs = (int64_t)e->Np * ld;  // (blocks of bw source columns, stored at row stride ld)
            switch (dtype) {
                case SALNMF_F64: hipLaunchKernelGGL(pad_rows_blocked_kernel<double>, dim3(g), dim3(256), 0, e->stream, out, static_cast<const double*>(e->stage_dev[slot]), rows, cols, nb, bs, clip_lo, bw, ld); break;
                case SALNMF_F32: hipLaunchKernelGGL(pad_rows_blocked_kernel<float>, dim3(g), dim3(256), 0, e->stream, out, static_cast<const float*>(e->stage_dev[slot]), rows, cols, nb, bs, clip_lo, bw, ld); break;
                case SALNMF_I32: hipLaunchKernelGGL(pad_rows_blocked_kernel<int32_t>, dim3(g), dim3(256), 0, e->stream, out, static_cast<const int32_t*>(e->stage_dev[slot]), rows, cols, nb, bs, clip_lo, bw, ld); break;
                case SALNMF_I64: hipLaunchKernelGGL(pad_rows_blocked_kernel<int64_t>, dim3(g), dim3(256), 0, e->stream, out, static_cast<const int64_t*>(e->stage_dev[slot]), rows, cols, nb, bs, clip_lo, bw, ld); break;
                default: hipLaunchKernelGGL(pad_rows_blocked_kernel<uint16_t>, dim3(g), dim3(256), 0, e->stream, out, static_cast<const uint16_t*>(e->stage_dev[slot]), rows, cols, nb, bs, clip_lo, bw, ld); break;
            }
        } else
        switch (dtype) {
            case SALNMF_F64: launch_pad_rows<double>(e, out, e->stage_dev[slot], rows, cols, ld, fill_cols, clip_lo); break;
            case SALNMF_F32: launch_pad_rows<float>(e, out, e->stage_dev[slot], rows, cols, ld, fill_cols, clip_lo); break;
            case SALNMF_I32: launch_pad_rows<int32_t>(e, out, e->stage_dev[slot], rows, cols, ld, fill_cols, clip_lo); break;
            case SALNMF_I64: launch_pad_rows<int64_t>(e, out, e->stage_dev[slot], rows, cols, ld, fill_cols, clip_lo); break;
            default: launch_pad_rows<uint16_t>(e, out, e->stage_dev[slot], rows, cols, ld, fill_cols, clip_lo); break;
        }
        HIPCK(hipGetLastError());
        HIPCK(hipEventRecord(e->stage_done[slot], e->stream));
    }
    if (e->Np > e->N) {
        for (int b = 0; b < nb; ++b) {
            // (blocks: the pad rows carry fill_rows in the block's own columns, fill_cols beyond -- both 0 for X)
            hipLaunchKernelGGL(fill_rows_kernel, dim3(4), dim3(256), 0, e->stream, dst + (size_t)b * e->Np * ld, e->N, e->Np, ld,
                               nb > 1 ? std::min(bw, cols - b * bw) : cols, fill_rows, fill_cols);
            HIPCK(hipGetLastError());
        }
    }
    HIPCK(hipStreamSynchronize(e->stream));  // the caller's array is free again (and so are the staging buffers)
    release_staging(e);
    return 0;
}

static int upload_padded(salnmf_engine* e, double* dst, const double* src, int cols, int ld, double fill_cols,
                         double fill_rows, double clip_lo) {
    return upload_rows_staged(e, dst, src, SALNMF_F64, cols, ld, fill_cols, fill_rows, clip_lo);
}

// device padded [.][ld] -> host compact [N][cols]: the mirror image of the ingest -- per 32 MB chunk an unpad kernel
// into a device staging buffer, an asynchronous DMA into a pinned buffer, and a multi-threaded host copy into the
// caller's (pageable) array while the next chunk is on its way
//   nb > 1: src is [nb][Np][ld], chunk b holding columns bw b .. bw b + bw - 1 (H with n_signatures > 64)
static int download_padded(salnmf_engine* e, double* dst, const double* src, int cols, int ld, int nb = 1, int bw = 0) {
    if (!e || !dst) return fail("null argument");
    CK(enter(e));
    CK(ensure_staging(e));
    const size_t row_bytes = (size_t)cols * sizeof(double);
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / row_bytes));
    const int64_t nchunks = (e->N + chunk_rows - 1) / chunk_rows;
    auto rows_of = [&](int64_t c) { return std::min<int64_t>(chunk_rows, e->N - c * chunk_rows); };
    auto issue = [&](int64_t c) -> int {
        const int slot = (int)(c & 1);
        const int64_t rows = rows_of(c);
        if (nb > 1)
            hipLaunchKernelGGL(unpad_blocked_kernel, dim3(2048), dim3(256), 0, e->stream, static_cast<double*>(e->stage_dev[slot]),
                               src + c * chunk_rows * ld, rows, cols, bw, ld, (int64_t)e->Np * ld);
        else
            hipLaunchKernelGGL(unpad_kernel, dim3(2048), dim3(256), 0, e->stream, static_cast<double*>(e->stage_dev[slot]),
                               src + c * chunk_rows * ld, rows, cols, ld);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(e->stage_host[slot], e->stage_dev[slot], (size_t)rows * row_bytes, hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipEventRecord(e->stage_done[slot], e->stream));
        return 0;
    };
    if (nchunks > 0) CK(issue(0));
    for (int64_t c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) CK(issue(c + 1));  // the other slot: its previous content was consumed in the last round
        HIPCK(hipEventSynchronize(e->stage_done[c & 1]));
        parallel_copy((char*)dst + (size_t)c * chunk_rows * row_bytes, e->stage_host[c & 1], (size_t)rows_of(c) * row_bytes);
    }
    HIPCK(hipStreamSynchronize(e->stream));
    release_staging(e);
    return check_abort(e);
}

int salnmf_upload_X_typed(salnmf_engine* e, const void* X, int dtype, int clip) {
    if (!e) return fail("null engine");
    e->xrowsum_valid = e->lgam_valid = e->x32_valid = e->xlx_valid = false;
    // pad rows / columns are exactly 0 (never clipped): they must contribute X/P = 0
    return upload_rows_staged(e, e->X, X, dtype, e->V, VMAX, 0.0, 0.0, clip ? kEps : 0.0, e->NB);
}
int salnmf_upload_X(salnmf_engine* e, const double* X, int clip) { return salnmf_upload_X_typed(e, X, SALNMF_F64, clip); }
int salnmf_upload_W(salnmf_engine* e, const double* W) {
    if (!e) return fail("null engine");
    CK(enter(e));
    e->keep_valid = false;
    return upload(e, e->W, W, (size_t)e->K * e->V);
}
int salnmf_upload_H(salnmf_engine* e, const double* H) {
    if (e) CK(enter(e));
    if (e) e->h_pending = e->keep_valid = false;
    // pad columns 0, pad rows 1: finite, and positive in the rows so that P > 0 there
    if (e && e->NC > 1) return upload_rows_staged(e, e->H, H, SALNMF_F64, e->K, e->KP, 0.0, 1.0, 0.0, e->NC, e->kc[0].K);
    return upload_padded(e, e ? e->H : nullptr, H, e ? e->K : 0, e ? e->KP : 0, 0.0, 1.0, 0.0);
}

int salnmf_set_H_scale(salnmf_engine* e, const double* scale) {
    if (!e || !scale) return fail("null argument");
    if (e->NC > 1) return single_block(e, "a lazily applied exposure scale");
    CK(enter(e));
    CK(flush_H_scale(e));  // (an earlier pending rescale is applied first)
    std::vector<double> cs((size_t)e->KP, 1.0);
    for (int k = 0; k < e->K; ++k) cs[(size_t)k] = scale[k];
    HIPCK(hipMemcpyAsync(e->cs, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));  // (cs is a local)
    e->h_pending = true;  // every reader of H applies clip(H * scale) on the fly until a pass rewrites H in full
    return 0;
}

int salnmf_set_weights(salnmf_engine* e, const double* weights_kl, const double* weights_lhalf) {
    if (!e) return fail("null engine");
    CK(enter(e));
    HIPCK(hipStreamSynchronize(e->stream));
    auto set = [&](double*& dev, const double* host, double filler) -> int {
        if (!host) {
            if (dev) HIPCK(hipFree(dev));
            dev = nullptr;
            return 0;
        }
        if (!dev) HIPCK(hipMalloc(&dev, (size_t)e->Np * sizeof(double)));
        return upload_padded(e, dev, host, 1, 1, filler, filler, 0.0);
    };
    CK(set(e->wkl, weights_kl, 1.0));
    CK(set(e->wlh, weights_lhalf, 0.0));
    return 0;
}

static int download(salnmf_engine* e, double* dst, const double* src, size_t n) {
    if (!e || !dst) return fail("null argument");
    CK(enter(e));
    HIPCK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    return check_abort(e);
}
int salnmf_download_W(salnmf_engine* e, double* W) {
    if (!e) return fail("null engine");
    return download(e, W, e->W, (size_t)e->K * e->V);
}
int salnmf_download_H(salnmf_engine* e, double* H) {
    if (!e) return fail("null engine");
    CK(enter(e));
    CK(flush_H_scale(e));
    return download_padded(e, H, e->H, e->K, e->KP, e->NC, e->kc[0].K);
}

// n joint steps in one persistent launch (fused_kernel<..., PERSIST>): every workgroup must be resident, which the
// geometry guarantees (grid <= number of CUs, one workgroup per CU by its LDS footprint)
static int kl_steps_persistent(salnmf_engine* e, int n, int n_given) {
    FusedParams p = fused_params(e);
    p.nsteps = n;
    p.n_given = n_given;
    p.Wmut = e->W;
    p.G = e->red;
    p.sync = e->psync;
    p.abort_host = e->pabort;
    HIPCK(hipMemsetAsync(e->psync, 0, SYNC_WORDS * sizeof(unsigned), e->stream));
    const FusedSel sel{e->KS, e->KTM, e->KR, true, true, false, false, true};
    if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr))
        return fail("this build carries no persistent kernel for KS=%d KTM=%d KR=%d (build with SALNMF_WITH_PERSISTENT)", e->KS, e->KTM, e->KR);
    HIPCK(hipGetLastError());
    return 0;
}

// a persistent launch whose waits gave up leaves the resident state half updated: say so at the next sync point
static int check_abort(salnmf_engine* e) {
    if (e->pabort && *e->pabort == 2u) {
        return fail("a peer-to-peer exchange gave up waiting for another rank (salnmf_p2p_kernels.h: 20 s); the engine's W and H are "
                    "invalid -- did every rank issue the same calls?");
    }
    if (e->pabort && *e->pabort) {
        return fail("a wait inside the persistent KL kernel gave up (its workgroups were not all resident: is another process "
                    "using this GPU?); the engine's W and H are invalid -- upload them again, and call salnmf_set_persistent(e, 0)");
    }
    return 0;
}

// ---- n_features > 96: the passes run once per 96-feature block of X and W (include/salnmf.h: limits)
// H half (update_H, _utils_klnmf.py:220-278; the H half of update_WH, :343-361) from (W, H) into Hout: U = R W^T summed
// over the blocks' launches through Uacc, the last block's launch updates H
static int blocked_update_H(salnmf_engine* e, double* Hout, double hfloor = kEps, bool weighted = true) {
    for (int b = 0; b < e->NB; ++b) {
        FusedParams p = fused_params(e);
        to_block(e, p, b);
        p.Hout = Hout;
        p.hfloor = hfloor;
        if (!weighted) p.wkl = p.wlh = nullptr;
        p.Uacc = e->Uacc;
        p.ublock = b == 0 ? 1 : (b == e->NB - 1 ? 3 : 2);
        CK(weight_arrays(e, p));  // (the BLOCKED instantiation is the weighted-capable one)
        const FusedSel sel{e->KS, e->KTM, e->KR, false, true, false, true, false, true};
        if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", e->KS, e->KTM, e->KR);
        HIPCK(hipGetLastError());
    }
    return 0;
}
// numerator (w_kl * X / (W H)) @ H^T of every block from (W, H) -> Gblk (compact [K][width] per block)
static int blocked_numerators(salnmf_engine* e, bool weighted = true) {
    for (int b = 0; b < e->NB; ++b) {
        FusedParams p = fused_params(e);
        to_block(e, p, b);
        if (!weighted) p.wkl = p.wlh = nullptr;
        CK((launch_fused<true, false, false>(e, p)));
        TailParams t = tail_params(e, e->grid, e->Gblk + (size_t)b * e->K * VMAX, 0, 0, 0, false);
        t.V = block_width(e, b);
        hipLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, t);
        HIPCK(hipGetLastError());
    }
    // sample shards: the blocks' compact numerators lie back to back (K * V doubles in all): one exchange for all of them
    if (sharded(e)) CK(allreduce(e, e->Gblk, (size_t)e->K * e->V));
    return 0;
}
// both halves in ONE pass per feature block (round 5): the block's numerator (-> Gblk, as blocked_numerators) and its share of
// U = R W^T, accumulated over the blocks; the last block's pass writes clip(H * U) to Hout.  Every block's numerator is formed
// from the OLD H -- only the last pass rewrites it, tile by tile behind that tile's own numerator -- so Hout may be H itself.
// P = H W[:, block] is formed once per block instead of twice (96 x 100 000 x 3 blocks: 335 -> 215 us per joint step).
static int blocked_joint_passes(salnmf_engine* e, double* Hout, double hfloor = kEps, bool weighted = true) {
    for (int b = 0; b < e->NB; ++b) {
        FusedParams p = fused_params(e);
        to_block(e, p, b);
        p.Hout = Hout;
        p.hfloor = hfloor;
        if (!weighted) p.wkl = p.wlh = nullptr;
        p.Uacc = e->Uacc;
        p.ublock = b == 0 ? 1 : (b == e->NB - 1 ? 3 : 2);
        CK(weight_arrays(e, p));
        const FusedSel sel{e->KS, e->KTM, e->KR, true, true, false, true, false, true};
        if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", e->KS, e->KTM, e->KR);
        HIPCK(hipGetLastError());
        TailParams t = tail_params(e, e->grid, e->Gblk + (size_t)b * e->K * VMAX, 0, 0, 0, false);
        t.V = block_width(e, b);
        hipLaunchKernelGGL(tail_kernel, dim3(e->K), dim3(TAIL_BLOCK), 0, e->stream, t);
        HIPCK(hipGetLastError());
    }
    if (sharded(e)) CK(allreduce(e, e->Gblk, (size_t)e->K * e->V));
    return 0;
}
static int blocked_finish_W(salnmf_engine* e, int n_given, int clip_mode) {
    hipLaunchKernelGGL(w_finish_blocked_kernel, dim3(e->K), dim3(256), 0, e->stream, e->Gblk, e->red, e->W, e->W, e->V, e->K, n_given, clip_mode);
    HIPCK(hipGetLastError());
    return 0;
}
// one joint step (update_WH, _utils_klnmf.py:281-361): both halves from the OLD (W, H), one pass per feature block
static int blocked_kl_step_once(salnmf_engine* e, int n_given) {
    if (n_given >= e->K) {  // W untouched (:330-331): in place
        CK(blocked_update_H(e, e->H));
        e->h_pending = false;
        return 0;
    }
    CK(blocked_joint_passes(e, e->H));  // (both halves read the old H, a pending rescale included)
    CK(blocked_finish_W(e, n_given, SALNMF_CLIP_ALL));
    e->h_pending = false;  // the new H was written in full
    return 0;
}


// ---- n_signatures > 64: the passes run per chunk of <= 64 signatures (include/salnmf.h: limits).  P = H W is a sum over
// the chunks, so it is formed FIRST, by a chain of forward launches through e->PR (each adds its chunk's product; the
// last one turns the sum into what is needed: the ratio X / P, the divergence, the per-sample divergences or P itself);
// the update passes then run on the given ratio, once per chunk, each with the geometry of its chunk's size.
//   W, hscale (MvNMF line-search trials): the signature matrix instead of e->W, and H read as clip(H * hscale[k]) -- both
//   compact over all signatures
//   b: the feature block (engines with more than 96 features AND more than 64 signatures run the chain block by block)
static FwdParams chunk_fwd_params(salnmf_engine* e, const salnmf_engine::Chunk& c, int ci, const double* W = nullptr, const double* hscale = nullptr,
                                  int b = 0) {
    FwdParams p{};
    p.X = e->X + (size_t)b * e->Np * VMAX;
    p.H = e->H + (size_t)ci * e->Np * e->KP;
    p.W = (W ? W : e->W) + (size_t)c.k0 * e->V + (size_t)VMAX * b;
    p.hscale = hscale ? hscale + c.k0 : nullptr;
    p.xlx = e->xlx ? e->xlx + (size_t)b * e->Np * 16 : nullptr;
    p.N = e->N;
    p.V = block_width(e, b);
    p.ldw = e->V;
    p.K = c.K;
    p.ntiles = e->ntiles;
    return p;
}
// the chain: chunks 0 .. NC-2 accumulate into e->PR (mode 2), the last chunk runs `last_mode` with `last` as its template
// (out, weights) on top of the accumulated product
static int chunk_chain(salnmf_engine* e, int last_mode, const FwdParams& last, const double* W = nullptr, const double* hscale = nullptr, int b = 0) {
    for (int ci = 0; ci < e->NC; ++ci) {
        const auto& c = e->kc[(size_t)ci];
        FwdParams p = chunk_fwd_params(e, c, ci, W, hscale, b);
        p.pin = ci == 0 ? nullptr : e->PR;
        const bool is_last = ci == e->NC - 1;
        if (is_last) {
            p.wkl = last.wkl;
            p.wlh = last.wlh;
            p.out = last.out;
        } else {
            p.out = e->PR;
        }
        if (launch_forward_inst(c.KS, FWD_PIN + (is_last ? last_mode : 2), p, e->fgrid, e->stream, nullptr, nullptr))
            return fail("no forward instantiation for KS=%d", c.KS);
        HIPCK(hipGetLastError());
    }
    return 0;
}
static int chunk_ratio(salnmf_engine* e) {  // e->PR = X / (H W)
    FwdParams last{};
    last.out = e->PR;
    return chunk_chain(e, 4, last);
}
// the update passes of every chunk on the ratio in e->PR: H half (do_u) into the chunk's columns of H, in place; W half
// (do_g): numerator slabs, reduced and applied to the chunk's rows of W by the ordinary tail
//   weighted = false: MvNMF's passes (no sample weights: mvnmf.py:56,162-165); g_only: the numerator rows are reduced into
//   e->red and W is left alone (the MvNMF W step takes its own root from them)
static int chunk_passes(salnmf_engine* e, bool do_g, bool do_u, int n_given, int clip_mode, bool weighted = true, bool g_only = false) {
    const bool shard = sharded(e) && do_g && n_given < e->K;
    for (int ci = 0; ci < e->NC; ++ci) {
        const auto& c = e->kc[(size_t)ci];
        const int given = std::max(0, std::min(c.K, n_given - c.k0));  // given rows inside this chunk
        const bool g = do_g && given < c.K;
        if (!g && !do_u) continue;
        FusedParams p = fused_params(e);
        p.X = e->PR;
        p.H = p.Hout = e->H + (size_t)ci * e->Np * e->KP;
        p.W = e->W + (size_t)c.k0 * e->V;
        p.K = c.K;
        p.hscale = nullptr;
        if (!weighted) p.wkl = p.wlh = nullptr;
        CK(weight_arrays(e, p));
        FusedSel sel{c.KS, c.KTM, c.KR, g, do_u, false, true, false, false};
        sel.RGIVEN = true;
        if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", c.KS, c.KTM, c.KR);
        HIPCK(hipGetLastError());
        if (g) {
            TailParams t = tail_params(e, e->grid, e->red + (size_t)c.k0 * e->V, given, clip_mode, (g_only || shard) ? 0 : 1, false);
            t.W = t.Wout = e->W + (size_t)c.k0 * e->V;
            t.K = c.K;
            hipLaunchKernelGGL(tail_kernel, dim3(c.K), dim3(TAIL_BLOCK), 0, e->stream, t);
            HIPCK(hipGetLastError());
        }
    }
    if (shard) {
        // the chunks' numerator rows lie in e->red as one K x V matrix: one exchange, then the W update of every row (the
        // given rows' numerators were not formed: they are not read either)
        CK(allreduce(e, e->red, (size_t)e->K * e->V));
        if (!g_only) CK(launch_tail(e, 0, e->red, n_given, clip_mode, 1));
    }
    return 0;
}
// one joint step (update_WH, _utils_klnmf.py:281-361): both halves use the ratio of the OLD state, so each chunk's pass
// updates its columns of H and its rows of W in place
static int chunked_kl_step_once(salnmf_engine* e, int n_given) {
    CK(chunk_ratio(e));
    return chunk_passes(e, n_given < e->K, true, n_given, SALNMF_CLIP_ALL);
}

// ---- n_features > 96 AND n_signatures > 64 (round 5): the two decompositions together.  Per feature block b the chain over
// the chunks forms that block's ratio R_b = X_b / (H W_b) in e->PR; on it every chunk runs its numerator pass (G of the pair
// (chunk, block) -> its rows of the block's compact numerator in Gblk) and its share U += R_b W_(chunk, b)^T of the update_H
// product, accumulated over the blocks through the chunk's part of Uacc; the last block's pass updates the chunk's columns of
// H.  The chains read the OLD H of all chunks: a joint step writes the new H to the second buffer (Hout), an update_H alone
// may go in place (the last block's chain is done before its passes).  W afterwards as on feature blocks (blocked_finish_W).
static inline bool grid_split(const salnmf_engine* e) { return e->NB > 1 && e->NC > 1; }
static int grid_passes(salnmf_engine* e, bool do_g, bool do_u, int n_given, double* Hout, bool weighted = true) {
    const size_t hc = (size_t)e->Np * e->KP;
    bool any_g = false;
    for (int b = 0; b < e->NB; ++b) {
        FwdParams last{};
        last.out = e->PR;
        CK(chunk_chain(e, 4, last, nullptr, nullptr, b));
        const int vb = block_width(e, b);
        for (int ci = 0; ci < e->NC; ++ci) {
            const auto& c = e->kc[(size_t)ci];
            const int given = std::max(0, std::min(c.K, n_given - c.k0));
            FusedParams p = fused_params(e);
            p.X = e->PR;
            p.V = vb;
            p.ldw = e->V;
            p.W = e->W + (size_t)c.k0 * e->V + (size_t)VMAX * b;
            p.H = p.Hout = e->H + (size_t)ci * hc;
            p.K = c.K;
            p.hscale = nullptr;
            if (!weighted) p.wkl = p.wlh = nullptr;
            CK(weight_arrays(e, p));
            const bool g = do_g && given < c.K;
            if (g && do_u) {  // both halves of the pair in one pass over the block's ratio
                p.Hout = Hout + (size_t)ci * hc;
                p.Uacc = e->Uacc + (size_t)ci * hc;
                p.ublock = b == 0 ? 1 : (b == e->NB - 1 ? 3 : 2);
                FusedSel sel{c.KS, c.KTM, c.KR, true, true, false, true, false, true};
                sel.RGIVEN = true;
                if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", c.KS, c.KTM, c.KR);
                HIPCK(hipGetLastError());
                TailParams t = tail_params(e, e->grid, e->Gblk + (size_t)b * e->K * VMAX + (size_t)c.k0 * vb, 0, 0, 0, false);
                t.V = vb;
                t.K = c.K;
                hipLaunchKernelGGL(tail_kernel, dim3(c.K), dim3(TAIL_BLOCK), 0, e->stream, t);
                HIPCK(hipGetLastError());
                any_g = true;
                continue;
            }
            if (g) {
                FusedSel sel{c.KS, c.KTM, c.KR, true, false, false, true, false, false};
                sel.RGIVEN = true;
                if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", c.KS, c.KTM, c.KR);
                HIPCK(hipGetLastError());
                TailParams t = tail_params(e, e->grid, e->Gblk + (size_t)b * e->K * VMAX + (size_t)c.k0 * vb, 0, 0, 0, false);
                t.V = vb;
                t.K = c.K;
                hipLaunchKernelGGL(tail_kernel, dim3(c.K), dim3(TAIL_BLOCK), 0, e->stream, t);
                HIPCK(hipGetLastError());
                any_g = true;
            }
            if (do_u) {
                p.Hout = Hout + (size_t)ci * hc;
                p.Uacc = e->Uacc + (size_t)ci * hc;
                p.ublock = b == 0 ? 1 : (b == e->NB - 1 ? 3 : 2);
                FusedSel sel{c.KS, c.KTM, c.KR, false, true, false, true, false, true};
                sel.RGIVEN = true;
                if (launch_fused_inst(sel, p, e->grid, e->stream, nullptr, nullptr)) return fail("no kernel instantiation for KS=%d KTM=%d KR=%d", c.KS, c.KTM, c.KR);
                HIPCK(hipGetLastError());
            }
        }
    }
    if (any_g && sharded(e)) CK(allreduce(e, e->Gblk, (size_t)e->K * e->V));
    if (do_u) e->h_pending = false;
    return 0;
}
static int grid_kl_step_once(salnmf_engine* e, int n_given) {
    if (n_given >= e->K) return grid_passes(e, false, true, n_given, e->H);  // W untouched (_utils_klnmf.py:330-331)
    CK(ensure_halt(e));
    CK(grid_passes(e, true, true, n_given, e->Halt));
    CK(blocked_finish_W(e, n_given, SALNMF_CLIP_ALL));
    std::swap(e->H, e->Halt);
    return 0;
}

int salnmf_set_lockstep(salnmf_engine* e, int on) {
    if (!e) return fail("null engine");
    e->lockstep = on != 0;
    return 0;
}

int salnmf_set_w_dma(salnmf_engine* e, int on) {
    if (!e) return fail("null engine");
    e->w_dma = on != 0;
    return 0;
}

int salnmf_set_mv_queued(salnmf_engine* e, int on) {
    if (!e) return fail("null engine");
    CK(enter(e));  // (an engine left ahead steps back first: the two forms speculate differently)
    e->mv_queued = on != 0;
    return 0;
}

int salnmf_set_small_cohort_tiles(salnmf_engine* e, int max_tiles) {
    if (!e) return fail("null engine");
    if (max_tiles < 0) return fail("max_tiles must not be negative (0 turns the one-workgroup kernel off)");
    e->small_max_tiles = max_tiles;
    return 0;
}

int salnmf_set_batched_sample_solves(salnmf_engine* e, int on) {
    if (!e) return fail("null engine");
    e->batched_samples = on != 0;
    return 0;
}

int salnmf_set_persistent(salnmf_engine* e, int on) {
    if (!e) return fail("null engine");
    if (on && split(e)) return single_block(e, "the persistent kernel");
    if (on && !built_with_persistent()) return fail("this build carries no persistent kernel (compile with -DSALNMF_WITH_PERSISTENT)");
    e->persistent = on != 0;
    return 0;
}

// n joint steps on the fp32 matrix cores (opt-in, salnmf_kernels_f32.h): shadow copies in, steps, H back out
static int kl_steps_f32(salnmf_engine* e, int n_steps, int n_given) {
    CK(flush_H_scale(e));
    const int64_t nx = e->Np * VMAX, nh = e->Np * e->KP;
    if (!e->X32) {
        HIPCK(hipMalloc(&e->X32, (size_t)nx * sizeof(float)));
        e->x32_valid = false;
    }
    if (!e->H32) HIPCK(hipMalloc(&e->H32, (size_t)nh * sizeof(float)));
    if (!e->x32_valid) {
        hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(2048), dim3(256), 0, e->stream, e->X32, e->X, nx);
        HIPCK(hipGetLastError());
        e->x32_valid = true;
    }
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(2048), dim3(256), 0, e->stream, e->H32, e->H, nh);
    HIPCK(hipGetLastError());
    Fused32Params p;
    p.X = e->X32;
    p.H = e->H32;
    p.Gpart = e->Gpart;
    p.N = e->N;
    p.ntiles = e->ntiles;
    p.V = e->V;
    p.K = e->K;
    p.hfloor = (float)kEps;
    for (int i = 0; i < n_steps; ++i) {
        p.W = e->W;
        if (launch_fused_f32_inst(e->KS, p, e->grid, e->stream)) return fail("no kernel instantiation for KS=%d", e->KS);
        HIPCK(hipGetLastError());
        if (n_given < e->K) {  // _utils_klnmf.py:330-331: W untouched when every signature is given
            if (sharded(e))
                CK(sharded_tail(e, n_given, SALNMF_CLIP_ALL));
            else
                CK(launch_tail(e, e->grid, e->red, n_given, SALNMF_CLIP_ALL, 1));
        }
    }
    hipLaunchKernelGGL(cvt_f32_f64_kernel, dim3(2048), dim3(256), 0, e->stream, e->H, e->H32, nh);
    HIPCK(hipGetLastError());
    return 0;
}

int salnmf_set_precision(salnmf_engine* e, int precision) {
    if (e && split(e)) return single_block(e, "the fp32 fast mode");
    if (!e) return fail("null engine");
    if (precision != SALNMF_PRECISION_F64 && precision != SALNMF_PRECISION_F32_FAST) return fail("unknown precision %d", precision);
    e->fast32 = precision == SALNMF_PRECISION_F32_FAST;
    return 0;
}

// small cohorts: one workgroup, all steps of a call in one launch (salnmf_small.hip); unweighted fp64 steps of an unsharded
// engine with at most 16 signatures, at least one of them free
static bool small_path(const salnmf_engine* e, int n_given) {
    return e->NB == 1 && e->NC == 1 && !e->fast32 && !e->persistent && !sharded(e) && !e->wkl && !e->wlh && e->K <= 16 && n_given < e->K &&
           e->ntiles <= std::min(e->small_max_tiles, SMALL_MAX_TILES);
}

int salnmf_kl_step(salnmf_engine* e, int n_steps, int n_given) {
    if (!e) return fail("null engine");
    if (n_given < 0 || n_given > e->K) return fail("n_given out of range");
    CK(enter(e));
    if (grid_split(e)) {
        e->keep_valid = false;  // (the joint step uses the second H buffer itself)
        for (int i = 0; i < n_steps; ++i) CK(grid_kl_step_once(e, n_given));
        return 0;
    }
    if (e->NB > 1) {
        e->keep_valid = false;  // (the joint step uses the second H buffer itself)
        for (int i = 0; i < n_steps; ++i) CK(blocked_kl_step_once(e, n_given));
        return 0;
    }
    if (e->NC > 1) {
        for (int i = 0; i < n_steps; ++i) CK(chunked_kl_step_once(e, n_given));
        return 0;
    }
    if (e->fast32 && n_steps > 0) {
        if (e->wkl || e->wlh) return fail("the fp32 fast mode has no weighted step: clear the weights or set SALNMF_PRECISION_F64");
        return kl_steps_f32(e, n_steps, n_given);
    }
    int i = 0;
    if (small_path(e, n_given) && n_steps > 0) {
        CK(flush_H_scale(e));  // (after an MvNMF step) the kernel reads H as it is
        SmallParams sp{e->X, e->H, e->W, e->W, e->red, e->V, e->K, (int)e->ntiles, 0, n_given, SALNMF_CLIP_ALL};
        constexpr int kMaxPerLaunch = 4096;  // bounds one launch to tens of milliseconds
        while (i < n_steps) {
            sp.nsteps = std::min(kMaxPerLaunch, n_steps - i);
            if (launch_small_kl_steps(e->KS, sp, e->stream)) return fail("no small-cohort kernel for KS=%d", e->KS);
            HIPCK(hipGetLastError());
            i += sp.nsteps;
        }
        return 0;
    }
    if (e->persistent && !sharded(e) && !e->wkl && !e->wlh && n_given < e->K && n_steps >= 2) {
        CK(flush_H_scale(e));  // (after an MvNMF step) the persistent kernel reads H as it is
        constexpr int kMaxPerLaunch = 64;  // bounds one launch to a few milliseconds
        while (n_steps - i >= 2) {
            const int n = std::min(kMaxPerLaunch, n_steps - i);
            CK(kl_steps_persistent(e, n, n_given));
            i += n;
        }
    }
    for (; i < n_steps; ++i) CK(kl_step_once(e, n_given, nullptr));
    return 0;
}

int salnmf_kl_step_keep(salnmf_engine* e, int n_steps, int n_given) {
    if (!e) return fail("null engine");
    if (n_given < 0 || n_given > e->K) return fail("n_given out of range");
    if (n_steps < 1) return fail("n_steps must be positive");
    CK(enter(e));
    if (e->NC == 1) CK(ensure_halt(e));
    if (!e->Wkeep) HIPCK(hipMalloc(&e->Wkeep, (size_t)e->K * e->V * sizeof(double)));
    e->keep_valid = false;
    CK(flush_H_scale(e));
    if (split(e)) {
        // feature blocks: the joint step exchanges H with the second buffer itself; signature chunks: the passes update H
        // in place.  The kept state is a copy
        if (!e->Hkeep) HIPCK(hipMalloc(&e->Hkeep, h_doubles(e) * sizeof(double)));
        HIPCK(hipMemcpyAsync(e->Hkeep, e->H, h_doubles(e) * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        HIPCK(hipMemcpyAsync(e->Wkeep, e->W, (size_t)e->K * e->V * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        CK(salnmf_kl_step(e, n_steps, n_given));
        e->keep_has_W = true;
        e->keep_valid = true;
        return 0;
    }
    if (e->fast32 || e->persistent || small_path(e, n_given)) {
        // the fp32 fast mode, the persistent kernel and the small-cohort kernel update the state in place: keep a copy instead
        HIPCK(hipMemcpyAsync(e->Halt, e->H, (size_t)e->Np * e->KP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        HIPCK(hipMemcpyAsync(e->Wkeep, e->W, (size_t)e->K * e->V * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        CK(salnmf_kl_step(e, n_steps, n_given));
        e->keep_has_W = true;  // H / W hold the new state, Halt / Wkeep the copy of the old one: the same roles as below
    } else {
        CK(kl_step_once(e, n_given, nullptr, true));
        for (int i = 1; i < n_steps; ++i) CK(kl_step_once(e, n_given, nullptr));
    }
    e->keep_valid = true;
    return 0;
}

int salnmf_kl_rollback(salnmf_engine* e) {
    if (!e) return fail("null engine");
    if (!e->keep_valid) return fail("no kept state: salnmf_kl_rollback undoes the last salnmf_kl_step_keep, once");
    // (the discarded steps may still be running: they write the buffers that become scratch, in stream order)
    if (split(e)) {
        std::swap(e->H, e->Hkeep);
        std::swap(e->W, e->Wkeep);
        e->keep_valid = false;
        return 0;
    }
    std::swap(e->H, e->Halt);
    if (e->keep_has_W) std::swap(e->W, e->Wkeep);
    e->h_pending = false;
    e->keep_valid = false;
    return 0;
}

int salnmf_update_H(salnmf_engine* e) {
    if (!e) return fail("null engine");
    CK(enter(e));
    if (grid_split(e)) return grid_passes(e, false, true, 0, e->H);
    if (e->NB > 1) {
        CK(blocked_update_H(e, e->H));
        e->h_pending = false;
        return 0;
    }
    if (e->NC > 1) {
        CK(chunk_ratio(e));
        return chunk_passes(e, false, true, 0, 0);
    }
    FusedParams p = fused_params(e);
    CK((launch_fused<false, true, false>(e, p)));
    e->h_pending = false;
    return 0;
}

int salnmf_kl_step_partial(salnmf_engine* e) {
    if (e && split(e)) return single_block(e, "the split step");
    if (!e) return fail("null engine");
    CK(enter(e));
    FusedParams p = fused_params(e);
    CK((launch_fused<true, true, false>(e, p)));
    e->h_pending = false;
    return launch_tail(e, e->grid, e->red, 0, 0, 0);
}

int salnmf_kl_step_finish(salnmf_engine* e, int n_given, int clip_mode) {
    if (e && split(e)) return single_block(e, "the split step");
    if (!e) return fail("null engine");
    CK(enter(e));
    if (n_given >= e->K) return 0;
    return launch_tail(e, 0, e->red, n_given, clip_mode, 1);
}

int salnmf_update_W(salnmf_engine* e, int n_given, int clip_mode) {
    if (!e) return fail("null engine");
    CK(enter(e));
    if (n_given >= e->K) return 0;  // _utils_klnmf.py:204-205
    if (grid_split(e)) {
        CK(grid_passes(e, true, false, n_given, nullptr));
        return blocked_finish_W(e, n_given, clip_mode);
    }
    if (e->NB > 1) {
        CK(blocked_numerators(e));
        return blocked_finish_W(e, n_given, clip_mode);
    }
    if (e->NC > 1) {
        CK(chunk_ratio(e));
        return chunk_passes(e, true, false, n_given, clip_mode);
    }
    FusedParams p = fused_params(e);
    CK((launch_fused<true, false, false>(e, p)));
    if (sharded(e)) return sharded_tail(e, n_given, clip_mode);
    return launch_tail(e, e->grid, e->red, n_given, clip_mode, 1);
}

static int fwd_params(salnmf_engine* e, FwdParams& p) {
    CK(ensure_xlogx(e));  // (mode 0 reads them; computed once per upload of X)
    p = FwdParams{};
    p.X = e->X;
    p.H = e->H;
    p.W = e->W;
    p.wkl = e->wkl;
    p.wlh = e->wlh;
    p.hscale = e->h_pending ? e->cs : nullptr;
    p.xlx = e->xlx;
    p.out = e->objpart;
    p.N = e->N;
    p.V = e->V;
    p.ldw = e->V;
    p.K = e->K;
    p.ntiles = e->ntiles;
    return 0;
}

// objective of (W, H[, hscale]) -> device scalar e->scal[slot] (all-reduced), no host sync
// the forward passes of one objective evaluation -> e->objpart[0 .. *nparts): per-workgroup partials, to be summed in order
//   direct != nullptr: where the engine runs ONE forward launch per objective (one feature block, one signature chunk), that
//   launch also adds its partials up (forward_kernel: sum_out) and stores the objective to *direct; bound to ev if given.
//   *nparts == 0 then says that no reduction is left to do
static int objective_partials(salnmf_engine* e, const double* W, const double* hscale, bool weighted, int grid, int* nparts,
                              double* direct = nullptr, hipEvent_t ev = nullptr) {
    FwdParams p;
    CK(fwd_params(e, p));
    p.W = W;
    if (hscale) p.hscale = hscale;  // else: the pending rescale, if any
    if (!weighted) {
        p.wkl = nullptr;
        p.wlh = nullptr;
    }
    const int fgrid = grid > 0 ? grid : e->fgrid;
    if (e->NB > 1 && e->NC > 1) {
        // a sum over the feature blocks of the chunk chain's divergence; the l-half penalty once (the last chunk's share by
        // block 0's last launch, the other chunks' by the small kernel)
        int n = 0;
        for (int b = 0; b < e->NB; ++b) {
            FwdParams last = p;
            if (b > 0) last.wlh = nullptr;
            last.out = e->objpart + n;
            CK(chunk_chain(e, 0, last, W, hscale, b));
            n += e->fgrid;
        }
        if (p.wlh) {
            for (int ci = 0; ci + 1 < e->NC; ++ci) {
                hipLaunchKernelGGL(lhalf_penalty_kernel, dim3(e->fgrid), dim3(256), 0, e->stream, e->H + (size_t)ci * e->Np * e->KP, p.wlh, e->N,
                                   e->kc[(size_t)ci].K, e->KP, e->objpart + n);
                HIPCK(hipGetLastError());
                n += e->fgrid;
            }
        }
        *nparts = n;
        return 0;
    }
    if (e->NC > 1) {
        // the chain over the signature chunks; the last launch evaluates the divergence (and its own chunk's share of the
        // l-half penalty, klnmf.py:75-79), the other chunks' shares come from a small kernel each
        FwdParams last = p;
        last.out = e->objpart;
        CK(chunk_chain(e, 0, last, W, hscale));
        int n = e->fgrid;
        if (p.wlh) {
            for (int ci = 0; ci + 1 < e->NC; ++ci) {
                hipLaunchKernelGGL(lhalf_penalty_kernel, dim3(e->fgrid), dim3(256), 0, e->stream, e->H + (size_t)ci * e->Np * e->KP, p.wlh, e->N,
                                   e->kc[(size_t)ci].K, e->KP, e->objpart + n);
                HIPCK(hipGetLastError());
                n += e->fgrid;
            }
        }
        *nparts = n;
        return 0;
    }
    if (e->NB > 1) {
        // the KL divergence is a sum over the features: one forward pass per feature block, each with the x-only
        // constants of its own features; the l-half penalty (klnmf.py:75-79) once
        for (int b = 0; b < e->NB; ++b) {
            FwdParams pb = p;
            pb.X = e->X + (size_t)b * e->Np * VMAX;
            pb.W = W + (size_t)VMAX * b;
            pb.V = block_width(e, b);
            pb.xlx = e->xlx + (size_t)b * e->Np * 16;
            if (b > 0) pb.wlh = nullptr;
            pb.out = e->objpart + (size_t)b * fgrid;
            CK(launch_forward<0>(e, pb, fgrid));
        }
    } else {
        if (direct) {
            p.sum_out = direct;
            p.sum_counter = e->klcnt + 1;  // (its own word: the MvNMF update_H pass's counter is word 0)
            CK(launch_forward<0>(e, p, fgrid, nullptr, ev));
            *nparts = 0;
            return 0;
        }
        CK(launch_forward<0>(e, p, fgrid));
    }
    *nparts = e->NB * fgrid;
    return 0;
}

// objective of (W, H[, hscale]) -> device scalar *out (all-reduced), no host sync
static int objective_to_ptr(salnmf_engine* e, const double* W, const double* hscale, bool weighted, double* out, int grid = 0) {
    int nparts = 0;
    CK(objective_partials(e, W, hscale, weighted, grid, &nparts, out));
    if (nparts > 0) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->objpart, nparts, 1, 1, out);
        HIPCK(hipGetLastError());
    }
    return allreduce(e, out, 1);
}
static int objective_to_slot(salnmf_engine* e, const double* W, const double* hscale, bool weighted, int slot, int grid = 0) {
    return objective_to_ptr(e, W, hscale, weighted, e->scal + slot, grid);
}

static int read_scalars(salnmf_engine* e, int first, int count, double* out) {
    HIPCK(hipMemcpyAsync(e->hpin, e->scal + first, count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    CK(check_abort(e));
    for (int i = 0; i < count; ++i) out[i] = e->hpin[i];
    return 0;
}

#ifdef SALNMF_DEV_POISON
// development builds only: raw device buffers for ad-hoc probes (tests/dev/poison_probe.py) (0 = xlx, 1 = objpart, 2 = KLpart)
extern "C" int salnmf_debug_read(salnmf_engine* e, int which, double* out, int64_t count) {
    const double* src = which == 0 ? e->xlx : which == 1 ? e->objpart : e->KLpart;
    if (!src) return fail("buffer not allocated");
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpy(out, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int salnmf_objective(salnmf_engine* e, double* out) {
    if (!e || !out) return fail("null argument");
    CK(enter(e));
    CK(objective_to_slot(e, e->W, nullptr, true, 0));
    return read_scalars(e, 0, 1, out);
}

// the pinned ring of objective slots and slot's event
static int ensure_objective_slot(salnmf_engine* e, int slot) {
    if (slot < 0 || slot >= SALNMF_OBJECTIVE_SLOTS) return fail("slot must be in [0, %d)", SALNMF_OBJECTIVE_SLOTS);
    static_assert(SALNMF_OBJECTIVE_SLOTS * sizeof(double) <= SMALL_PINNED_BYTES, "the ring lives in one small pinned block");
    if (!e->objpin) HIPCK(acquire_pinned((void**)&e->objpin, 1));
    if (e->objev.empty()) e->objev.assign(SALNMF_OBJECTIVE_SLOTS, nullptr);
    // (with the system-scope fence: the host reads the slot once the event has completed)
    if (!e->objev[slot]) HIPCK(hipEventCreateWithFlags(&e->objev[slot], hipEventDisableTiming));
    return 0;
}

int salnmf_objective_async(salnmf_engine* e, int slot) {
    if (!e) return fail("null engine");
    CK(enter(e));
    CK(ensure_objective_slot(e, slot));
    // The value lands in pinned host memory straight from the reducing kernel, and the slot's event is that kernel's own
    // completion signal: the reader waits for THIS objective only, not for whatever was queued behind it (the next
    // block of steps), and no copy packet sits in the stream.  A sharded engine all-reduces the device copy first.
    int nparts = 0;
    if (!sharded(e)) {
        // (one forward launch per objective: it adds its partials up itself and its completion is the slot's event)
        CK(objective_partials(e, e->W, nullptr, true, 0, &nparts, e->objpin + slot, e->objev[slot]));
        if (nparts > 0) {
            hipExtLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, nullptr, e->objev[slot], 0, e->objpart, nparts, 1, 1,
                                  e->objpin + slot, (const double*)nullptr);
            HIPCK(hipGetLastError());
        }
        return 0;
    }
    CK(objective_partials(e, e->W, nullptr, true, 0, &nparts));
    if (!e->objring) HIPCK(hipMalloc(&e->objring, SALNMF_OBJECTIVE_SLOTS * sizeof(double)));
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, e->stream, e->objpart, nparts, 1, 1, e->objring + slot, (const double*)nullptr);
    HIPCK(hipGetLastError());
    CK(allreduce(e, e->objring + slot, 1));
    hipExtLaunchKernelGGL(copy_scalar_kernel, dim3(1), dim3(1), 0, e->stream, nullptr, e->objev[slot], 0, e->objpin + slot, (const double*)(e->objring + slot));
    HIPCK(hipGetLastError());
    return 0;
}

int salnmf_kl_step_objective(salnmf_engine* e, int slot, int n_steps, int n_given, int keep) {
    if (!e) return fail("null engine");
    if (n_given < 0 || n_given > e->K) return fail("n_given out of range");
    if (n_steps < 0) return fail("n_steps must not be negative");
    CK(enter(e));
    const bool fold = n_steps > 0 && !split(e) && !e->fast32 && !e->persistent && !sharded(e) && !e->wkl && !e->wlh && n_given < e->K &&
                      !small_path(e, n_given);  // (small cohorts: a forward pass, then all steps in one launch)
    if (!fold) {
        // the objective as a forward pass of its own, then the steps
        CK(salnmf_objective_async(e, slot));
        if (n_steps == 0) return 0;
        return keep ? salnmf_kl_step_keep(e, n_steps, n_given) : salnmf_kl_step(e, n_steps, n_given);
    }
    CK(ensure_objective_slot(e, slot));
    CK(ensure_xlogx(e));
    if (keep) {
        CK(ensure_halt(e));
        if (!e->Wkeep) HIPCK(hipMalloc(&e->Wkeep, (size_t)e->K * e->V * sizeof(double)));
        CK(flush_H_scale(e));
        e->keep_valid = false;
    }
    CK(kl_step_once(e, n_given, nullptr, keep != 0, slot));
    for (int i = 1; i < n_steps; ++i) CK(kl_step_once(e, n_given, nullptr));
    if (keep) e->keep_valid = true;
    return 0;
}

int salnmf_objective_read(salnmf_engine* e, int first, int count, double* out) {
    if (!e || !out) return fail("null argument");
    if (first < 0 || count < 0 || first + count > SALNMF_OBJECTIVE_SLOTS) return fail("slots out of range");
    if (count == 0) return 0;
    CK(enter(e));
    for (int i = first; i < first + count; ++i)
        if (e->objev.empty() || !e->objev[i]) return fail("slot %d has never been queued", i);
    // slots are filled in stream order: the caller reads ranges in the order it queued them, so the last one decides
    for (int i = first; i < first + count; ++i) HIPCK(hipEventSynchronize(e->objev[i]));
    for (int i = 0; i < count; ++i) out[i] = e->objpin[first + i];
    return check_abort(e);
}

int salnmf_samplewise_kl(salnmf_engine* e, double* out) {
    if (!e || !out) return fail("null argument");
    CK(enter(e));
    double* dev = nullptr;
    FwdParams p;
    CK(fwd_params(e, p));
    HIPCK(hipMalloc(&dev, (size_t)e->NB * e->Np * sizeof(double)));
    if (e->NC > 1) {
        int rcc = 0;
        for (int b = 0; b < e->NB && !rcc; ++b) {  // (feature blocks as well: a sum over the blocks, as below)
            FwdParams last{};
            last.out = dev + (size_t)b * e->Np;
            rcc = chunk_chain(e, 1, last, nullptr, nullptr, b);
        }
        if (!rcc) rcc = download(e, out, dev, (size_t)e->N);
        std::vector<double> part(e->NB > 1 ? (size_t)e->N : 0);
        for (int b = 1; b < e->NB && !rcc; ++b) {
            rcc = download(e, part.data(), dev + (size_t)b * e->Np, (size_t)e->N);
            if (!rcc)
                for (int64_t n = 0; n < e->N; ++n) out[n] += part[(size_t)n];
        }
        (void)hipFree(dev);
        return rcc;
    }
    int rc = 0;
    for (int b = 0; b < e->NB && !rc; ++b) {  // (per-sample divergences are sums over the features: one pass per feature block)
        FwdParams pb = p;
        if (e->NB > 1) {
            pb.X = e->X + (size_t)b * e->Np * VMAX;
            pb.W = e->W + (size_t)VMAX * b;
            pb.V = block_width(e, b);
        }
        pb.out = dev + (size_t)b * e->Np;
        rc = launch_forward<1>(e, pb);
    }
    if (!rc) rc = download(e, out, dev, (size_t)e->N);
    if (!rc && e->NB > 1) {
        std::vector<double> part((size_t)e->N);
        for (int b = 1; b < e->NB && !rc; ++b) {
            rc = download(e, part.data(), dev + (size_t)b * e->Np, (size_t)e->N);
            if (!rc)
                for (int64_t n = 0; n < e->N; ++n) out[n] += part[(size_t)n];
        }
    }
    (void)hipFree(dev);
    return rc;
}

int salnmf_reconstruct(salnmf_engine* e, double* out) {
    if (!e || !out) return fail("null argument");
    CK(enter(e));
    double* dev = nullptr;
    FwdParams p;
    CK(fwd_params(e, p));
    HIPCK(hipMalloc(&dev, (size_t)e->Np * VMAX * sizeof(double)));
    p.out = dev;
    int rc = 0;
    if (e->NC > 1 && e->NB > 1) {
        std::vector<double> part((size_t)e->N * VMAX);
        for (int b = 0; b < e->NB && !rc; ++b) {  // the chunk chain of one feature block at a time
            FwdParams last{};
            last.out = dev;
            const int vb = block_width(e, b);
            rc = chunk_chain(e, 2, last, nullptr, nullptr, b);
            if (!rc) rc = download_padded(e, part.data(), dev, vb, VMAX);
            if (!rc)
                for (int64_t n = 0; n < e->N; ++n)
                    memcpy(out + (size_t)n * e->V + (size_t)VMAX * b, part.data() + (size_t)n * vb, (size_t)vb * sizeof(double));
        }
    } else if (e->NC > 1) {
        FwdParams last{};
        last.out = dev;
        rc = chunk_chain(e, 2, last);
        if (!rc) rc = download_padded(e, out, dev, e->V, VMAX);
    } else if (e->NB == 1) {
        rc = launch_forward<2>(e, p);
        if (!rc) rc = download_padded(e, out, dev, e->V, VMAX);
    } else {
        // one feature block at a time: H @ W[:, block] into the scratch image, its columns into the caller's rows
        std::vector<double> part((size_t)e->N * VMAX);
        for (int b = 0; b < e->NB && !rc; ++b) {
            FwdParams pb = p;
            pb.W = e->W + (size_t)VMAX * b;
            pb.V = block_width(e, b);
            rc = launch_forward<2>(e, pb);
            if (!rc) rc = download_padded(e, part.data(), dev, pb.V, VMAX);
            if (!rc)
                for (int64_t n = 0; n < e->N; ++n)
                    memcpy(out + (size_t)n * e->V + (size_t)VMAX * b, part.data() + (size_t)n * pb.V, (size_t)pb.V * sizeof(double));
        }
    }
    (void)hipFree(dev);
    return rc;
}

#include "salnmf_host_mv.h"  // MvNMF entry points and their host-side logic
#include "salnmf_host_corr.h"  // CorrNMF / MultimodalCorrNMF entry points
#include "salnmf_host_init.h"  // device-side initialisation entry points
#include "salnmf_host_dist.h"  // multi-GPU entry points
#include "salnmf_host_profile.h"  // measurement entry points

}  // extern "C"
