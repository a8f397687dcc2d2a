// All-reduce of a small fp64 vector (the K x V numerator of the W update, the MvNMF line-search sums) over the
// GPUs of one node by direct peer stores through xGMI -- the exchange step of the sample-sharded update
// (the reference has no multi-device path; SURVEY.md section 8e specifies one all-reduce of K*V doubles per W update).
//
// Why not the RCCL call: the vector is 38 KB at K = 50.  A library all-reduce of that size is pure latency (several
// kernel-internal ring / tree hops, ~25-40 us on 8 GPUs), which is a third of a c3 step.  Here every rank stores its
// vector straight into an inbox on each peer (7 links in parallel, < 1 us of wire time), waits for the peers' words
// in its own inbox and adds the n vectors in rank order (its own term from the register
// it still holds) -- the same order on every rank, so all ranks end up with bit-identical sums and hence bit-identical W.
//
// Memory: the inboxes are uncached device allocations (hipDeviceMallocUncached: remote stores are visible to the home
// GPU without any cache maintenance there), exported / opened with hipIpc*MemHandle.  Layout per engine:
//   inbox[parity 0..1][source rank 0..n-1] = { uint64 word[2 * max_count] }: element i as { tag | low half }, { tag | high half }
// Two parities: a rank can run at most one exchange ahead of a peer (its exchange s+1 cannot finish before the peer has
// sent s+1, which the peer does only after it has finished reading s), so the slot of exchange s+2 is free by then.
// The tag is derived from the exchange's sequence number (monotonic; never reset; never 0).
// Ordering: none needed beyond the atomicity of an 8-byte store -- every word says by its tag whether it has arrived
// (p2p_exchange).  Rounds 2-4 drained the payload stores, raised one flag per workgroup and peer and read the payload after
// the flags: three trips over the fabric on the critical path where this layout has one.  Every wait is bounded (20 s of the
// 100 MHz clock): on expiry the host-visible abort word is set, and a device word that makes the engine's later exchanges
// return at once.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "salnmf_kernels.h"

namespace salnmf {

constexpr int P2P_MAX_RANKS = 8;
constexpr int P2P_BLOCK = 256;
constexpr unsigned long long P2P_TIMEOUT_TICKS = 2000000000ull;  // default: 20 s of the 100 MHz clock (ranks may be that far apart on the host side); salnmf_set_p2p_timeout_ms overrides
constexpr int P2P_MAX_WG = 64;  // flags per slot: workgroups of one exchange (max_count <= 16384 doubles), or signature rows

struct P2PParams {
    double* buf;            // in: this rank's vector, out: the sum over the ranks (p2p_allreduce_kernel)
    int count;
    int rank, n_ranks;
    int parity;
    unsigned long long seq;
    unsigned long long tag;  // this exchange's arrival tag in the high half of a word: ((seq mod (2^32 - 1)) + 1) << 32
    size_t slot;            // 8-byte words per (parity, source) slot = 2 * max_count
    size_t max_count;
    double* inbox[P2P_MAX_RANKS];  // base of every rank's inbox as mapped here ([rank] = the local allocation)
    unsigned long long timeout_ticks;  // of the 100 MHz clock
    unsigned* abort_host;   // pinned host word: 2 = an exchange gave up
    unsigned* abort_dev;    // device word, set with it: later exchanges of this engine return at once instead of waiting again
    // optional (salnmf_profile_sharded_steps): six s_memrealtime stamps (100 MHz) per workgroup of tail_p2p_kernel --
    // [0] start [1] local slabs reduced [2] row stored to the peers [3] every peer's row seen
    // [4] rows summed [5] W row finished
    unsigned long long* stamps;
};

__device__ __forceinline__ void p2p_stamp(const P2PParams& p, int wg, int i, int tid) {
    if (p.stamps != nullptr && tid == 0) p.stamps[wg * 6 + i] = __builtin_amdgcn_s_memrealtime();
}

typedef __attribute__((address_space(1))) double gdouble_t;
typedef __attribute__((address_space(1))) unsigned long long gflag_t;

// One workgroup's part of an exchange: element `idx` of the vector (value v, where `active`), completion flag `flag_idx`
// of the slot.  Called by every thread of the workgroup; returns the sum over the ranks in rank order (0 where inactive
// or after a wait gave up, in which case the abort word is set).
// The engine's "an earlier exchange gave up" word.  A kernel that has other work before its exchange issues this load first,
// so that its round trip is not on the exchange's critical path.
__device__ __forceinline__ unsigned p2p_abort_word(const P2PParams& p, int) {
    // (every lane: one request per wave; a lane's own copy lets it skip its wait below, the workgroup's common decision is lane 0's)
    return __hip_atomic_load((__attribute__((address_space(1))) unsigned*)p.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

//   abort_word: p2p_abort_word(p, tid)
// Round 5: the payload carries its own arrival tag (the "low-latency" layout of collective libraries): a double travels as
// two 8-byte words { tag : 32 | half of the value : 32 }, each stored atomically; the reader polls ITS OWN element's words from
// every source until all carry this exchange's tag.  No drain of the stores (the acknowledgement's trip back), no flag store
// behind it (a second trip) and no separate read of the payload after the flags (a third): one one-way trip on the critical
// path.  8-byte stores are single transactions on the fabric; a reader that sees one word of a pair new and the other old
// simply polls again.
__device__ __forceinline__ double p2p_exchange(const P2PParams& p, int idx, bool active, int flag_idx, double v, int tid, unsigned abort_word) {
    typedef __attribute__((address_space(1))) unsigned long long gword_t;
    const size_t mine = ((size_t)p.parity * p.n_ranks + p.rank) * p.slot;
    __shared__ int failed;
    if (tid == 0) failed = abort_word != 0;  // (read behind the barrier below; a wait that gives up sets it too)
    const unsigned long long tag = p.tag;    // never 0 (the inbox starts zeroed), differs between exchanges s and s + 2 (host: next_exchange)
    // (this rank's own contribution stays in its register: no round trip through its own uncached inbox)
    if (active) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        const unsigned long long w0 = tag | (bits & 0xFFFFFFFFull), w1 = tag | (bits >> 32);
        for (int r = 0; r < p.n_ranks; ++r)
            if (r != p.rank) {
                gword_t* dst = (gword_t*)(p.inbox[r] + mine) + 2 * (size_t)idx;
                __hip_atomic_store(dst, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(dst + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
    }
    p2p_stamp(p, flag_idx, 2, tid);
    // every lane waits for its own element from every other source (the inbox is uncached memory: no load is served from a
    // cache).  The clock is read only once a poll has come back empty: the common case -- the peers' rows are there -- pays
    // for neither the read nor the loop.  A lane that knows of an earlier failure does not wait again.
    double t[P2P_MAX_RANKS];
#pragma unroll
    for (int r = 0; r < P2P_MAX_RANKS; ++r) t[r] = 0.0;
    if (active && p.n_ranks > 1 && abort_word == 0) {
        unsigned long long t0 = 0;
        for (unsigned spins = 0;; ++spins) {
            unsigned long long w[P2P_MAX_RANKS][2];
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; ++r) {
                const bool peer = r < p.n_ranks && r != p.rank;
                const gword_t* src = (const gword_t*)(p.inbox[p.rank] + ((size_t)p.parity * p.n_ranks + (peer ? r : p.rank)) * p.slot) + 2 * (size_t)idx;
                w[r][0] = peer ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : tag;
                w[r][1] = peer ? __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : tag;
            }
            bool all = true;
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; ++r) all = all && (w[r][0] >> 32) == (tag >> 32) && (w[r][1] >> 32) == (tag >> 32);
            if (all) {
#pragma unroll
                for (int r = 0; r < P2P_MAX_RANKS; ++r) t[r] = __longlong_as_double((long long)((w[r][0] & 0xFFFFFFFFull) | (w[r][1] << 32)));
                break;
            }
            if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                failed = 1;
                __hip_atomic_store((__attribute__((address_space(1))) unsigned*)p.abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((__attribute__((address_space(1))) unsigned*)p.abort_host, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    p2p_stamp(p, flag_idx, 3, tid);
    if (failed || !active) return 0.0;
    // the sum in rank order (this rank's own term from the register)
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < P2P_MAX_RANKS; ++r)
        if (r < p.n_ranks) s += (r == p.rank) ? v : t[r];
    return s;
}

__global__ void __launch_bounds__(P2P_BLOCK) p2p_allreduce_kernel(P2PParams p) {
    const int tid = threadIdx.x;
    const int i = blockIdx.x * P2P_BLOCK + tid;
    const bool active = i < p.count;
    const unsigned ab = p2p_abort_word(p, tid);
    const double s = p2p_exchange(p, i, active, blockIdx.x, active ? p.buf[i] : 0.0, tid, ab);
    if (active) p.buf[i] = s;
}

// The W tail of a sharded joint step with the exchange inside: workgroup k reduces row k of the local numerator slabs,
// exchanges that row with the peers (flag k of the slot) and finishes row k of W -- one launch where the RCCL path
// takes a reduce launch, the library all-reduce and a finish launch.  Arithmetic and summation orders of the two
// tail_row halves are those of tail_kernel.
struct TailP2PParams {
    TailParams t;
    P2PParams x;
};

__global__ void __launch_bounds__(TAIL_BLOCK) tail_p2p_kernel(TailP2PParams p) {
    __shared__ TailScratch S;
    const int k = blockIdx.x, tid = threadIdx.x;
    const TailParams& t = p.t;
    p2p_stamp(p.x, k, 0, tid);
    const unsigned ab = p2p_abort_word(p.x, tid);  // (in flight beside the slab loads)
    const double wold = (tid < t.V) ? t.W[k * t.V + tid] : 0.0;  // old row of W, for the second half: likewise
    tail_row<TAIL_BLOCK, false>(S, tid, k, t.Gpart, t.nslabs, t.G, t.W, t.Wout, t.V, t.K, t.n_given, t.clip_mode, false);
    p2p_stamp(p.x, k, 1, tid);
    const bool active = tid < t.V;  // (tail_row left the row's local sum in S.red[0][v], behind a barrier)
    const double total = p2p_exchange(p.x, k * t.V + tid, active, k, active ? S.red[0][tid] : 0.0, tid, ab);
    p2p_stamp(p.x, k, 4, tid);
    // the reduced row goes to G (the engine's buffer) and, through LDS, straight into the W update: no store -> load
    // round trip through global memory in between
    if (active) {
        t.G[k * t.V + tid] = total;
        S.red[0][tid] = total;
    }
    __syncthreads();
    tail_row<TAIL_BLOCK, false>(S, tid, k, nullptr, -1, t.G, t.W, t.Wout, t.V, t.K, t.n_given, t.clip_mode, true, wold);
    p2p_stamp(p.x, k, 5, tid);
}

}  // namespace salnmf
