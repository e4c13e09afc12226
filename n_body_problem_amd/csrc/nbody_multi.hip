// nbody_multi.hip -- rows sharded over the GPUs of one node, the exchange owned by the library.
//
// The reference is single-GPU (main_project/kernel.cu:1225-1242: one device, default stream); SURVEY.md 8b/8e make the
// multi-GPU step the build's own contract: "ctx owns scratch, replicas, streams, RCCL comms ... multi-GPU driven from one
// host thread with ncclGroupStart/End (or one thread per device) -- invisible at the ABI".  A nbody_multi owns, per
// local rank: one shard context (nbody_create_shard), a full replica of the positions, the rank's velocity rows, a
// compute stream, a second compute stream for the launch that waits for the exchange, a communication stream and one
// RCCL communicator.  nbody_multi_step is the whole step of every local rank:
//
//     own-chunk force launch (needs no remote rows)  ||  all-gather of the previous step's updated rows (RCCL, in place)
//     complement force launch, ordered after the all-gather, on the second stream (fills the first launch's tail)
//     pair-once mode: nbody_sym_reduce -> all-gather of the column-side sums (one per step, exposed)
//     update (kick-drift) or kick (kick-drift-kick); the next all-gather is issued right behind it
//
// NBODY_EXCHANGE_RING spells the position all-gather out as P-1 ncclSend/ncclRecv hops with one event per hop; the
// force launch of chunk (rank - h) starts as hop h lands.  Two process models, one code path: every rank in this
// process (nbody_multi_create: one communicator per device from an id made here, collectives of the local ranks fused with
// ncclGroupStart/End), or one rank per process (nbody_multi_create_rank: an id the caller distributes).
// NBODY_TRANSPORT_PEER_COPY (single process only) moves the same slices with hipMemcpyPeerAsync instead of RCCL: RCCL
// refuses two ranks on one device, so this is also how two shards on ONE GPU are tested against a single context.
//
// Failure detection: the communicators are made NON-BLOCKING (ncclCommInitRankConfig, blocking = 0) by a helper
// thread the caller waits for under its timeout -- so a peer that never arrives is reported by nbody_multi_create* itself (the
// bootstrap is the first thing that can hang; RCCL 2.27.7 sits in it inside the call) -- and after every RCCL call that answers
// ncclInProgress the library polls ncclCommGetAsyncError before it records anything on the stream behind the call.  The asynchronous error state is polled after
// every step and inside every wait; a wait that exceeds the timeout aborts the communicators and returns NBODY_ERR_DEVICE
// instead of hanging on a dead peer.
//
// Determinism: chunk boundaries are multiples of split_len (whole split groups in the pair-once mode), so the state is
// bit-identical to one context on the same padded system for any world size, exchange mode or arrival order.
#include "../../include/nbody.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Channel {  // one exchange in flight per channel and local rank
    std::vector<hipEvent_t> ready;  // on the compute stream: the rank's slice is final and it reads nobody else's any more
    std::vector<hipEvent_t> done;   // on the comm stream: the rank's part of the exchange has been enqueued/has landed
};

struct Rank {
    int rank = 0, device = 0;
    nbody_ctx *ctx = nullptr;
    float *pos = nullptr;       // replica: n_padded x float4
    float *vel = nullptr;       // own rows: chunk x float4
    float *colparts = nullptr;  // pair-once: NBODY_SYM_GROUPS x n_padded x float4
    hipStream_t compute = nullptr, side = nullptr, comm = nullptr;
    ncclComm_t nccl = nullptr;
    hipEvent_t ev_start = nullptr, ev_side = nullptr;
    std::vector<hipEvent_t> ev_hop;  // ring: hop h has been sent by this rank (PEER) / has landed at this rank (RCCL)
    std::vector<hipEvent_t> ev_step; // nbody_multi_step_n: the end of the last kStepsInFlight steps on the compute stream
    unsigned long long *scratch = nullptr;  // 4 x 8 bytes on the device (checksum, all-reduce staging)
    float *vel_all = nullptr;      // n_padded x float4: every rank's velocity rows (layout refresh, multi-process download)
    int64_t *order_dev = nullptr;  // NBODY_ORDER_MORTON: slot k of the replica holds the caller's body order_dev[k] (n_bodies)
    hipEvent_t ev_vel = nullptr;   // on the comm stream: this rank's part of a velocity gather has been enqueued / has landed
    // measurement (nbody_multi_timing_*): event pairs not yet added to the totals, by kind
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    double span_ms[4] = {0, 0, 0, 0};
    int64_t span_count[4] = {0, 0, 0, 0};
    double host_ms = 0;
    int64_t steps = 0;
};

enum { kSpanPosComm = 0,   // comm stream: the position exchange from the moment it may start to its arrival
       kSpanPosWait = 1,   // consumer stream: from the moment the waiting force launch could have started to the arrival
       kSpanColumns = 2,   // compute stream: the pair-once column-sum exchange (not hidden by design)
       kSpanReorder = 3 }; // compute stream: a layout refresh

}  // namespace

constexpr int kStepsInFlight = 4;  // nbody_multi_step_n lets the host run this many steps ahead of the device

struct nbody_multi {
    nbody_multi_config cfg{};
    int world = 1;
    int64_t n_bodies = 0, n_padded = 0, chunk = 0, split_len = 0;
    std::vector<Rank> ranks;  // the local ranks
    Channel ch_pos, ch_col;
    std::vector<int64_t> order;  // NBODY_ORDER_MORTON: host copy of Rank::order_dev, valid while order_host_valid (the
                                 // layout is refreshed on the device; the host asks for the order when it needs it)
    bool order_host_valid = false;
    std::vector<float> eps_caller;             // the per-particle softening lengths as given (the caller's order), or empty
    bool eps_on = false;
    int64_t reorder_period = 0, steps_since_order = 0;  // NBODY_ORDER_MORTON: refresh the layout every so many steps (0: never)
    bool exchange_in_flight = false;  // the position exchange of the last update has been issued and not yet consumed
    bool kdk_ready = false;
    bool have_state = false;
    double timeout_s = 600.0;  // per wait: a wait never covers more than kStepsInFlight steps
    bool timing = false;       // nbody_multi_timing_enable
    bool failed = false;       // an exchange failed or timed out: the communicators are gone, destroy must not wait for peers
    std::string err;
};

static thread_local std::string g_multi_create_error;

static int mfail(nbody_multi *m, int status, const std::string &msg)
{
    if (m)
        m->err = msg;
    else
        g_multi_create_error = msg;
    return status;
}

// Nothing is thrown across the C ABI: the entry points that build host vectors and strings run inside this.
template <class F>
static int guarded(nbody_multi *m, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try {
            return mfail(m, NBODY_ERR_ALLOC, "host allocation failed");
        } catch (...) {
            return NBODY_ERR_ALLOC;
        }
    } catch (...) {
        try {
            return mfail(m, NBODY_ERR_DEVICE, "unexpected C++ exception");
        } catch (...) {
            return NBODY_ERR_DEVICE;
        }
    }
}

#define MHIP(m, call)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return mfail((m), e_ == hipErrorOutOfMemory ? NBODY_ERR_ALLOC : NBODY_ERR_DEVICE,                  \
                         std::string(#call) + ": " + hipGetErrorString(e_));                                   \
    } while (0)
// A non-blocking communicator answers ncclInProgress where a blocking one would have blocked (ncclGroupEnd above all): the
// call counts as made only once ncclCommGetAsyncError has left that state -- an event recorded on the stream before that
// could land in front of the collective's kernel.  nccl_settle waits for it, bounded by the timeout.
static int nccl_settle(nbody_multi *m, const char *what);
#define MNCCL(m, call)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ == ncclInProgress) {                                                                            \
            int s_ = nccl_settle((m), #call);                                                                  \
            if (s_ != NBODY_OK)                                                                                \
                return s_;                                                                                     \
        } else if (r_ != ncclSuccess)                                                                          \
            return mfail((m), NBODY_ERR_DEVICE, std::string(#call) + ": RCCL: " + ncclGetErrorString(r_));     \
    } while (0)
// The event edges of the RCCL branch, numbered: tools/edge_mutations.py builds the library with one of them left out
// (-DNB_DROP_EDGE=k, never in the product) and checks that the multi-process tests over the RCCL test double turn red -- the
// evidence that those tests can see a missing hipStreamWaitEvent (profiles/r04_edge_mutations.txt).
#ifndef NB_DROP_EDGE
#define NB_DROP_EDGE 0
#endif
#define EDGE(k, m, call)                                                                                       \
    do {                                                                                                       \
        if (NB_DROP_EDGE != (k))                                                                               \
            MHIP((m), call);                                                                                   \
    } while (0)
#define MCTX(m, r, call)                                                                                       \
    do {                                                                                                       \
        int s_ = (call);                                                                                       \
        if (s_ != NBODY_OK)                                                                                    \
            return mfail((m), s_, std::string(#call) + " (rank " + std::to_string((r).rank) + "): " +           \
                                      nbody_last_error((r).ctx));                                              \
    } while (0)

__global__ void nbody_checksum_kernel(const unsigned *words, size_t n, unsigned long long *out)
{
    unsigned long long s = 0;  // integer sums are associative: the result does not depend on the order of the atomics
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        s += (unsigned long long)words[i] * (unsigned long long)((i & 1023u) + 1u);
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(out, s);
}

static bool rccl(const nbody_multi *m) { return m->cfg.transport == NBODY_TRANSPORT_RCCL; }
static bool pair_once(const nbody_multi *m) { return m->cfg.force_mode == NBODY_FORCE_SYMMETRIC; }
static bool all_local(const nbody_multi *m) { return (int)m->ranks.size() == m->world; }

// ---- pure host helpers (CPU-testable) --------------------------------------------------------------------------

extern "C" int nbody_multi_geometry(int64_t n_bodies, int world_size, int force_mode, int64_t split_len, int64_t *n_padded,
                                    int64_t *rows_per_rank, int64_t *split_len_out)
{
    if (n_bodies < 0 || world_size < 1 || (force_mode != NBODY_FORCE_ONE_SIDED && force_mode != NBODY_FORCE_SYMMETRIC))
        return NBODY_ERR_INVALID;
    if (split_len == 0)
        split_len = force_mode == NBODY_FORCE_SYMMETRIC ? nbody_pair_once_split_len(n_bodies) : nbody_default_split_len(n_bodies);
    if (split_len <= 0 || split_len % 64 != 0 || (force_mode == NBODY_FORCE_SYMMETRIC && split_len % 256 != 0))
        return NBODY_ERR_INVALID;
    const int64_t n_splits = n_bodies > 0 ? (n_bodies + split_len - 1) / split_len : 1;
    int64_t padded, chunk;
    if (force_mode == NBODY_FORCE_SYMMETRIC) {
        // the partial sums are added in NBODY_SYM_GROUPS groups of ceil(n_splits / groups) splits; a rank owns whole groups
        if (NBODY_SYM_GROUPS % world_size != 0)
            return NBODY_ERR_INVALID;
        const int64_t group_splits = (n_splits + NBODY_SYM_GROUPS - 1) / NBODY_SYM_GROUPS;
        padded = NBODY_SYM_GROUPS * group_splits * split_len;
        chunk = (NBODY_SYM_GROUPS / world_size) * group_splits * split_len;
    } else {
        const int64_t splits_per_rank = (n_splits + world_size - 1) / world_size;
        chunk = splits_per_rank * split_len;
        padded = chunk * world_size;
    }
    if (padded > ((int64_t)1 << 30))
        return NBODY_ERR_INVALID;
    if (n_padded) *n_padded = padded;
    if (rows_per_rank) *rows_per_rank = chunk;
    if (split_len_out) *split_len_out = split_len;
    return NBODY_OK;
}

extern "C" int nbody_multi_ring_schedule(int rank, int world_size, int hop, int *send_chunk, int *recv_chunk)
{
    if (world_size < 1 || rank < 0 || rank >= world_size || hop < 1 || hop >= world_size)
        return NBODY_ERR_INVALID;
    // hop h: every rank passes on the chunk it received in hop h-1 (its own in hop 1) to rank+1
    if (send_chunk) *send_chunk = ((rank - hop + 1) % world_size + world_size) % world_size;
    if (recv_chunk) *recv_chunk = ((rank - hop) % world_size + world_size) % world_size;
    return NBODY_OK;
}

extern "C" const char *nbody_multi_last_error(const nbody_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

extern "C" int nbody_multi_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) <= NBODY_UNIQUE_ID_BYTES, "ncclUniqueId does not fit NBODY_UNIQUE_ID_BYTES");
    if (!id128)
        return NBODY_ERR_INVALID;
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess)
        return mfail(nullptr, NBODY_ERR_DEVICE, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memset(id128, 0, NBODY_UNIQUE_ID_BYTES);
    std::memcpy(id128, &id, sizeof id);
    return NBODY_OK;
}

// ---- creation ----------------------------------------------------------------------------------------------------

static int make_events(nbody_multi *m, std::vector<hipEvent_t> &v, size_t n)
{
    v.assign(n, nullptr);
    for (auto &e : v)
        MHIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return NBODY_OK;
}

static int setup_ranks(nbody_multi *m)
{
    const size_t nl = m->ranks.size();
    m->ch_pos.ready.assign(nl, nullptr);
    m->ch_pos.done.assign(nl, nullptr);
    m->ch_col.ready.assign(nl, nullptr);
    m->ch_col.done.assign(nl, nullptr);
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        int s = nbody_create_shard(&r.ctx, r.device, m->n_padded, (int64_t)r.rank * m->chunk, m->chunk, m->split_len);
        if (s != NBODY_OK)
            return mfail(m, s, std::string("nbody_create_shard: ") + nbody_last_error(nullptr));
        MCTX(m, r, nbody_set_force_mode(r.ctx, m->cfg.force_mode));
        MCTX(m, r, nbody_set_integrator(r.ctx, m->cfg.integrator));
        MHIP(m, hipStreamCreateWithFlags(&r.compute, hipStreamNonBlocking));
        MHIP(m, hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking));
        // The exchange's kernels (RCCL's, or the copy engine's fallbacks) must win CUs from a force launch that fills every
        // CU with 0.6 ms workgroups, and the complement launch waits for them: the communication stream gets the highest
        // priority the device offers (at the default priority the column-sum exchange sat 35 ms behind another rank's force
        // kernels on a shared GPU, at the highest 0.7 ms: profiles/r03_rehearsal_comm_priority_ab_two_ranks_one_gpu.txt).
        {
            int least = 0, greatest = 0;
            MHIP(m, hipDeviceGetStreamPriorityRange(&least, &greatest));
            MHIP(m, hipStreamCreateWithPriority(&r.comm, hipStreamNonBlocking, greatest));
        }
        if (m->n_padded) {
            MHIP(m, hipMalloc((void **)&r.pos, sizeof(float) * 4 * (size_t)m->n_padded));
            MHIP(m, hipMemsetAsync(r.pos, 0, sizeof(float) * 4 * (size_t)m->n_padded, r.compute));
            MHIP(m, hipMalloc((void **)&r.vel, sizeof(float) * 4 * (size_t)m->chunk));
            MHIP(m, hipMemsetAsync(r.vel, 0, sizeof(float) * 4 * (size_t)m->chunk, r.compute));
        }
        MHIP(m, hipMalloc((void **)&r.scratch, 4 * sizeof(unsigned long long)));
        if (pair_once(m) && m->world > 1 && m->n_padded) {
            MHIP(m, hipMalloc((void **)&r.colparts, sizeof(float) * 4 * (size_t)NBODY_SYM_GROUPS * (size_t)m->n_padded));
            MCTX(m, r, nbody_sym_set_colparts(r.ctx, r.colparts));
        }
        for (hipEvent_t *e : {&r.ev_start, &r.ev_side, &r.ev_vel, &m->ch_pos.ready[i], &m->ch_pos.done[i], &m->ch_col.ready[i],
                              &m->ch_col.done[i]})
            MHIP(m, hipEventCreateWithFlags(e, hipEventDisableTiming));
        if (m->cfg.body_order == NBODY_ORDER_MORTON && m->n_bodies) {
            MHIP(m, hipMalloc((void **)&r.order_dev, sizeof(int64_t) * (size_t)m->n_bodies));
            MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
            MCTX(m, r, nbody_order_identity(r.ctx, r.order_dev, m->n_bodies));
        }
        int rc = make_events(m, r.ev_hop, (size_t)m->world);
        if (rc == NBODY_OK)
            rc = make_events(m, r.ev_step, (size_t)kStepsInFlight);
        if (rc != NBODY_OK)
            return rc;
        MHIP(m, hipStreamSynchronize(r.compute));
    }
    if (!rccl(m))  // peer copies between distinct devices go over xGMI directly when peer access is on
        for (Rank &a : m->ranks)
            for (Rank &b : m->ranks)
                if (a.device != b.device) {
                    int can = 0;
                    if (hipSetDevice(a.device) == hipSuccess && hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can)
                        (void)hipDeviceEnablePeerAccess(b.device, 0);  // "already enabled" is fine
                    (void)hipGetLastError();
                }
    return NBODY_OK;
}

static int create_common(nbody_multi **out, const nbody_multi_config *cfg, int world, nbody_multi **made)
{
    if (!out)
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create: out is NULL");
    *out = nullptr;
    if (!cfg)
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create: config is NULL");
    if ((cfg->body_order != NBODY_ORDER_GIVEN && cfg->body_order != NBODY_ORDER_MORTON) ||
        (cfg->integrator != NBODY_INTEGRATOR_KICK_DRIFT && cfg->integrator != NBODY_INTEGRATOR_KDK) ||
        (cfg->exchange != NBODY_EXCHANGE_ALLGATHER && cfg->exchange != NBODY_EXCHANGE_RING) ||
        (cfg->transport != NBODY_TRANSPORT_RCCL && cfg->transport != NBODY_TRANSPORT_PEER_COPY))
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create: unknown integrator, exchange or transport");
    int force_mode = cfg->force_mode;
    if (force_mode == NBODY_FORCE_AUTO)  // the pair-once kernels where they are faster and the rank count allows them
        force_mode = cfg->n_bodies >= NBODY_PAIR_ONCE_MIN_BODIES && NBODY_SYM_GROUPS % world == 0 ? NBODY_FORCE_SYMMETRIC
                                                                                                  : NBODY_FORCE_ONE_SIDED;
    int64_t padded = 0, chunk = 0, split = 0;
    if (nbody_multi_geometry(cfg->n_bodies, world, force_mode, cfg->split_len, &padded, &chunk, &split) != NBODY_OK)
        return mfail(nullptr, NBODY_ERR_INVALID,
                     "nbody_multi_create: bad geometry (n_bodies >= 0, split_len a multiple of 64 -- of 256 in the pair-once mode --, and the pair-once mode "
                     "shards over 1, 2, 4 or 8 ranks)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return mfail(nullptr, NBODY_ERR_NO_DEVICE, std::string("nbody_multi_create: no HIP device (") + hipGetErrorString(e) +
                                                       "); this library has no CPU path");
    nbody_multi *m = new (std::nothrow) nbody_multi;
    if (!m)
        return mfail(nullptr, NBODY_ERR_ALLOC, "nbody_multi_create: host allocation failed");
    m->cfg = *cfg;
    m->cfg.force_mode = force_mode;
    m->world = world;
    m->n_bodies = cfg->n_bodies;
    m->n_padded = padded;
    m->chunk = chunk;
    m->split_len = split;
    if (const char *t = getenv("NBODY_EXCHANGE_TIMEOUT_S"))  // the one environment variable the library reads (nbody.h)
        if (atof(t) > 0)
            m->timeout_s = atof(t);
    if (cfg->create_timeout_s > 0)  // bounds the creation of the communicators too (nbody_multi_set_timeout comes too late for it)
        m->timeout_s = (double)cfg->create_timeout_s;
    *made = m;
    return NBODY_OK;
}

extern "C" int nbody_multi_destroy(nbody_multi *m);
static void abort_communicators(nbody_multi *m);

// Every local communicator out of ncclInProgress: the call (creation, a collective, a group) has been made and, for a call
// that takes a stream, enqueued.  An error state or a wait beyond the timeout aborts the communicators.
static int nccl_settle(nbody_multi *m, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (Rank &r : m->ranks) {
        if (!r.nccl)
            continue;
        for (;;) {
            ncclResult_t state = ncclSuccess;
            const ncclResult_t q = ncclCommGetAsyncError(r.nccl, &state);
            if (q == ncclSuccess && state == ncclSuccess)
                break;
            if (q != ncclSuccess || state != ncclInProgress) {
                const ncclResult_t bad = q != ncclSuccess ? q : state;
                const std::string msg = std::string(what) + ": RCCL reported an error on rank " + std::to_string(r.rank) + ": " +
                                        ncclGetErrorString(bad) + " (" + ncclGetLastError(r.nccl) + "); the communicators were aborted";
                abort_communicators(m);
                return mfail(m, NBODY_ERR_DEVICE, msg);
            }
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > m->timeout_s) {
                abort_communicators(m);
                return mfail(m, NBODY_ERR_DEVICE, std::string(what) + ": timed out after " + std::to_string((int)waited) +
                                                      " s inside RCCL (a peer rank never arrived or is stuck); the communicators were aborted");
            }
            std::this_thread::sleep_for(std::chrono::microseconds(++spins < 200 ? 5 : 200));
        }
    }
    return NBODY_OK;
}

// One non-blocking communicator per local rank from the id (one process per GPU: one rank; every rank in this process: all of
// them inside one group, what ncclCommInitAll does with blocking ones).  The bootstrap is where a job with a missing rank
// hangs first, and RCCL 2.27.7 (ROCm 7.2) sits in it INSIDE ncclCommInitRankConfig whatever config.blocking says
// (tests/rccl_probe, profiles/r04_rccl_nonblocking_probe.txt) -- so the creation runs in a helper thread and the caller waits
// for it under the timeout: where RCCL honours blocking = 0 the thread polls ncclCommGetAsyncError, where it does not the
// thread sits in the call; either way nbody_multi_create* is back in time.  A helper that is still inside RCCL then is left
// behind (it owns everything it touches; if it ever comes back it aborts what it made).
struct CommCreation {
    std::vector<int> devices, ranks;
    int world = 0;
    ncclUniqueId id;
    std::vector<ncclComm_t> comms;
    ncclResult_t result = ncclSuccess;
    std::string where;
    std::atomic<bool> done{false}, abandoned{false};
};

static void create_communicators_thread(std::shared_ptr<CommCreation> job)
{
    ncclConfig_t config = NCCL_CONFIG_INITIALIZER;
    config.blocking = 0;
    const bool group = job->devices.size() > 1;
    ncclResult_t bad = ncclSuccess;
    const char *where = "ncclCommInitRankConfig";
    if (group) {
        bad = ncclGroupStart();
        where = "ncclGroupStart";
    }
    for (size_t i = 0; i < job->devices.size() && bad == ncclSuccess; ++i) {
        if (hipSetDevice(job->devices[i]) != hipSuccess) {
            bad = ncclUnhandledCudaError;
            where = "hipSetDevice";
            break;
        }
        const ncclResult_t q = ncclCommInitRankConfig(&job->comms[i], job->world, job->id, job->ranks[i], &config);
        if (q != ncclSuccess && q != ncclInProgress) {
            bad = q;
            where = "ncclCommInitRankConfig";
            job->comms[i] = nullptr;
        }
    }
    if (group) {
        const ncclResult_t g = ncclGroupEnd();
        if (bad == ncclSuccess && g != ncclSuccess && g != ncclInProgress) {
            bad = g;
            where = "ncclGroupEnd";
        }
    }
    // a communicator that honours blocking = 0 is ready once its asynchronous state has left ncclInProgress
    for (size_t i = 0; i < job->comms.size() && bad == ncclSuccess; ++i)
        while (job->comms[i] && !job->abandoned.load()) {
            ncclResult_t state = ncclSuccess;
            const ncclResult_t q = ncclCommGetAsyncError(job->comms[i], &state);
            if (q != ncclSuccess || (state != ncclSuccess && state != ncclInProgress)) {
                bad = q != ncclSuccess ? q : state;
                where = "creating the RCCL communicators";
                break;
            }
            if (state == ncclSuccess)
                break;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    if (bad != ncclSuccess || job->abandoned.load()) {  // nobody will use them
        for (ncclComm_t &c : job->comms)
            if (c) {
                (void)ncclCommAbort(c);
                c = nullptr;
            }
    }
    job->result = bad;
    job->where = where;
    job->done.store(true);
}

static int init_communicators(nbody_multi *m, const ncclUniqueId &id)
{
    std::shared_ptr<CommCreation> job;
    try {
        job = std::make_shared<CommCreation>();
        for (const Rank &r : m->ranks) {
            job->devices.push_back(r.device);
            job->ranks.push_back(r.rank);
        }
        job->comms.assign(m->ranks.size(), nullptr);
        job->world = m->world;
        job->id = id;
        std::thread(create_communicators_thread, job).detach();
    } catch (...) {
        return mfail(m, NBODY_ERR_ALLOC, "creating the RCCL communicators: no helper thread");
    }
    bool dup = false;
    for (size_t i = 0; i < m->ranks.size(); ++i)
        for (size_t j = 0; j < i; ++j)
            dup |= m->ranks[i].device == m->ranks[j].device;
    const char *hint = dup ? " (two ranks on one device: NBODY_TRANSPORT_PEER_COPY serves those)" : "";
    const auto t0 = std::chrono::steady_clock::now();
    while (!job->done.load()) {
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > m->timeout_s) {
            job->abandoned.store(true);
            m->failed = true;
            return mfail(m, NBODY_ERR_DEVICE, "creating the RCCL communicators: timed out after " + std::to_string((int)waited) +
                                                  " s (a peer rank never arrived: RCCL is still in its bootstrap)" + hint);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (job->result != ncclSuccess)
        return mfail(m, NBODY_ERR_DEVICE, job->where + ": RCCL: " + ncclGetErrorString(job->result) + hint);
    for (size_t i = 0; i < m->ranks.size(); ++i)
        m->ranks[i].nccl = job->comms[i];
    return NBODY_OK;
}

extern "C" int nbody_multi_create(nbody_multi **out, const nbody_multi_config *cfg, const int *devices, int n_devices)
{
    if (!devices || n_devices < 1)
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create: need at least one device");
    nbody_multi *m = nullptr;
    int rc = create_common(out, cfg, n_devices, &m);
    if (rc != NBODY_OK)
        return rc;
    // two ranks on one device: RCCL itself refuses them ("invalid usage", profiles/r02_rccl_two_ranks_one_device_refused.txt)
    // and init_communicators passes that on with a hint; the library has no opinion of its own (the RCCL test double of
    // tests/fake_rccl serves any devices)
    m->ranks.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        m->ranks[(size_t)i].rank = i;
        m->ranks[(size_t)i].device = devices[i];
    }
    rc = setup_ranks(m);
    if (rc == NBODY_OK && rccl(m)) {
        ncclUniqueId id;
        ncclResult_t r = ncclGetUniqueId(&id);
        rc = r != ncclSuccess ? mfail(m, NBODY_ERR_DEVICE, std::string("ncclGetUniqueId: RCCL: ") + ncclGetErrorString(r))
                              : init_communicators(m, id);
    }
    if (rc != NBODY_OK) {
        g_multi_create_error = m->err;
        nbody_multi_destroy(m);
        return rc;
    }
    *out = m;
    return NBODY_OK;
}

extern "C" int nbody_multi_create_rank(nbody_multi **out, const nbody_multi_config *cfg, int device, int rank, int world_size,
                                       const void *unique_id128)
{
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create_rank: rank outside [0, world_size)");
    if (!unique_id128)
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create_rank: unique id is NULL (nbody_multi_unique_id on rank 0, "
                                                 "then hand the 128 bytes to every rank)");
    nbody_multi *m = nullptr;
    int rc = create_common(out, cfg, world_size, &m);
    if (rc != NBODY_OK)
        return rc;
    if (!rccl(m) && world_size > 1) {
        delete m;
        return mfail(nullptr, NBODY_ERR_INVALID, "nbody_multi_create_rank: ranks in different processes exchange over RCCL only");
    }
    m->ranks.resize(1);
    m->ranks[0].rank = rank;
    m->ranks[0].device = device;
    rc = setup_ranks(m);
    if (rc == NBODY_OK && rccl(m)) {
        ncclUniqueId id;
        std::memcpy(&id, unique_id128, sizeof id);
        rc = init_communicators(m, id);
    }
    if (rc != NBODY_OK) {
        g_multi_create_error = m->err;
        nbody_multi_destroy(m);
        return rc;
    }
    *out = m;
    return NBODY_OK;
}

// Drains the local streams for at most `seconds`: false when something is still running then (a peer that will never answer).
static bool drain_streams(nbody_multi *m, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool busy = false;
        for (Rank &r : m->ranks) {
            (void)hipSetDevice(r.device);
            for (hipStream_t s : {r.compute, r.side, r.comm})
                if (s && hipStreamQuery(s) == hipErrorNotReady)
                    busy = true;
        }
        if (!busy)
            return true;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds)
            return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

static void abort_communicators(nbody_multi *m)
{
    for (Rank &r : m->ranks)
        if (r.nccl) {
            (void)ncclCommAbort(r.nccl);
            r.nccl = nullptr;
        }
    m->failed = true;
}

extern "C" int nbody_multi_destroy(nbody_multi *m)
{
    if (!m)
        return NBODY_OK;
    // No unbounded wait on a peer: after a failure the communicators are already gone; otherwise the streams get the
    // exchange timeout (at most a minute) to drain, and what is still stuck then is aborted instead of destroyed.
    if (!drain_streams(m, m->failed ? 5.0 : std::min(m->timeout_s, 60.0))) {
        abort_communicators(m);
        (void)drain_streams(m, 5.0);
    }
    for (size_t i = 0; i < m->ranks.size(); ++i) {
        Rank &r = m->ranks[i];
        (void)hipSetDevice(r.device);
        if (r.nccl)
            (void)ncclCommDestroy(r.nccl);
        if (r.ctx)
            nbody_destroy(r.ctx);
        for (float *p : {r.pos, r.vel, r.colparts, r.vel_all})
            if (p)
                (void)hipFree(p);
        if (r.order_dev)
            (void)hipFree(r.order_dev);
        if (r.scratch)
            (void)hipFree(r.scratch);
        for (Rank::Span &sp : r.spans) {
            (void)hipEventDestroy(sp.a);
            (void)hipEventDestroy(sp.b);
        }
        for (hipEvent_t e : r.ev_pool)
            (void)hipEventDestroy(e);
        for (hipEvent_t e : {r.ev_start, r.ev_side, r.ev_vel})
            if (e)
                (void)hipEventDestroy(e);
        for (hipEvent_t e : r.ev_hop)
            if (e)
                (void)hipEventDestroy(e);
        for (hipEvent_t e : r.ev_step)
            if (e)
                (void)hipEventDestroy(e);
        for (Channel *c : {&m->ch_pos, &m->ch_col}) {
            if (i < c->ready.size() && c->ready[i]) (void)hipEventDestroy(c->ready[i]);
            if (i < c->done.size() && c->done[i]) (void)hipEventDestroy(c->done[i]);
        }
        for (hipStream_t s : {r.compute, r.side, r.comm})
            if (s)
                (void)hipStreamDestroy(s);
    }
    delete m;
    return NBODY_OK;
}

// ---- waiting, with failure detection ---------------------------------------------------------------------------------

static int timed_out(nbody_multi *m, double waited);

static int poll_async_errors(nbody_multi *m)
{
    for (Rank &r : m->ranks)
        if (r.nccl) {
            ncclResult_t async = ncclSuccess;
            ncclResult_t q = ncclCommGetAsyncError(r.nccl, &async);
            if (q != ncclSuccess || (async != ncclSuccess && async != ncclInProgress)) {
                const ncclResult_t bad = q != ncclSuccess ? q : async;
                const std::string what = "RCCL reported an asynchronous error on rank " + std::to_string(r.rank) + ": " +
                                         ncclGetErrorString(bad) + " (" + ncclGetLastError(r.nccl) + "); the communicators were aborted";
                abort_communicators(m);  // nothing may wait for the peers any more, nbody_multi_destroy included
                return mfail(m, NBODY_ERR_DEVICE, what);
            }
        }
    return NBODY_OK;
}

// Waits until every stream of every local rank has drained, polling RCCL for asynchronous errors; a wait longer than
// the timeout aborts the communicators (a dead or stuck peer) and reports it instead of hanging.
static int wait_all(nbody_multi *m)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        bool busy = false;
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            for (hipStream_t s : {r.compute, r.side, r.comm}) {
                hipError_t e = hipStreamQuery(s);
                if (e == hipErrorNotReady)
                    busy = true;
                else if (e != hipSuccess)
                    return mfail(m, NBODY_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(e));
            }
        }
        if (!busy)
            break;
        if ((++spins & 63u) == 0) {
            int rc = poll_async_errors(m);
            if (rc != NBODY_OK)
                return rc;
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > m->timeout_s)
                return timed_out(m, waited);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 20 : 200));
    }
    for (Rank &r : m->ranks) {
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipGetLastError());
    }
    return poll_async_errors(m);
}

static int timed_out(nbody_multi *m, double waited)
{
    abort_communicators(m);
    return mfail(m, NBODY_ERR_DEVICE, "timed out after " + std::to_string((int)waited) +
                                          " s waiting for the step (a peer rank is gone or stuck); the RCCL "
                                          "communicators were aborted");
}

// Waits until slot `slot` of every local rank's step events has fired: the host never runs more than kStepsInFlight
// steps ahead, so every wait is short and a dead peer is noticed within the timeout whatever the length of the run.
static int wait_step(nbody_multi *m, int slot)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (Rank &r : m->ranks) {
        MHIP(m, hipSetDevice(r.device));
        for (;;) {
            hipError_t e = hipEventQuery(r.ev_step[(size_t)slot]);
            if (e == hipSuccess)
                break;
            if (e != hipErrorNotReady)
                return mfail(m, NBODY_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(e));
            if ((++spins & 63u) == 0) {
                int rc = poll_async_errors(m);
                if (rc != NBODY_OK)
                    return rc;
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (waited > m->timeout_s)
                    return timed_out(m, waited);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 20 : 200));
        }
    }
    return NBODY_OK;
}

extern "C" int nbody_multi_sync(nbody_multi *m)
{
    if (!m)
        return NBODY_ERR_INVALID;
    return wait_all(m);
}

extern "C" int nbody_multi_set_timeout(nbody_multi *m, double seconds)
{
    if (!m || !(seconds > 0))
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_set_timeout: seconds must be > 0");
    m->timeout_s = seconds;
    return NBODY_OK;
}

// ---- measurement: where a rank's step goes besides its kernels ---------------------------------------------------------
// With timing on, the exchanges are bracketed by events on the streams they run on or are waited for on (Span kinds above);
// the totals are sums of event-pair durations, read and reset by nbody_multi_timing_read.  Finished pairs are folded into
// the totals as the list grows, so a long run holds a bounded number of events.

static void fold_spans(Rank &r, bool wait)
{
    size_t done = 0;
    for (; done < r.spans.size(); ++done) {
        Rank::Span &sp = r.spans[done];
        if (wait)
            (void)hipEventSynchronize(sp.b);
        else if (hipEventQuery(sp.b) != hipSuccess)
            break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) {
            r.span_ms[sp.kind] += t;
            ++r.span_count[sp.kind];
        }
        r.ev_pool.push_back(sp.a);
        r.ev_pool.push_back(sp.b);
    }
    r.spans.erase(r.spans.begin(), r.spans.begin() + (ptrdiff_t)done);
}

static hipEvent_t timing_event(Rank &r)
{
    hipEvent_t e = nullptr;
    if (!r.ev_pool.empty()) {
        e = r.ev_pool.back();
        r.ev_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

// Records the start of a span on `stream` (the device must be current); nullptr when timing is off.
static hipEvent_t span_begin(nbody_multi *m, Rank &r, hipStream_t stream)
{
    if (!m->timing)
        return nullptr;
    if (r.spans.size() >= 64)
        fold_spans(r, false);
    hipEvent_t a = timing_event(r);
    if (a)
        (void)hipEventRecord(a, stream);
    return a;
}

static void span_end(nbody_multi *m, Rank &r, hipEvent_t a, int kind, hipStream_t stream)
{
    if (!a)
        return;
    hipEvent_t b = timing_event(r);
    if (!b) {
        r.ev_pool.push_back(a);
        return;
    }
    (void)hipEventRecord(b, stream);
    r.spans.push_back({a, b, kind});
}

extern "C" int nbody_multi_timing_enable(nbody_multi *m, int on)
{
    if (!m)
        return NBODY_ERR_INVALID;
    m->timing = on != 0;
    for (Rank &r : m->ranks)
        if (r.ctx)
            (void)nbody_timing_enable(r.ctx, on);
    return NBODY_OK;
}

extern "C" int nbody_multi_timing_read(nbody_multi *m, int local_index, double *out16)
{
    if (!m || !out16 || local_index < 0 || local_index >= (int)m->ranks.size())
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_timing_read: bad argument");
    Rank &r = m->ranks[(size_t)local_index];
    MHIP(m, hipSetDevice(r.device));
    fold_spans(r, true);
    double k[6] = {0, 0, 0, 0, 0, 0};
    MCTX(m, r, nbody_timing_read_ex(r.ctx, k));
    out16[0] = (double)r.steps;
    out16[1] = r.host_ms;
    out16[2] = k[0];  // force kernels: sum of launch durations
    out16[3] = k[1];
    out16[4] = k[2];  // what runs behind the force pass (summation, update, kicks)
    out16[5] = k[3];
    out16[6] = k[4];  // auxiliary stream (diagonal tiles)
    out16[7] = k[5];
    for (int i = 0; i < 4; ++i) {
        out16[8 + 2 * i] = r.span_ms[i];
        out16[9 + 2 * i] = (double)r.span_count[i];
        r.span_ms[i] = 0;
        r.span_count[i] = 0;
    }
    r.steps = 0;
    r.host_ms = 0;
    return NBODY_OK;
}

// ---- the exchanges -------------------------------------------------------------------------------------------------

typedef float *Rank::*RankBuffer;

// All-gather of equal slices, in place: rank q's slice is [q * slice_floats, (q + 1) * slice_floats) of every rank's
// buffer.  Enqueued on the comm streams behind each rank's `ready` point on its compute stream.
static int start_allgather(nbody_multi *m, Channel &ch, RankBuffer buf, size_t slice_floats)
{
    const size_t nl = m->ranks.size();
    std::vector<hipEvent_t> began(nl, nullptr);
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(ch.ready[i], r.compute));
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        if (rccl(m)) {  // the collective itself orders the ranks: nobody's slice is written before everybody has entered
            EDGE(1, m, hipStreamWaitEvent(r.comm, ch.ready[i], 0));
        } else {        // peer copies write into the others' buffers: wait until they have stopped reading them
            for (size_t j = 0; j < nl; ++j)
                MHIP(m, hipStreamWaitEvent(r.comm, ch.ready[j], 0));
        }
        began[i] = span_begin(m, r, r.comm);
    }
    if (rccl(m)) {
        MNCCL(m, ncclGroupStart());
        for (Rank &r : m->ranks) {
            float *b = r.*buf;
            ncclResult_t q = ncclAllGather(b + (size_t)r.rank * slice_floats, b, slice_floats, ncclFloat, r.nccl, r.comm);
            if (q != ncclSuccess) {
                (void)ncclGroupEnd();
                return mfail(m, NBODY_ERR_DEVICE, std::string("ncclAllGather: RCCL: ") + ncclGetErrorString(q));
            }
        }
        MNCCL(m, ncclGroupEnd());
    } else {
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            for (Rank &d : m->ranks)
                if (d.rank != r.rank)
                    MHIP(m, hipMemcpyPeerAsync(d.*buf + (size_t)r.rank * slice_floats, d.device,
                                               r.*buf + (size_t)r.rank * slice_floats, r.device, slice_floats * sizeof(float),
                                               r.comm));
        }
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(ch.done[i], r.comm));
        span_end(m, r, began[i], kSpanPosComm, r.comm);
    }
    return NBODY_OK;
}

// Orders `stream` of local rank i behind the arrival of every other rank's slice -- and behind the departure of its own:
// the consumer is also the next writer of the rank's slice (update, kick-drift, the next column-side sums), and a peer copy
// that is still reading it runs on the rank's OWN communication stream (RCCL: done[i] marks the whole collective).
static int wait_allgather(nbody_multi *m, Channel &ch, size_t i, hipStream_t stream)
{
    Rank &r = m->ranks[i];
    MHIP(m, hipSetDevice(r.device));
    if (rccl(m)) {
        EDGE(2, m, hipStreamWaitEvent(stream, ch.done[i], 0));
    } else {
        for (size_t j = 0; j < m->ranks.size(); ++j)
            MHIP(m, hipStreamWaitEvent(stream, ch.done[j], 0));
    }
    return NBODY_OK;
}

// The position ring: P - 1 hops on the comm streams; ev_hop[h] of a rank marks the arrival of hop h at that rank (RCCL)
// or the departure of its hop-h copy towards rank + 1 (peer copies).
static int start_ring(nbody_multi *m)
{
    const size_t nl = m->ranks.size();
    const int P = m->world;
    const size_t chunk_floats = 4 * (size_t)m->chunk;
    std::vector<hipEvent_t> began(nl, nullptr);
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(m->ch_pos.ready[i], r.compute));
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        if (rccl(m)) {
            EDGE(5, m, hipStreamWaitEvent(r.comm, m->ch_pos.ready[i], 0));
        } else {
            for (size_t j = 0; j < nl; ++j)
                MHIP(m, hipStreamWaitEvent(r.comm, m->ch_pos.ready[j], 0));
        }
        began[i] = span_begin(m, r, r.comm);
    }
    for (int h = 1; h < P; ++h) {
        if (rccl(m)) {
            MNCCL(m, ncclGroupStart());
            for (Rank &r : m->ranks) {
                int send_c = 0, recv_c = 0;
                nbody_multi_ring_schedule(r.rank, P, h, &send_c, &recv_c);
                ncclResult_t q = ncclSend(r.pos + (size_t)send_c * chunk_floats, chunk_floats, ncclFloat, (r.rank + 1) % P, r.nccl,
                                          r.comm);
                if (q == ncclSuccess)
                    q = ncclRecv(r.pos + (size_t)recv_c * chunk_floats, chunk_floats, ncclFloat, (r.rank + P - 1) % P, r.nccl,
                                 r.comm);
                if (q != ncclSuccess) {
                    (void)ncclGroupEnd();
                    return mfail(m, NBODY_ERR_DEVICE, std::string("ncclSend/ncclRecv: RCCL: ") + ncclGetErrorString(q));
                }
            }
            MNCCL(m, ncclGroupEnd());
        } else {
            for (size_t i = 0; i < nl; ++i) {
                Rank &r = m->ranks[i];
                Rank &next = m->ranks[(i + 1) % nl];  // all ranks are local here, in rank order
                Rank &prev = m->ranks[(i + nl - 1) % nl];
                int send_c = 0;
                nbody_multi_ring_schedule(r.rank, P, h, &send_c, nullptr);
                MHIP(m, hipSetDevice(r.device));
                if (h > 1)  // the chunk passed on arrived with the previous hop, on the previous rank's stream
                    MHIP(m, hipStreamWaitEvent(r.comm, prev.ev_hop[(size_t)h - 1], 0));
                MHIP(m, hipMemcpyPeerAsync(next.pos + (size_t)send_c * chunk_floats, next.device,
                                           r.pos + (size_t)send_c * chunk_floats, r.device, chunk_floats * sizeof(float), r.comm));
            }
        }
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            MHIP(m, hipEventRecord(r.ev_hop[(size_t)h], r.comm));
        }
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(m->ch_pos.done[i], r.comm));
        span_end(m, r, began[i], kSpanPosComm, r.comm);
    }
    return NBODY_OK;
}

static int wait_ring_hop(nbody_multi *m, size_t i, int h, hipStream_t stream)
{
    Rank &r = m->ranks[i];
    const size_t nl = m->ranks.size();
    MHIP(m, hipSetDevice(r.device));
    hipEvent_t e = rccl(m) ? r.ev_hop[(size_t)h] : m->ranks[(i + nl - 1) % nl].ev_hop[(size_t)h];
    if (rccl(m))
        EDGE(3, m, hipStreamWaitEvent(stream, e, 0));
    else
        MHIP(m, hipStreamWaitEvent(stream, e, 0));
    return NBODY_OK;
}

static int exchange_own_rows(nbody_multi *m)
{
    if (m->world == 1)
        return NBODY_OK;
    int rc = m->cfg.exchange == NBODY_EXCHANGE_RING ? start_ring(m) : start_allgather(m, m->ch_pos, &Rank::pos, 4 * (size_t)m->chunk);
    if (rc == NBODY_OK)
        m->exchange_in_flight = true;
    return rc;
}

// ---- the step --------------------------------------------------------------------------------------------------------

// Partial sums of every local rank's rows from every column: the own chunk first (it needs no remote data and runs
// beside the exchange in flight), the other chunks as they become current.
static int forces_all_columns(nbody_multi *m, float softening)
{
    const size_t nl = m->ranks.size();
    const int P = m->world;
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        const int64_t lo = (int64_t)r.rank * m->chunk;
        MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
        if (P > 1) {
            MHIP(m, hipSetDevice(r.device));
            MHIP(m, hipEventRecord(r.ev_start, r.compute));  // the previous update (the reader of the partial sums) is done
        }
        MCTX(m, r, nbody_forces(r.ctx, r.pos, lo, m->chunk, softening));
    }
    if (P == 1)
        return NBODY_OK;
    if (m->cfg.exchange == NBODY_EXCHANGE_RING && m->exchange_in_flight) {
        for (int h = 1; h < P; ++h)
            for (size_t i = 0; i < nl; ++i) {
                Rank &r = m->ranks[i];
                int recv_c = 0;
                nbody_multi_ring_schedule(r.rank, P, h, nullptr, &recv_c);
                MHIP(m, hipSetDevice(r.device));
                hipEvent_t t0 = span_begin(m, r, r.compute);
                int rc = wait_ring_hop(m, i, h, r.compute);
                if (rc != NBODY_OK)
                    return rc;
                span_end(m, r, t0, kSpanPosWait, r.compute);
                MCTX(m, r, nbody_forces(r.ctx, r.pos, (int64_t)recv_c * m->chunk, m->chunk, softening));
            }
        // the next writer of the rank's own rows (the update behind these launches) follows the rank's own sends as well
        for (size_t i = 0; i < nl; ++i) {
            MHIP(m, hipSetDevice(m->ranks[i].device));
            EDGE(4, m, hipStreamWaitEvent(m->ranks[i].compute, m->ch_pos.done[i], 0));
        }
        m->exchange_in_flight = false;
        return NBODY_OK;
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        const int64_t lo = (int64_t)r.rank * m->chunk;
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipStreamWaitEvent(r.side, r.ev_start, 0));
        if (m->exchange_in_flight) {
            hipEvent_t t0 = span_begin(m, r, r.side);  // fires when the previous update is over: the launch could start here
            int rc = wait_allgather(m, m->ch_pos, i, r.side);
            if (rc != NBODY_OK)
                return rc;
            span_end(m, r, t0, kSpanPosWait, r.side);
        }
        // all other chunks in one launch on the second stream: its workgroups fill the CUs the first launch's tail leaves idle
        MCTX(m, r, nbody_set_stream(r.ctx, r.side));
        MCTX(m, r, nbody_forces_complement(r.ctx, r.pos, lo, m->chunk, softening));
        MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(r.ev_side, r.side));
        MHIP(m, hipStreamWaitEvent(r.compute, r.ev_side, 0));
    }
    m->exchange_in_flight = false;
    return NBODY_OK;
}

// Pair-once mode: rank q's update needs colparts[g][rows of q] for every group g, and rank r holds colparts[g][every body]
// for its own groups g.  So r hands q the rows-of-q segment of each of its 8 / P groups and receives q's in return:
// (P - 1) x 8 / P segments of C x 16 B per rank (14 MiB at N = 2^20, P = 8) -- an eighth of gathering whole group slices,
// all point to point, which is what xGMI is.  The segments land where sym_combine_kernel reads them.
static int exchange_column_sums(nbody_multi *m)
{
    const size_t nl = m->ranks.size();
    const int P = m->world, per_rank = NBODY_SYM_GROUPS / P;
    const size_t seg = 4 * (size_t)m->chunk, group = 4 * (size_t)m->n_padded;  // floats
    Channel &ch = m->ch_col;
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(ch.ready[i], r.compute));  // own groups summed; the previous step's reader is done
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        if (rccl(m)) {  // a segment is written once its owner has posted the receive, behind its own `ready`
            EDGE(6, m, hipStreamWaitEvent(r.comm, ch.ready[i], 0));
        } else {        // peer copies write into the others' buffers: wait until they have stopped reading them
            for (size_t j = 0; j < nl; ++j)
                MHIP(m, hipStreamWaitEvent(r.comm, ch.ready[j], 0));
        }
    }
    if (rccl(m)) {
        MNCCL(m, ncclGroupStart());
        for (Rank &r : m->ranks)
            for (int q = 0; q < P; ++q) {
                if (q == r.rank)
                    continue;
                for (int k = 0; k < per_rank; ++k) {
                    ncclResult_t e = ncclSend(r.colparts + (size_t)(r.rank * per_rank + k) * group + (size_t)q * seg, seg, ncclFloat,
                                              q, r.nccl, r.comm);
                    if (e == ncclSuccess)
                        e = ncclRecv(r.colparts + (size_t)(q * per_rank + k) * group + (size_t)r.rank * seg, seg, ncclFloat, q,
                                     r.nccl, r.comm);
                    if (e != ncclSuccess) {
                        (void)ncclGroupEnd();
                        return mfail(m, NBODY_ERR_DEVICE, std::string("ncclSend/ncclRecv: RCCL: ") + ncclGetErrorString(e));
                    }
                }
            }
        MNCCL(m, ncclGroupEnd());
    } else {
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            for (Rank &d : m->ranks) {
                if (d.rank == r.rank)
                    continue;
                for (int k = 0; k < per_rank; ++k) {
                    const size_t at = (size_t)(r.rank * per_rank + k) * group + (size_t)d.rank * seg;
                    MHIP(m, hipMemcpyPeerAsync(d.colparts + at, d.device, r.colparts + at, r.device, seg * sizeof(float), r.comm));
                }
            }
        }
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(ch.done[i], r.comm));
    }
    return NBODY_OK;
}

static int sum_forces(nbody_multi *m)
{
    if (!pair_once(m) || m->world == 1)
        return NBODY_OK;  // one context: nbody_update / nbody_kdk_* run the reduction themselves
    std::vector<hipEvent_t> began(m->ranks.size(), nullptr);
    for (size_t i = 0; i < m->ranks.size(); ++i) {
        Rank &r = m->ranks[i];
        MCTX(m, r, nbody_sym_reduce(r.ctx));
        MHIP(m, hipSetDevice(r.device));
        began[i] = span_begin(m, r, r.compute);
    }
    int rc = exchange_column_sums(m);
    // beside the exchange: the row-side sums need the rank's own partial sums only (0.2 ms at P = 8, N = 2^20, against ~0.1 ms
    // of wire time: the exchange hides behind them)
    for (size_t i = 0; rc == NBODY_OK && i < m->ranks.size(); ++i)
        MCTX(m, m->ranks[i], nbody_sym_rowsum(m->ranks[i].ctx));
    for (size_t i = 0; rc == NBODY_OK && i < m->ranks.size(); ++i) {
        rc = wait_allgather(m, m->ch_col, i, m->ranks[i].compute);
        span_end(m, m->ranks[i], began[i], kSpanColumns, m->ranks[i].compute);
    }
    return rc;
}

static int step_enqueue(nbody_multi *m, float dt, float softening);

static int step_async(nbody_multi *m, float dt, float softening)
{
    if (m->failed)
        return mfail(m, NBODY_ERR_DEVICE, "nbody_multi_step: an earlier exchange failed and the communicators were aborted (" +
                                              m->err + ")");
    if (!m->have_state && m->n_padded)
        return mfail(m, NBODY_ERR_STATE, "nbody_multi_step: call nbody_multi_set_state first");
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = step_enqueue(m, dt, softening);
    if (m->timing) {
        const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (Rank &r : m->ranks) {
            r.host_ms += ms / (double)m->ranks.size();  // the host thread enqueues the local ranks one after the other
            ++r.steps;
        }
    }
    return rc;
}

static int step_enqueue(nbody_multi *m, float dt, float softening)
{
    int rc;
    if (m->cfg.integrator == NBODY_INTEGRATOR_KDK) {
        // velocity Verlet: the drifted rows are exchanged BEFORE the forces; the own-chunk launch still runs beside the exchange
        if (!m->kdk_ready) {
            rc = forces_all_columns(m, softening);
            if (rc == NBODY_OK)
                rc = sum_forces(m);
            if (rc != NBODY_OK)
                return rc;
            for (Rank &r : m->ranks)
                MCTX(m, r, nbody_kdk_prepare(r.ctx));
            m->kdk_ready = true;
        }
        for (Rank &r : m->ranks)
            MCTX(m, r, nbody_kdk_kick_drift(r.ctx, r.pos, r.vel, dt));
        rc = exchange_own_rows(m);
        if (rc == NBODY_OK)
            rc = forces_all_columns(m, softening);
        if (rc == NBODY_OK)
            rc = sum_forces(m);
        if (rc != NBODY_OK)
            return rc;
        for (Rank &r : m->ranks)
            MCTX(m, r, nbody_kdk_kick(r.ctx, r.vel, dt));
        return poll_async_errors(m);
    }
    rc = forces_all_columns(m, softening);
    if (rc == NBODY_OK)
        rc = sum_forces(m);
    if (rc != NBODY_OK)
        return rc;
    for (Rank &r : m->ranks)
        MCTX(m, r, nbody_update(r.ctx, r.pos, r.vel, dt));
    rc = exchange_own_rows(m);
    return rc == NBODY_OK ? poll_async_errors(m) : rc;
}

// Replica current on every rank: an exchange still in flight is simply waited for (its data are not consumed twice).
static int settle(nbody_multi *m)
{
    int rc = wait_all(m);
    m->exchange_in_flight = false;  // everything has landed: the next force launches need not wait for anything
    return rc;
}

// The layout decays as the bodies move (N = 2^20 Plummer sphere: half of the gain is gone after ~300 steps of dt = 1e-3,
// profiles/r02_longrun_morton_decay_n1048576.txt): with a reorder period the step that is due first refreshes it.
static int reorder_if_due(nbody_multi *m)
{
    if (m->reorder_period > 0 && m->cfg.body_order == NBODY_ORDER_MORTON && m->steps_since_order >= m->reorder_period)
        return nbody_multi_reorder(m);
    return NBODY_OK;
}

extern "C" int nbody_multi_step_async(nbody_multi *m, float dt, float softening)
{
    if (!m)
        return NBODY_ERR_INVALID;
    int rc = reorder_if_due(m);
    if (rc == NBODY_OK)
        rc = step_async(m, dt, softening);
    m->steps_since_order += rc == NBODY_OK;
    return rc;
}

extern "C" int nbody_multi_step(nbody_multi *m, float dt, float softening)
{
    if (!m)
        return NBODY_ERR_INVALID;
    int rc = nbody_multi_step_async(m, dt, softening);
    return rc == NBODY_OK ? settle(m) : rc;
}

extern "C" int nbody_multi_step_n(nbody_multi *m, int k, float dt, float softening)
{
    if (!m || k < 0)
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_step_n: bad argument");
    for (int s = 0; s < k; ++s) {
        const int slot = s % kStepsInFlight;
        int rc = s >= kStepsInFlight ? wait_step(m, slot) : NBODY_OK;  // step s - kStepsInFlight has finished
        if (rc == NBODY_OK)
            rc = reorder_if_due(m);
        if (rc == NBODY_OK)
            rc = step_async(m, dt, softening);
        if (rc != NBODY_OK)
            return rc;
        ++m->steps_since_order;
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            MHIP(m, hipEventRecord(r.ev_step[(size_t)slot], r.compute));
        }
    }
    return settle(m);
}

// ---- state in and out ------------------------------------------------------------------------------------------------
// NBODY_ORDER_MORTON: the replicas hold the bodies in nbody_morton_order of the positions; Rank::order_dev says, on every
// rank's own device, which of the caller's bodies sits in slot k.  The layout is laid and refreshed ON THE DEVICE
// (csrc/nbody_order.hip): every rank sorts its own replica and gets the same permutation, the velocity rows are re-dealt
// through one gather of all rows, and the host sees the order only when it asks for it (download, nbody_multi_order).

static int upload_softening(nbody_multi *m);

// The host copy of the order, fetched when it is stale.
static int host_order(nbody_multi *m)
{
    if (m->cfg.body_order != NBODY_ORDER_MORTON || !m->n_bodies) {
        m->order.clear();
        return NBODY_OK;
    }
    if (m->order_host_valid)
        return NBODY_OK;
    Rank &r0 = m->ranks[0];
    m->order.resize((size_t)m->n_bodies);
    MHIP(m, hipSetDevice(r0.device));
    MHIP(m, hipStreamSynchronize(r0.compute));
    MHIP(m, hipMemcpy(m->order.data(), r0.order_dev, sizeof(int64_t) * (size_t)m->n_bodies, hipMemcpyDeviceToHost));
    m->order_host_valid = true;
    return NBODY_OK;
}

// Every rank's velocity rows into every local rank's vel_all, on the communication streams; the compute streams are ordered
// behind the arrival.  The state must be settled (nothing in flight writes a velocity).
static int gather_all_velocities(nbody_multi *m)
{
    const size_t nl = m->ranks.size(), chunk_floats = 4 * (size_t)m->chunk;
    for (Rank &r : m->ranks) {
        MHIP(m, hipSetDevice(r.device));
        if (!r.vel_all)
            MHIP(m, hipMalloc((void **)&r.vel_all, sizeof(float) * 4 * (size_t)m->n_padded));
    }
    if (rccl(m)) {
        MNCCL(m, ncclGroupStart());
        for (Rank &r : m->ranks) {
            ncclResult_t q = ncclAllGather(r.vel, r.vel_all, chunk_floats, ncclFloat, r.nccl, r.comm);
            if (q != ncclSuccess) {
                (void)ncclGroupEnd();
                return mfail(m, NBODY_ERR_DEVICE, std::string("ncclAllGather: RCCL: ") + ncclGetErrorString(q));
            }
        }
        MNCCL(m, ncclGroupEnd());
    } else {
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            for (Rank &d : m->ranks)
                MHIP(m, hipMemcpyPeerAsync(d.vel_all + (size_t)r.rank * chunk_floats, d.device, r.vel, r.device,
                                           chunk_floats * sizeof(float), r.comm));
        }
    }
    for (Rank &r : m->ranks) {
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(r.ev_vel, r.comm));
    }
    for (size_t i = 0; i < nl; ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        if (rccl(m)) {
            MHIP(m, hipStreamWaitEvent(r.compute, r.ev_vel, 0));
        } else {
            for (size_t j = 0; j < nl; ++j)
                MHIP(m, hipStreamWaitEvent(r.compute, m->ranks[j].ev_vel, 0));
        }
    }
    return NBODY_OK;
}

// Applies the permutation every local rank's context holds (nbody_order_compute / nbody_order_set: the same on every rank)
// to the whole state: the replica in place, the rank's velocity rows out of a gather of all rows, the rank's softening
// copy, and the order array (composed).  The state must be settled.  Asynchronous on the compute streams.
static int apply_order(nbody_multi *m)
{
    const int64_t n = m->n_bodies;
    for (Rank &r : m->ranks) {
        MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
        MCTX(m, r, nbody_order_gather(r.ctx, r.pos, r.pos, 0, n, 4));
    }
    if (m->world == 1) {
        Rank &r = m->ranks[0];
        MCTX(m, r, nbody_order_gather(r.ctx, r.vel, r.vel, 0, n, 4));
    } else {
        int rc = gather_all_velocities(m);
        if (rc != NBODY_OK)
            return rc;
        for (Rank &r : m->ranks)  // beyond n the permutation is the identity: the padding rows copy themselves
            MCTX(m, r, nbody_order_gather(r.ctx, r.vel, r.vel_all, (int64_t)r.rank * m->chunk, m->chunk, 4));
    }
    for (Rank &r : m->ranks) {
        MCTX(m, r, nbody_order_permute_softening(r.ctx));
        if (r.order_dev)
            MCTX(m, r, nbody_order_gather(r.ctx, r.order_dev, r.order_dev, 0, n, 2));
        MCTX(m, r, nbody_invalidate_forces(r.ctx));
    }
    m->order_host_valid = false;
    m->kdk_ready = false;
    m->exchange_in_flight = false;
    return NBODY_OK;
}

// A new curve through the positions the replicas hold now.
static int reorder_device(nbody_multi *m)
{
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    std::vector<hipEvent_t> began(m->ranks.size(), nullptr);
    for (size_t i = 0; i < m->ranks.size(); ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        began[i] = span_begin(m, r, r.compute);
        MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
        MCTX(m, r, nbody_order_compute(r.ctx, r.pos, m->n_bodies, r.order_dev));  // ties follow the caller's indices
    }
    rc = apply_order(m);
    for (size_t i = 0; rc == NBODY_OK && i < m->ranks.size(); ++i) {
        Rank &r = m->ranks[i];
        MHIP(m, hipSetDevice(r.device));
        span_end(m, r, began[i], kSpanReorder, r.compute);
    }
    m->steps_since_order = 0;
    return rc;
}

static bool morton(const nbody_multi *m) { return m->cfg.body_order == NBODY_ORDER_MORTON && m->n_bodies > 0; }

// Host rows of n_bodies float4 (the caller's order) into a padded staging vector: the padding = zero-mass bodies at the
// origin, the reference's own device (kernel.cu:265-277): they add exactly 0.
static void pad_rows(const nbody_multi *m, const float *host, std::vector<float> &out)
{
    out.assign(4 * (size_t)m->n_padded, 0.f);
    if (m->n_bodies)
        std::memcpy(out.data(), host, sizeof(float) * 4 * (size_t)m->n_bodies);
}

static int upload_positions(nbody_multi *m, const std::vector<float> &pos)
{
    for (Rank &r : m->ranks) {
        if (!m->n_padded)
            break;
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipMemcpyAsync(r.pos, pos.data(), sizeof(float) * pos.size(), hipMemcpyHostToDevice, r.compute));
        MHIP(m, hipStreamSynchronize(r.compute));
        MCTX(m, r, nbody_invalidate_forces(r.ctx));
    }
    return NBODY_OK;
}

static int set_state_impl(nbody_multi *m, const float *host_pos, const float *host_vel)
{
    if (!m || ((!host_pos || !host_vel) && m->n_bodies))
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_set_state: NULL argument");
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    std::vector<float> pos, vel;
    pad_rows(m, host_pos, pos);
    pad_rows(m, host_vel, vel);
    rc = upload_positions(m, pos);
    if (rc != NBODY_OK)
        return rc;
    for (Rank &r : m->ranks) {
        if (!m->n_padded)
            break;
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipMemcpyAsync(r.vel, vel.data() + 4 * (size_t)r.rank * (size_t)m->chunk, sizeof(float) * 4 * (size_t)m->chunk,
                               hipMemcpyHostToDevice, r.compute));
        MHIP(m, hipStreamSynchronize(r.compute));
        if (r.order_dev) {
            MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
            MCTX(m, r, nbody_order_identity(r.ctx, r.order_dev, m->n_bodies));
        }
    }
    m->order_host_valid = false;
    m->kdk_ready = false;
    m->have_state = true;
    m->steps_since_order = 0;
    if (m->eps_on)
        rc = upload_softening(m);  // the caller's order, like the state just uploaded
    if (rc == NBODY_OK && morton(m))
        rc = reorder_device(m);    // the curve through these positions; velocities and softening lengths follow their bodies
    return rc;
}

extern "C" int nbody_multi_set_state(nbody_multi *m, const float *host_pos, const float *host_vel)
{
    return guarded(m, [&] { return set_state_impl(m, host_pos, host_vel); });
}

// setParticlesPosition / setParticlesVelocity (kernel.cu:163-188) are independent copies in the reference: so are these.
// New positions keep every body's velocity and softening length (NBODY_ORDER_MORTON: each body's new position goes to the
// slot the body is in, then the new curve is laid through them); new velocities leave the positions alone.
static int set_positions_impl(nbody_multi *m, const float *host_pos)
{
    if (!m || (!host_pos && m->n_bodies))
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_set_positions: NULL argument");
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    std::vector<float> pos;
    pad_rows(m, host_pos, pos);
    if (morton(m) && m->have_state) {  // body order[k] of the new positions into slot k, where its velocity already is
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            if (!r.vel_all)
                MHIP(m, hipMalloc((void **)&r.vel_all, sizeof(float) * 4 * (size_t)m->n_padded));
            MHIP(m, hipMemcpyAsync(r.vel_all, pos.data(), sizeof(float) * pos.size(), hipMemcpyHostToDevice, r.compute));
            MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
            MCTX(m, r, nbody_order_set(r.ctx, r.order_dev, m->n_bodies, 0));
            MCTX(m, r, nbody_order_gather(r.ctx, r.pos, r.vel_all, 0, m->n_bodies, 4));
            MHIP(m, hipStreamSynchronize(r.compute));
            MCTX(m, r, nbody_invalidate_forces(r.ctx));
        }
    } else {
        rc = upload_positions(m, pos);
    }
    m->kdk_ready = false;
    m->have_state = true;
    m->steps_since_order = 0;
    if (rc == NBODY_OK && morton(m))
        rc = reorder_device(m);
    return rc;
}

static int set_velocities_impl(nbody_multi *m, const float *host_vel)
{
    if (!m || (!host_vel && m->n_bodies))
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_set_velocities: NULL argument");
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    std::vector<float> vel;
    pad_rows(m, host_vel, vel);
    for (Rank &r : m->ranks) {
        if (!m->n_padded)
            break;
        MHIP(m, hipSetDevice(r.device));
        const size_t lo = (size_t)r.rank * (size_t)m->chunk;
        if (morton(m)) {  // all rows in the caller's order, then this rank's rows through the order array
            if (!r.vel_all)
                MHIP(m, hipMalloc((void **)&r.vel_all, sizeof(float) * 4 * (size_t)m->n_padded));
            MHIP(m, hipMemcpyAsync(r.vel_all, vel.data(), sizeof(float) * vel.size(), hipMemcpyHostToDevice, r.compute));
            MCTX(m, r, nbody_set_stream(r.ctx, r.compute));
            MCTX(m, r, nbody_order_set(r.ctx, r.order_dev, m->n_bodies, 0));
            MCTX(m, r, nbody_order_gather(r.ctx, r.vel, r.vel_all, (int64_t)lo, m->chunk, 4));
        } else {
            MHIP(m, hipMemcpyAsync(r.vel, vel.data() + 4 * lo, sizeof(float) * 4 * (size_t)m->chunk, hipMemcpyHostToDevice, r.compute));
        }
        MHIP(m, hipStreamSynchronize(r.compute));
    }
    return NBODY_OK;
}

extern "C" int nbody_multi_set_positions(nbody_multi *m, const float *host_pos)
{
    return guarded(m, [&] { return set_positions_impl(m, host_pos); });
}

extern "C" int nbody_multi_set_velocities(nbody_multi *m, const float *host_vel)
{
    return guarded(m, [&] { return set_velocities_impl(m, host_vel); });
}

static int upload_softening(nbody_multi *m)  // m->eps_caller (the caller's order, or empty) in the order of the replicas
{
    std::vector<float> eps;
    const bool on = !m->eps_caller.empty() || (m->eps_on && m->n_bodies == 0);
    if (on) {
        eps.assign((size_t)m->n_padded, 0.f);
        if (morton(m)) {
            int rc = host_order(m);
            if (rc != NBODY_OK)
                return rc;
            for (int64_t k = 0; k < m->n_bodies; ++k)
                eps[(size_t)k] = m->eps_caller[(size_t)m->order[(size_t)k]];
        } else if (m->n_bodies) {
            std::memcpy(eps.data(), m->eps_caller.data(), sizeof(float) * (size_t)m->n_bodies);
        }
    }
    for (Rank &r : m->ranks)
        MCTX(m, r, nbody_upload_particle_softening(r.ctx, on ? eps.data() : nullptr));
    m->kdk_ready = false;
    return NBODY_OK;
}

extern "C" int nbody_multi_set_particle_softening(nbody_multi *m, const float *host_eps)
{
    if (!m)
        return NBODY_ERR_INVALID;
    return guarded(m, [&] {
        int rc = settle(m);
        if (rc != NBODY_OK)
            return rc;
        m->eps_on = host_eps != nullptr;
        if (host_eps)
            m->eps_caller.assign(host_eps, host_eps + m->n_bodies);
        else
            m->eps_caller.clear();
        return upload_softening(m);
    });
}

extern "C" int nbody_multi_set_reorder_period(nbody_multi *m, int64_t steps)
{
    if (!m || steps < 0)
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_set_reorder_period: steps must be >= 0");
    m->reorder_period = steps;
    return NBODY_OK;
}

extern "C" int nbody_multi_reorder(nbody_multi *m)
{
    if (!m)
        return NBODY_ERR_INVALID;
    if (!morton(m) || !m->have_state) {
        m->steps_since_order = 0;
        return NBODY_OK;  // nothing to refresh
    }
    return guarded(m, [&] { return reorder_device(m); });
}

static int download_device_order(nbody_multi *m, float *host_pos, float *host_vel);

extern "C" int nbody_multi_download(nbody_multi *m, float *host_pos, float *host_vel)
{
    if (!m)
        return NBODY_ERR_INVALID;
    return guarded(m, [&] {
        if (!morton(m))
            return download_device_order(m, host_pos, host_vel);
        // the replicas hold the bodies in nbody_morton_order: back to the caller's
        std::vector<float> pos(host_pos ? 4 * (size_t)m->n_bodies : 0), vel(host_vel ? 4 * (size_t)m->n_bodies : 0);
        int rc = download_device_order(m, host_pos ? pos.data() : nullptr, host_vel ? vel.data() : nullptr);
        if (rc == NBODY_OK)
            rc = host_order(m);
        if (rc != NBODY_OK)
            return rc;
        for (int64_t k = 0; k < m->n_bodies; ++k) {
            if (host_pos)
                std::memcpy(host_pos + 4 * (size_t)m->order[(size_t)k], &pos[4 * (size_t)k], sizeof(float) * 4);
            if (host_vel)
                std::memcpy(host_vel + 4 * (size_t)m->order[(size_t)k], &vel[4 * (size_t)k], sizeof(float) * 4);
        }
        return (int)NBODY_OK;
    });
}

extern "C" int nbody_multi_order(nbody_multi *m, int64_t *perm)
{
    if (!m || (!perm && m->n_bodies))
        return NBODY_ERR_INVALID;
    if (!m->have_state && m->n_bodies)
        return mfail(m, NBODY_ERR_STATE, "nbody_multi_order: no state was ever set");
    return guarded(m, [&] {
        int rc = settle(m);
        if (rc == NBODY_OK)
            rc = host_order(m);
        if (rc != NBODY_OK)
            return rc;
        for (int64_t k = 0; k < m->n_bodies; ++k)
            perm[k] = m->order.empty() ? k : m->order[(size_t)k];
        return (int)NBODY_OK;
    });
}

static int download_device_order(nbody_multi *m, float *host_pos, float *host_vel)
{
    if (!m->have_state && m->n_padded)
        return mfail(m, NBODY_ERR_STATE, "nbody_multi_download: no state was ever set");
    int rc = settle(m);
    if (rc != NBODY_OK || !m->n_bodies)
        return rc;
    Rank &r0 = m->ranks[0];
    MHIP(m, hipSetDevice(r0.device));
    if (host_pos)
        MHIP(m, hipMemcpy(host_pos, r0.pos, sizeof(float) * 4 * (size_t)m->n_bodies, hipMemcpyDeviceToHost));
    if (!host_vel)
        return NBODY_OK;
    auto copy_rows = [&](const float *dev_rows, int rank) -> hipError_t {  // the real bodies among this rank's rows
        const int64_t lo = (int64_t)rank * m->chunk, hi = std::min(lo + m->chunk, m->n_bodies);
        if (hi <= lo)
            return hipSuccess;
        return hipMemcpy(host_vel + 4 * (size_t)lo, dev_rows, sizeof(float) * 4 * (size_t)(hi - lo), hipMemcpyDeviceToHost);
    };
    if (all_local(m)) {
        for (Rank &r : m->ranks) {
            MHIP(m, hipSetDevice(r.device));
            MHIP(m, copy_rows(r.vel, r.rank));
        }
        return NBODY_OK;
    }
    rc = gather_all_velocities(m);  // collective: every process downloads
    if (rc == NBODY_OK)
        rc = wait_all(m);
    if (rc != NBODY_OK)
        return rc;
    MHIP(m, hipSetDevice(r0.device));
    MHIP(m, hipMemcpy(host_vel, r0.vel_all, sizeof(float) * 4 * (size_t)m->n_bodies, hipMemcpyDeviceToHost));
    return NBODY_OK;
}

// Sums `n` doubles over all ranks (in rank order inside this process; ncclSum across processes).
static int allreduce_sum(nbody_multi *m, double *v, int n)
{
    if (all_local(m) || m->world == 1)
        return NBODY_OK;
    Rank &r = m->ranks[0];
    MHIP(m, hipSetDevice(r.device));
    MHIP(m, hipMemcpyAsync(r.scratch, v, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, r.comm));
    MNCCL(m, ncclAllReduce(r.scratch, r.scratch, (size_t)n, ncclDouble, ncclSum, r.nccl, r.comm));
    MHIP(m, hipMemcpyAsync(v, r.scratch, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, r.comm));
    return wait_all(m);
}

extern "C" int nbody_multi_energy(nbody_multi *m, float softening, double *out3)
{
    if (!m || !out3)
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_energy: NULL argument");
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    double sum[4] = {0, 0, 0, 0};
    for (Rank &r : m->ranks) {
        double e[3];
        MCTX(m, r, nbody_energy(r.ctx, r.pos, r.vel, softening, e));
        for (int k = 0; k < 3; ++k)
            sum[k] += e[k];
    }
    rc = allreduce_sum(m, sum, 4);
    for (int k = 0; k < 3; ++k)
        out3[k] = sum[k];
    return rc;
}

extern "C" int nbody_multi_momentum(nbody_multi *m, double *out4)
{
    if (!m || !out4)
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_momentum: NULL argument");
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    double sum[4] = {0, 0, 0, 0};
    for (Rank &r : m->ranks) {
        double p[4];
        MCTX(m, r, nbody_momentum(r.ctx, r.pos, r.vel, p));
        for (int k = 0; k < 4; ++k)
            sum[k] += p[k];
    }
    rc = allreduce_sum(m, sum, 4);
    for (int k = 0; k < 4; ++k)
        out4[k] = sum[k];
    return rc;
}

// out2 = {smallest, largest} checksum of the position replicas over ALL ranks: equal when every rank holds the same bits.
extern "C" int nbody_multi_replica_checksums(nbody_multi *m, uint64_t *out2)
{
    if (!m || !out2)
        return mfail(m, NBODY_ERR_INVALID, "nbody_multi_replica_checksums: NULL argument");
    int rc = settle(m);
    if (rc != NBODY_OK)
        return rc;
    unsigned long long lo = ~0ull, hi = 0ull;
    for (Rank &r : m->ranks) {
        unsigned long long s = 0;
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipMemsetAsync(r.scratch, 0, sizeof(unsigned long long), r.compute));
        if (m->n_padded)
            hipLaunchKernelGGL(nbody_checksum_kernel, dim3(1024), dim3(256), 0, r.compute, reinterpret_cast<const unsigned *>(r.pos),
                               4 * (size_t)m->n_padded, r.scratch);
        MHIP(m, hipGetLastError());
        MHIP(m, hipMemcpyAsync(&s, r.scratch, sizeof s, hipMemcpyDeviceToHost, r.compute));
        MHIP(m, hipStreamSynchronize(r.compute));
        lo = std::min(lo, s);
        hi = std::max(hi, s);
    }
    if (!all_local(m) && m->world > 1) {
        Rank &r = m->ranks[0];
        unsigned long long v[2] = {hi, ~lo};  // max over ranks of {s, ~s} = {max s, ~min s}
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipMemcpyAsync(r.scratch, v, sizeof v, hipMemcpyHostToDevice, r.comm));
        MNCCL(m, ncclAllReduce(r.scratch, r.scratch, 2, ncclUint64, ncclMax, r.nccl, r.comm));
        MHIP(m, hipMemcpyAsync(v, r.scratch, sizeof v, hipMemcpyDeviceToHost, r.comm));
        rc = wait_all(m);
        if (rc != NBODY_OK)
            return rc;
        hi = v[0];
        lo = ~v[1];
    }
    out2[0] = lo;
    out2[1] = hi;
    return NBODY_OK;
}

extern "C" int nbody_multi_info(const nbody_multi *m, int64_t *out8)
{
    if (!m || !out8)
        return NBODY_ERR_INVALID;
    int nccl_ranks = 0;
    for (const Rank &r : m->ranks)
        if (r.nccl) {
            int c = 0;
            if (ncclCommCount(r.nccl, &c) == ncclSuccess)
                nccl_ranks = c;
        }
    out8[0] = m->n_bodies;
    out8[1] = m->n_padded;
    out8[2] = m->chunk;
    out8[3] = m->split_len;
    out8[4] = m->world;
    out8[5] = (int64_t)m->ranks.size();
    out8[6] = nccl_ranks;  // ranks of the RCCL communicator (0: peer copies / a single rank without RCCL)
    out8[7] = m->cfg.exchange;
    return NBODY_OK;
}

extern "C" nbody_ctx *nbody_multi_shard(nbody_multi *m, int local_index)
{
    return (m && local_index >= 0 && local_index < (int)m->ranks.size()) ? m->ranks[(size_t)local_index].ctx : nullptr;
}

extern "C" float *nbody_multi_positions_device(nbody_multi *m, int local_index)
{
    return (m && local_index >= 0 && local_index < (int)m->ranks.size()) ? m->ranks[(size_t)local_index].pos : nullptr;
}

extern "C" float *nbody_multi_velocities_device(nbody_multi *m, int local_index)
{
    return (m && local_index >= 0 && local_index < (int)m->ranks.size()) ? m->ranks[(size_t)local_index].vel : nullptr;
}
