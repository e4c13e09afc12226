// fake_rccl.hip -- TEST INFRASTRUCTURE, not part of the product: a stand-in for the RCCL entry points
// n_body_problem_amd/csrc/nbody_multi.hip calls, so that the library's RCCL branch -- one rank per process
// (nbody_multi_create_rank: what the benchmark's ranks run on a multi-GPU node) and every rank in one process
// (nbody_multi_create) -- can be executed by several ranks that share ONE GPU.  Real RCCL refuses two ranks on one device,
// and the GPU boxes this repository is tested on have one.
//
// It is linked INTO a second build of the library (tests/fake_rccl/libnbody_amd_fake_rccl.so = the product's object files +
// this file, -Bsymbolic, no -lrccl: tests/fake_rccl/build_fake_rccl.py); the product library itself always links the real
// librccl.  Only tests load it (NBODY_AMD_LIBRARY).
//
// Round 4: it keeps RCCL's STREAM SEMANTICS (round 3's version waited for the stream inside every call and moved the bytes
// with the data already in place when the call returned, so no event edge of the library's exchange could fail a test):
//   * every call returns at once; a group takes effect at the ncclGroupEnd that closes it; nothing waits for the stream;
//   * the bytes move ON THE CALLER'S STREAM: a send buffer is read when the stream gets there (behind whatever the caller
//     ordered in front of the call), a receive buffer is written when every peer's stream has got there too, and work the
//     caller enqueues behind the call on that stream sees the result -- and nothing else does: a consumer on another stream
//     that forgot its hipStreamWaitEvent reads stale rows, a producer that forgot to order the communication stream behind
//     its update sends them;
//   * non-blocking communicators (ncclCommInitRankConfig, blocking = 0: what the library creates): creation runs in a
//     thread, ncclCommGetAsyncError answers ncclInProgress until every rank has arrived, a group of such communicators
//     answers ncclInProgress once before ncclSuccess, ncclCommAbort ends an unfinished creation.
// Two transports, chosen when the communicator is made (every rank publishes its process id):
//   * ranks in DIFFERENT processes: a POSIX shared-memory segment, pinned in every process (hipHostRegister), holds one slot
//     per rank and a control block of sequence numbers.  Per group and rank, on the caller's stream: [wait until the peers
//     have read the previous group out of my slot] . copies device -> my slot . signal kernel (data sequence number, what I
//     sent to whom) . wait kernel (one lane per peer spins on the peer's data sequence number, checks that the peer sent what
//     this rank expects, gives up after FAKE_RCCL_TIMEOUT_S or when the communicator is aborted) . copies peers' slots ->
//     device . signal kernel (read sequence number).  What real RCCL's kernels do with flags in peer memory, through host
//     memory (tools/probe_shm_flags.hip is the probe of the mechanism on this pool);
//   * every rank in ONE process (one thread driving all ranks, or one thread per rank): spinning kernels of ranks that share
//     the process's few hardware queues could wait for each other for ever, so this transport orders the streams with
//     events: publish the group and record `ready` . host rendezvous of the ranks' threads (none with one thread) . wait for
//     the peers' `ready`, copy device -> device out of their send buffers, record `done` . rendezvous . wait for the peers'
//     `done` (a send buffer is the sender's again only when every receiver has read it).  The host meets, the streams do
//     not wait for the host.
// A rank that never arrives makes the others fail (ncclSystemError / an asynchronous error) after FAKE_RCCL_TIMEOUT_S
// (default 60) seconds instead of hanging.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int kMaxWorld = 16;
constexpr size_t kAlign = 256;
constexpr size_t kHeaderBytes = 16384;
// the status block of a communicator (pinned, device-visible): words the wait kernels set / read with plain atomic stores and
// loads (no read-modify-write on host memory from the device)
enum { kStatusTimeout = 0, kStatusAbort = 1, kStatusMismatch = 2 };

size_t aligned(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

double timeout_seconds()
{
    const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
    return e && atof(e) > 0 ? atof(e) : 60.0;
}

uint64_t slot_bytes_setting()
{
    const char *mb = getenv("FAKE_RCCL_SLOT_MB");
    return (uint64_t)(mb && atol(mb) > 0 ? atol(mb) : 64) << 20;
}

struct RankCtl {  // one per rank, in the pinned segment: written by that rank's signal kernels, read by the peers' wait kernels
    uint32_t data_seq;  // the group whose outgoing messages are complete in this rank's slot
    uint32_t read_seq;  // the group whose incoming messages this rank has copied out of the others' slots
    uint32_t pid;       // host side, at creation
    uint32_t pad;
    uint64_t sig[kMaxWorld];  // sig[q]: signature of what this rank sent to rank q in group data_seq (and of its collectives)
    char fill[256 - 16 - 8 * kMaxWorld];
};
static_assert(sizeof(RankCtl) == 256, "RankCtl is one 256-byte block");

struct Shared {  // the head of the segment; `world` slots of slot_bytes follow at header_bytes
    std::atomic<uint32_t> ready;       // the creating rank has initialised the header
    std::atomic<uint32_t> arrived;     // creation barrier: ranks that have arrived in the current generation
    std::atomic<uint32_t> generation;  // creation barrier: bumped by the last arrival
    uint32_t world;
    uint64_t slot_bytes;
    uint64_t header_bytes;
    char fill[256 - 32];
    RankCtl ctl[kMaxWorld];
};
static_assert(sizeof(Shared) <= kHeaderBytes, "the header fits its block");

enum { kAllGather = 1, kSend = 2, kAllReduce = 3, kRecv = 4 };

struct Op {
    int kind;
    const void *send;
    void *recv;
    size_t bytes;  // per rank (all-gather), of the message (send / recv), of the vector (all-reduce)
    int peer;
    ncclDataType_t type;
    ncclRedOp_t red;
    size_t count;
    ncclComm_t comm;
    hipStream_t stream;
};

// every rank of a communicator in ONE process: what the ranks' threads share
struct LocalWorld {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<std::vector<Op>> ops;   // [rank]: the group the rank has published
    std::vector<hipEvent_t> ready, done;
    char *stage = nullptr;              // device memory, kStageBytes per rank: all-reduce contributions
    int members = 0;                    // communicators alive
};
constexpr size_t kStageBytes = 4096;

std::mutex g_registry_mu;
std::map<std::pair<uint64_t, uint64_t>, std::shared_ptr<LocalWorld>> g_registry;

struct Sig { uint64_t v[kMaxWorld]; };
struct Srcs { const void *p[kMaxWorld]; };

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local std::vector<ncclComm_t> g_group_inits;  // communicators whose creation was asked for inside the open group

uint64_t mix(uint64_t h, uint64_t v)
{
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h * 0xff51afd7ed558ccdull + 1;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclFloat: return 4;
    case ncclDouble: return 8;
    case ncclUint64: return 8;
    case ncclInt64: return 8;
    case ncclInt32: return 4;
    case ncclUint32: return 4;
    case ncclUint8: return 1;
    case ncclInt8: return 1;
    default: return 0;
    }
}

}  // namespace

struct ncclComm {
    int rank = 0, world = 0, device = 0;
    bool blocking = true;
    std::atomic<int> state{(int)ncclSuccess};  // ncclInProgress while the creation thread runs; an error once failed
    std::atomic<bool> abort_init{false};
    std::atomic<int> pending_polls{0};         // non-blocking: a group answers ncclInProgress once before ncclSuccess
    std::thread init_thread;
    std::pair<uint64_t, uint64_t> token{0, 0};
    // ranks in different processes
    Shared *sh = nullptr;        // host address of the segment
    char *sh_dev = nullptr;      // its device address
    size_t map_bytes = 0;
    bool registered = false;
    uint32_t seq = 0;
    uint32_t *status_host = nullptr, *status_dev = nullptr;  // kStatus* words: set by the wait kernels / by ncclCommAbort
    long long timeout_ticks = 0;
    // every rank in this process
    std::shared_ptr<LocalWorld> lw;
    hipStream_t last_stream = nullptr;
    bool used_stream = false;
    std::mutex err_mu;
    std::string last_error;

    char *slot_host(int r) const { return reinterpret_cast<char *>(sh) + sh->header_bytes + (size_t)r * sh->slot_bytes; }
    char *slot_dev(int r) const { return sh_dev + sh->header_bytes + (size_t)r * sh->slot_bytes; }
    size_t sub_bytes() const { return sh->slot_bytes / (size_t)(world + 1) / kAlign * kAlign; }
    RankCtl *ctl_dev(int r) const
    {
        return reinterpret_cast<RankCtl *>(sh_dev + (reinterpret_cast<char *>(&sh->ctl[0]) - reinterpret_cast<char *>(sh))) + r;
    }
};

namespace {

ncclResult_t fail(ncclComm *c, ncclResult_t code, const std::string &what)
{
    {
        std::lock_guard<std::mutex> lock(c->err_mu);
        c->last_error = "fake rccl: " + what;
    }
    c->state.store((int)code);
    if (c->lw) {
        std::lock_guard<std::mutex> lock(c->lw->mu);
        c->lw->broken = true;
        c->lw->cv.notify_all();
    }
    return code;
}

// ---- device side ---------------------------------------------------------------------------------------------------

__global__ void fake_signal_kernel(RankCtl *mine, int field, uint32_t seq, Sig sig, int world)
{
    if (field == 0 && (int)threadIdx.x < world)
        mine->sig[threadIdx.x] = sig.v[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(field == 0 ? &mine->data_seq : &mine->read_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One lane per peer: spins until the peer's sequence number has reached `want`, bounded by the wall clock and by the abort
// flag, so the kernel always ends.  field 0 also checks that the peer sent this rank what this rank expects to receive.
__global__ void fake_wait_kernel(RankCtl *ctl, int field, int world, int self, uint32_t want, Sig expect, uint32_t *status,
                                 long long timeout_ticks)
{
    const int peer = (int)threadIdx.x;
    if (peer >= world || peer == self)
        return;
    uint32_t *word = field == 0 ? &ctl[peer].data_seq : &ctl[peer].read_seq;
    const long long t0 = wall_clock64();
    bool arrived = true;
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (__hip_atomic_load(status + kStatusAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
            arrived = false;
            break;
        }
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(status + kStatusTimeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            arrived = false;
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
    if (arrived && field == 0 && __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == want) {
        const uint64_t got = __hip_atomic_load(&ctl[peer].sig[self], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (got != expect.v[peer])
            __hip_atomic_store(status + kStatusMismatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
}

// out[i] = the ranks' contributions combined in rank order (every rank gets the same bits): double sums, uint64 maxima
__global__ void fake_reduce_kernel(void *out, Srcs srcs, int world, size_t count, int is_double_sum)
{
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
        if (is_double_sum) {
            double a = static_cast<const double *>(srcs.p[0])[i];
            for (int r = 1; r < world; ++r)
                a += static_cast<const double *>(srcs.p[r])[i];
            static_cast<double *>(out)[i] = a;
        } else {
            uint64_t a = static_cast<const uint64_t *>(srcs.p[0])[i];
            for (int r = 1; r < world; ++r) {
                const uint64_t b = static_cast<const uint64_t *>(srcs.p[r])[i];
                a = a > b ? a : b;
            }
            static_cast<uint64_t *>(out)[i] = a;
        }
    }
}

bool reduce_supported(const Op &o)
{
    return (o.type == ncclDouble && o.red == ncclSum) || (o.type == ncclUint64 && o.red == ncclMax);
}

#define FHIP(c, call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return fail((c), ncclUnhandledCudaError, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

// ---- one group of one rank whose peers live in other processes ----------------------------------------------------------

uint64_t collective_signature(const std::vector<Op> &ops, ncclComm *c)
{
    uint64_t h = 0x1234;
    for (const Op &o : ops)
        if (o.comm == c && (o.kind == kAllGather || o.kind == kAllReduce))
            h = mix(mix(mix(h, (uint64_t)o.kind), o.bytes), (uint64_t)o.type * 64 + (uint64_t)o.red);
    return h;
}

ncclResult_t enqueue_process_group(ncclComm *c, const std::vector<Op> &ops)
{
    hipStream_t stream = nullptr;
    bool have = false;
    for (const Op &o : ops) {
        if (o.comm != c)
            continue;
        if (have && o.stream != stream)
            return fail(c, ncclInvalidUsage, "the calls of one rank in one group use several streams (the test double serves one)");
        stream = o.stream;
        have = true;
    }
    if (!have)
        return ncclSuccess;
    FHIP(c, hipSetDevice(c->device));
    const int W = c->world, me = c->rank;
    const size_t sub = c->sub_bytes();
    const uint32_t k = ++c->seq;
    c->last_stream = stream;
    c->used_stream = true;
    Sig none{};
    if (W > 1 && k > 1)  // my slot is free again once every peer has read group k - 1 out of it
        hipLaunchKernelGGL(fake_wait_kernel, dim3(1), dim3(64), 0, stream, c->ctl_dev(0), 1, W, me, k - 1, none, c->status_dev,
                           c->timeout_ticks);
    // outgoing: sends to q back to back in sub-slot q, collectives in the last sub-slot
    std::vector<size_t> at((size_t)W + 1, 0);
    Sig sig{}, expect{};
    const uint64_t coll = collective_signature(ops, c);
    for (int q = 0; q < W; ++q)
        sig.v[q] = expect.v[q] = coll;
    for (const Op &o : ops) {
        if (o.comm != c || o.kind == kRecv)
            continue;
        const int area = o.kind == kSend ? o.peer : W;
        if (o.kind == kAllReduce && !reduce_supported(o))
            return fail(c, ncclInvalidArgument, "ncclAllReduce: only double sums and uint64 maxima (what the library uses)");
        if (at[(size_t)area] + aligned(o.bytes) > sub)
            return fail(c, ncclInternalError, "the messages of one group exceed the slot (FAKE_RCCL_SLOT_MB)");
        if (o.bytes)
            FHIP(c, hipMemcpyAsync(c->slot_host(me) + (size_t)area * sub + at[(size_t)area], o.send, o.bytes, hipMemcpyDeviceToHost,
                                   stream));
        at[(size_t)area] += aligned(o.bytes);
        if (o.kind == kSend)
            sig.v[o.peer] = mix(sig.v[o.peer], o.bytes);
    }
    hipLaunchKernelGGL(fake_signal_kernel, dim3(1), dim3(64), 0, stream, c->ctl_dev(me), 0, k, sig, W);
    for (const Op &o : ops)
        if (o.comm == c && o.kind == kRecv)
            expect.v[o.peer] = mix(expect.v[o.peer], o.bytes);
    if (W > 1)
        hipLaunchKernelGGL(fake_wait_kernel, dim3(1), dim3(64), 0, stream, c->ctl_dev(0), 0, W, me, k, expect, c->status_dev,
                           c->timeout_ticks);
    // incoming
    std::vector<size_t> from((size_t)W, 0);  // recv: where the next message of peer p starts in p's sub-slot for me
    size_t coll_at = 0;
    for (const Op &o : ops) {
        if (o.comm != c)
            continue;
        if (o.kind == kRecv) {
            if (o.bytes)
                FHIP(c, hipMemcpyAsync(o.recv, c->slot_host(o.peer) + (size_t)me * sub + from[(size_t)o.peer], o.bytes,
                                       hipMemcpyHostToDevice, stream));
            from[(size_t)o.peer] += aligned(o.bytes);
        } else if (o.kind == kAllGather) {
            for (int r = 0; r < W && o.bytes; ++r) {
                char *dst = static_cast<char *>(o.recv) + (size_t)r * o.bytes;
                if (r != me)
                    FHIP(c, hipMemcpyAsync(dst, c->slot_host(r) + (size_t)W * sub + coll_at, o.bytes, hipMemcpyHostToDevice, stream));
                else if (dst != o.send)
                    FHIP(c, hipMemcpyAsync(dst, o.send, o.bytes, hipMemcpyDeviceToDevice, stream));
            }
            coll_at += aligned(o.bytes);
        } else if (o.kind == kAllReduce) {
            Srcs srcs{};
            for (int r = 0; r < W; ++r)
                srcs.p[r] = c->slot_dev(r) + (size_t)W * sub + coll_at;
            hipLaunchKernelGGL(fake_reduce_kernel, dim3(1), dim3(64), 0, stream, o.recv, srcs, W, o.count, o.type == ncclDouble ? 1 : 0);
            coll_at += aligned(o.bytes);
        }
    }
    hipLaunchKernelGGL(fake_signal_kernel, dim3(1), dim3(64), 0, stream, c->ctl_dev(me), 1, k, none, W);
    FHIP(c, hipGetLastError());
    return ncclSuccess;
}

// ---- one group of the ranks of one process ---------------------------------------------------------------------------------

// Host rendezvous of the ranks' threads: returns when all `world` ranks have arrived (the ranks of `mine` arrive together).
bool local_rendezvous(LocalWorld &lw, int mine)
{
    std::unique_lock<std::mutex> lock(lw.mu);
    if (lw.broken)
        return false;
    const uint64_t gen = lw.generation;
    lw.arrived += mine;
    if (lw.arrived >= lw.world) {
        lw.arrived = 0;
        ++lw.generation;
        lw.cv.notify_all();
        return true;
    }
    const bool ok = lw.cv.wait_for(lock, std::chrono::duration<double>(timeout_seconds()),
                                   [&] { return lw.generation != gen || lw.broken; });
    if (!ok || lw.broken) {
        lw.broken = true;
        lw.cv.notify_all();
        return false;
    }
    return true;
}

// the k-th op of `kind` in a rank's published group (sends: addressed to `dst`)
const Op *find_op(const std::vector<Op> &ops, int kind, int dst, int k)
{
    for (const Op &o : ops)
        if (o.kind == kind && (kind != kSend || o.peer == dst) && k-- == 0)
            return &o;
    return nullptr;
}

ncclResult_t enqueue_local_group(const std::vector<ncclComm *> &cohort, const std::vector<Op> &ops)
{
    LocalWorld &lw = *cohort[0]->lw;
    const int W = lw.world;
    std::vector<hipStream_t> streams(cohort.size(), nullptr);
    // publish, stage the all-reduce contributions, record `ready`
    for (size_t i = 0; i < cohort.size(); ++i) {
        ncclComm *c = cohort[i];
        FHIP(c, hipSetDevice(c->device));
        std::vector<Op> mine;
        bool have = false;
        for (const Op &o : ops)
            if (o.comm == c) {
                if (have && o.stream != streams[i])
                    return fail(c, ncclInvalidUsage, "the calls of one rank in one group use several streams (the test double serves one)");
                streams[i] = o.stream;
                have = true;
                mine.push_back(o);
            }
        size_t at = 0;
        for (const Op &o : mine)
            if (o.kind == kAllReduce) {
                if (!reduce_supported(o))
                    return fail(c, ncclInvalidArgument, "ncclAllReduce: only double sums and uint64 maxima (what the library uses)");
                if (at + aligned(o.bytes) > kStageBytes)
                    return fail(c, ncclInternalError, "all-reduce contributions of one group exceed the staging block");
                FHIP(c, hipMemcpyAsync(lw.stage + (size_t)c->rank * kStageBytes + at, o.send, o.bytes, hipMemcpyDeviceToDevice, streams[i]));
                at += aligned(o.bytes);
            }
        {
            std::lock_guard<std::mutex> lock(lw.mu);
            lw.ops[(size_t)c->rank] = std::move(mine);
        }
        FHIP(c, hipEventRecord(lw.ready[(size_t)c->rank], streams[i]));
        c->last_stream = streams[i];
        c->used_stream = true;
    }
    if (!local_rendezvous(lw, (int)cohort.size()))
        return fail(cohort[0], ncclSystemError, "a rank of this process did not enter the group within the timeout");
    // behind every peer's `ready`: the copies into this rank's receive buffers, then `done`
    for (size_t i = 0; i < cohort.size(); ++i) {
        ncclComm *c = cohort[i];
        FHIP(c, hipSetDevice(c->device));
        const int me = c->rank;
        for (int p = 0; p < W; ++p)
            if (p != me)
                FHIP(c, hipStreamWaitEvent(streams[i], lw.ready[(size_t)p], 0));
        std::vector<Op> mine;
        std::vector<std::vector<Op>> theirs((size_t)W);
        {
            std::lock_guard<std::mutex> lock(lw.mu);
            mine = lw.ops[(size_t)me];
            theirs = lw.ops;
        }
        int n_gather = 0, n_reduce = 0;
        size_t reduce_at = 0;
        std::vector<int> n_recv((size_t)W, 0);
        for (const Op &o : mine) {
            if (o.kind == kAllGather) {
                for (int r = 0; r < W; ++r) {
                    const Op *s = find_op(theirs[(size_t)r], kAllGather, 0, n_gather);
                    if (!s || s->bytes != o.bytes)
                        return fail(c, ncclInvalidUsage, "ncclAllGather: rank " + std::to_string(r) + " did not enter the same call");
                    char *dst = static_cast<char *>(o.recv) + (size_t)r * o.bytes;
                    if (o.bytes && dst != s->send)
                        FHIP(c, hipMemcpyAsync(dst, s->send, o.bytes, hipMemcpyDeviceToDevice, streams[i]));
                }
                ++n_gather;
            } else if (o.kind == kRecv) {
                const Op *s = find_op(theirs[(size_t)o.peer], kSend, me, n_recv[(size_t)o.peer]++);
                if (!s || s->bytes != o.bytes)
                    return fail(c, ncclInvalidUsage, "ncclRecv: rank " + std::to_string(o.peer) + " posted no matching ncclSend");
                if (o.bytes)
                    FHIP(c, hipMemcpyAsync(o.recv, s->send, o.bytes, hipMemcpyDeviceToDevice, streams[i]));
            } else if (o.kind == kAllReduce) {
                Srcs srcs{};
                for (int r = 0; r < W; ++r) {
                    const Op *s = find_op(theirs[(size_t)r], kAllReduce, 0, n_reduce);
                    if (!s || s->bytes != o.bytes || s->type != o.type || s->red != o.red)
                        return fail(c, ncclInvalidUsage, "ncclAllReduce: rank " + std::to_string(r) + " did not enter the same call");
                    srcs.p[r] = lw.stage + (size_t)r * kStageBytes + reduce_at;
                }
                hipLaunchKernelGGL(fake_reduce_kernel, dim3(1), dim3(64), 0, streams[i], o.recv, srcs, W, o.count,
                                   o.type == ncclDouble ? 1 : 0);
                reduce_at += aligned(o.bytes);
                ++n_reduce;
            } else if (o.kind == kSend) {
                const std::vector<Op> &dst_ops = theirs[(size_t)o.peer];
                int sends_before = 0, recvs = 0;
                for (const Op &m : mine)
                    if (&m != &o && m.kind == kSend && m.peer == o.peer)
                        ++sends_before;
                    else if (&m == &o)
                        break;
                for (const Op &d : dst_ops)
                    recvs += d.kind == kRecv && d.peer == me;
                if (sends_before >= recvs)
                    return fail(c, ncclInvalidUsage, "ncclSend: rank " + std::to_string(o.peer) + " posted no matching ncclRecv");
            }
        }
        FHIP(c, hipEventRecord(lw.done[(size_t)me], streams[i]));
        FHIP(c, hipGetLastError());
    }
    if (!local_rendezvous(lw, (int)cohort.size()))
        return fail(cohort[0], ncclSystemError, "a rank of this process did not finish the group within the timeout");
    // a send buffer belongs to the sender's stream again only when every receiver has read it
    for (size_t i = 0; i < cohort.size(); ++i) {
        ncclComm *c = cohort[i];
        FHIP(c, hipSetDevice(c->device));
        for (int p = 0; p < W; ++p)
            if (p != c->rank)
                FHIP(c, hipStreamWaitEvent(streams[i], lw.done[(size_t)p], 0));
    }
    return ncclSuccess;
}

ncclResult_t run(std::vector<Op> &ops)
{
    if (ops.empty())
        return ncclSuccess;
    std::vector<ncclComm *> comms;
    for (const Op &o : ops) {
        bool seen = false;
        for (ncclComm *k : comms)
            seen |= k == o.comm;
        if (!seen)
            comms.push_back(o.comm);
    }
    bool any_nonblocking = false;
    for (ncclComm *c : comms) {
        const int st = c->state.load();
        if (st == (int)ncclInProgress)
            return fail(c, ncclInvalidUsage, "a call on a communicator whose creation has not finished");
        if (st != (int)ncclSuccess)
            return (ncclResult_t)st;
        any_nonblocking |= !c->blocking;
    }
    ncclResult_t rc = ncclSuccess;
    std::vector<bool> handled(comms.size(), false);
    for (size_t i = 0; i < comms.size() && rc == ncclSuccess; ++i) {
        if (handled[i])
            continue;
        ncclComm *c = comms[i];
        if (!c->lw) {
            handled[i] = true;
            rc = enqueue_process_group(c, ops);
            continue;
        }
        std::vector<ncclComm *> cohort;  // the ranks of one world this thread drives in this group
        for (size_t j = i; j < comms.size(); ++j)
            if (!handled[j] && comms[j]->lw == c->lw) {
                cohort.push_back(comms[j]);
                handled[j] = true;
            }
        rc = enqueue_local_group(cohort, ops);
    }
    if (rc == ncclSuccess && any_nonblocking) {
        for (ncclComm *c : comms)
            if (!c->blocking)
                c->pending_polls.store(1);
        return ncclInProgress;  // what a non-blocking communicator may answer: the caller polls ncclCommGetAsyncError
    }
    return rc;
}

ncclResult_t submit(const Op &o)
{
    if (!o.comm)
        return ncclInvalidArgument;
    g_ops.push_back(o);
    if (g_depth > 0)
        return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

// ---- creation ----------------------------------------------------------------------------------------------------------

bool creation_barrier(ncclComm *c)
{
    Shared *sh = c->sh;
    const uint32_t gen = sh->generation.load(std::memory_order_acquire);
    if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == sh->world) {
        sh->arrived.store(0, std::memory_order_relaxed);
        sh->generation.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = timeout_seconds();
    while (sh->generation.load(std::memory_order_acquire) == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (c->abort_init.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            return false;
    }
    return true;
}

void create_body(ncclComm *c)
{
    auto bail = [&](const std::string &what) {
        if (c->sh) {
            munmap(c->sh, c->map_bytes);
            c->sh = nullptr;
        }
        fail(c, ncclSystemError, what);
    };
    char name[96];
    std::snprintf(name, sizeof name, "/fake_rccl_%016llx%016llx", (unsigned long long)c->token.first, (unsigned long long)c->token.second);
    const uint64_t slot_bytes = slot_bytes_setting();
    const size_t map_bytes = kHeaderBytes + (size_t)c->world * slot_bytes;
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] {
        return c->abort_init.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds();
    };
    bool creator = true;
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) {
        creator = false;
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {  // the creating rank is on its way
            if (late())
                return bail("the creating rank's segment never appeared");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    if (creator && ftruncate(fd, (off_t)map_bytes) != 0) {
        close(fd);
        shm_unlink(name);
        return bail("ftruncate of the segment failed");
    }
    if (!creator) {  // the creator sizes the segment before anybody maps it
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < map_bytes) {
            if (late()) {
                close(fd);
                return bail("the segment was never sized");
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void *mem = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) {
        if (creator)
            shm_unlink(name);
        return bail("mmap of the segment failed");
    }
    c->sh = static_cast<Shared *>(mem);
    c->map_bytes = map_bytes;
    if (creator) {  // a fresh segment is zero-filled: the atomics and the sequence numbers start at 0
        c->sh->world = (uint32_t)c->world;
        c->sh->slot_bytes = slot_bytes;
        c->sh->header_bytes = kHeaderBytes;
        c->sh->ready.store(1, std::memory_order_release);
    } else {
        while (c->sh->ready.load(std::memory_order_acquire) == 0) {
            if (late())
                return bail("the creating rank never initialised the segment");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    c->sh->ctl[c->rank].pid = (uint32_t)getpid();
    const bool all_here = creation_barrier(c);  // every rank has mapped the segment: its name can go (no litter in /dev/shm)
    if (creator)
        shm_unlink(name);
    if (!all_here)
        return bail("a rank did not arrive within " + std::to_string(timeout_seconds()) + " s (or the creation was aborted)");
    int same = 0;
    for (int r = 0; r < c->world; ++r)
        same += c->sh->ctl[r].pid == c->sh->ctl[c->rank].pid;
    if (hipSetDevice(c->device) != hipSuccess)
        return bail("hipSetDevice failed");
    if (same == c->world && c->world > 1) {
        // every rank in this process: the event transport; the segment has done its job (the rendezvous of the creation)
        munmap(c->sh, c->map_bytes);
        c->sh = nullptr;
        std::shared_ptr<LocalWorld> lw;
        {
            std::lock_guard<std::mutex> lock(g_registry_mu);
            auto &slot = g_registry[c->token];
            if (!slot) {
                slot = std::make_shared<LocalWorld>();
                slot->world = c->world;
                slot->ops.resize((size_t)c->world);
                slot->ready.assign((size_t)c->world, nullptr);
                slot->done.assign((size_t)c->world, nullptr);
                if (hipMalloc((void **)&slot->stage, kStageBytes * (size_t)c->world) != hipSuccess)
                    slot->broken = true;
            }
            lw = slot;
            ++lw->members;
        }
        if (lw->broken || hipEventCreateWithFlags(&lw->ready[(size_t)c->rank], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lw->done[(size_t)c->rank], hipEventDisableTiming) != hipSuccess) {
            c->lw = lw;
            fail(c, ncclUnhandledCudaError, "creating the events of the in-process transport failed");
            return;
        }
        c->lw = lw;
        if (!local_rendezvous(*lw, 1)) {  // every rank's events exist before anybody waits for one
            fail(c, ncclSystemError, "a rank of this process did not finish its creation");
            return;
        }
    } else if (same != 1 && c->world > 1) {
        return bail("ranks partly in one process, partly in others: not served by the test double");
    } else {
        if (hipHostRegister(c->sh, c->map_bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess)
            return bail("hipHostRegister of the segment failed");
        c->registered = true;
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, c->sh, 0) != hipSuccess)
            return bail("hipHostGetDevicePointer of the segment failed");
        c->sh_dev = static_cast<char *>(dev);
        if (hipHostMalloc((void **)&c->status_host, 64, hipHostMallocMapped) != hipSuccess)
            return bail("hipHostMalloc of the status block failed");
        std::memset(c->status_host, 0, 64);
        void *sdev = nullptr;
        if (hipHostGetDevicePointer(&sdev, c->status_host, 0) != hipSuccess)
            return bail("hipHostGetDevicePointer of the status block failed");
        c->status_dev = static_cast<uint32_t *>(sdev);
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0)
            khz = 100000;
        c->timeout_ticks = (long long)(timeout_seconds() * 1e3 * (double)khz);
    }
    c->state.store((int)ncclSuccess);
}

ncclResult_t start_creation(ncclComm_t *out, int world, ncclUniqueId id, int rank, bool blocking)
{
    if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world)
        return ncclInvalidArgument;
    ncclComm *c = new (std::nothrow) ncclComm;
    if (!c)
        return ncclSystemError;
    uint64_t token[2];
    std::memcpy(token, id.internal, sizeof token);
    c->token = {token[0], token[1]};
    c->rank = rank;
    c->world = world;
    c->blocking = blocking;
    if (hipGetDevice(&c->device) != hipSuccess)
        c->device = 0;
    c->state.store((int)ncclInProgress);
    *out = c;
    try {
        c->init_thread = std::thread(create_body, c);
    } catch (...) {
        c->state.store((int)ncclSystemError);
        return ncclSystemError;
    }
    if (g_depth > 0) {  // inside a group: the group's end answers for it
        g_group_inits.push_back(c);
        return ncclSuccess;
    }
    if (!blocking)
        return ncclInProgress;
    c->init_thread.join();
    return (ncclResult_t)c->state.load();
}

void release(ncclComm *c, bool aborted)
{
    if (!c)
        return;
    c->abort_init.store(true);
    if (c->init_thread.joinable())
        c->init_thread.join();
    (void)hipSetDevice(c->device);
    if (c->status_host && aborted)
        __atomic_store_n(&c->status_host[kStatusAbort], 1u, __ATOMIC_RELEASE);  // the wait kernels in flight give up
    if (c->used_stream)
        (void)hipStreamSynchronize(c->last_stream);  // bounded: every wait kernel ends (timeout, abort flag)
    if (c->lw) {
        std::lock_guard<std::mutex> lock(g_registry_mu);
        {
            std::lock_guard<std::mutex> l2(c->lw->mu);
            if (aborted) {
                c->lw->broken = true;
                c->lw->cv.notify_all();
            }
        }
        if (--c->lw->members == 0) {
            (void)hipDeviceSynchronize();
            for (hipEvent_t e : c->lw->ready)
                if (e)
                    (void)hipEventDestroy(e);
            for (hipEvent_t e : c->lw->done)
                if (e)
                    (void)hipEventDestroy(e);
            if (c->lw->stage)
                (void)hipFree(c->lw->stage);
            g_registry.erase(c->token);
        }
    }
    if (c->sh) {
        if (c->registered)
            (void)hipHostUnregister(c->sh);
        munmap(c->sh, c->map_bytes);
    }
    if (c->status_host)
        (void)hipHostFree(c->status_host);
    delete c;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<uint64_t> counter{0};
    std::memset(id, 0, sizeof *id);
    const uint64_t token[2] = {(uint64_t)getpid() << 32 | (uint64_t)counter.fetch_add(1),
                               (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count()};
    std::memcpy(id->internal, token, sizeof token);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) { return start_creation(out, world, id, rank, true); }

ncclResult_t ncclCommInitRankConfig(ncclComm_t *out, int world, ncclUniqueId id, int rank, ncclConfig_t *config)
{
    return start_creation(out, world, id, rank, !(config && config->blocking == 0));
}

// Every rank in this process, blocking: ncclCommInitRank of all ranks inside one group.
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devices)
{
    if (!comms || n < 1)
        return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    int before = 0;
    (void)hipGetDevice(&before);
    ncclGroupStart();
    ncclResult_t rc = ncclSuccess;
    for (int i = 0; i < n && rc == ncclSuccess; ++i) {
        if (devices)
            (void)hipSetDevice(devices[i]);
        rc = start_creation(&comms[i], n, id, i, true);
    }
    const ncclResult_t end = ncclGroupEnd();
    (void)hipSetDevice(before);
    return rc != ncclSuccess ? rc : end;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    release(c, false);
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    release(c, true);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count)
{
    if (!c || !count)
        return ncclInvalidArgument;
    *count = c->world;
    return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *async)
{
    if (!c || !async)
        return ncclInvalidArgument;
    const int st = c->state.load();
    if (st != (int)ncclSuccess) {
        *async = (ncclResult_t)st;  // ncclInProgress while the creation runs, an error once anything failed
        return ncclSuccess;
    }
    if (c->status_host) {
        const bool mismatch = __atomic_load_n(&c->status_host[kStatusMismatch], __ATOMIC_ACQUIRE) != 0;
        if (mismatch || __atomic_load_n(&c->status_host[kStatusTimeout], __ATOMIC_ACQUIRE) != 0) {
            fail(c, mismatch ? ncclInvalidUsage : ncclSystemError,
                 mismatch ? "a peer did not send what this rank expected to receive (the ranks' calls do not match)"
                          : "a peer's data did not arrive within " + std::to_string(timeout_seconds()) + " s");
            *async = (ncclResult_t)c->state.load();
            return ncclSuccess;
        }
    }
    if (c->pending_polls.load() > 0) {
        c->pending_polls.fetch_sub(1);
        *async = ncclInProgress;
        return ncclSuccess;
    }
    *async = ncclSuccess;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error";
    case ncclSystemError: return "unhandled system error";
    case ncclInternalError: return "internal error";
    case ncclInvalidArgument: return "invalid argument";
    case ncclInvalidUsage: return "invalid usage";
    case ncclInProgress: return "in progress";
    default: return "error";
    }
}

const char *ncclGetLastError(ncclComm_t c)
{
    static thread_local std::string copy;
    if (!c)
        return "";
    std::lock_guard<std::mutex> lock(c->err_mu);
    copy = c->last_error;
    return copy.c_str();
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0)
        return ncclInvalidUsage;
    if (--g_depth > 0)
        return ncclSuccess;
    std::vector<ncclComm_t> inits;
    inits.swap(g_group_inits);
    if (!inits.empty()) {  // a group of creations: blocking ones are waited for here, non-blocking ones are polled by the caller
        bool pending = false;
        ncclResult_t rc = ncclSuccess;
        for (ncclComm_t c : inits) {
            if (c->blocking) {
                if (c->init_thread.joinable())
                    c->init_thread.join();
                if (c->state.load() != (int)ncclSuccess)
                    rc = (ncclResult_t)c->state.load();
            } else {
                pending = true;
            }
        }
        if (!g_ops.empty())
            return ncclInvalidUsage;  // creations and communication in one group: not served
        return rc != ncclSuccess ? rc : pending ? ncclInProgress : ncclSuccess;
    }
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    Op o{};
    o.kind = kAllGather;
    o.send = send;
    o.recv = recv;
    o.bytes = count * type_bytes(type);
    o.type = type;
    o.comm = comm;
    o.stream = stream;
    return submit(o);
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->world)
        return ncclInvalidArgument;
    Op o{};
    o.kind = kSend;
    o.send = send;
    o.bytes = count * type_bytes(type);
    o.peer = peer;
    o.comm = comm;
    o.stream = stream;
    return submit(o);
}

ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->world)
        return ncclInvalidArgument;
    Op o{};
    o.kind = kRecv;
    o.recv = recv;
    o.bytes = count * type_bytes(type);
    o.peer = peer;
    o.comm = comm;
    o.stream = stream;
    return submit(o);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    Op o{};
    o.kind = kAllReduce;
    o.send = send;
    o.recv = recv;
    o.count = count;
    o.bytes = count * type_bytes(type);
    o.type = type;
    o.red = op;
    o.comm = comm;
    o.stream = stream;
    return submit(o);
}

}  // extern "C"
