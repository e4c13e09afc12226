// fake_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for the fifteen RCCL entry points
// n_body_problem_amd/csrc/nbody_multi.hip calls, so that the ONE-RANK-PER-PROCESS path of the library
// (nbody_multi_create_rank: what the benchmark's ranks run on a multi-GPU node) can be executed by several processes that
// share ONE GPU.  Real RCCL refuses two ranks on one device, and the GPU boxes this repository is tested on have one.
//
// It is linked INTO a second build of the library (tests/fake_rccl/libnbody_amd_fake_rccl.so = the product's object files +
// this file, -Bsymbolic, no -lrccl: tests/fake_rccl/build_fake_rccl.py); the product library itself always links the real librccl.
// Only tests load it (NBODY_AMD_LIBRARY).  Both ways the library makes communicators are covered: ncclCommInitRank (one rank per
// process, the segment) and ncclCommInitAll (all ranks in one process, one thread, every collective inside one group: plain
// memory instead of the segment, no barriers).  What it keeps of RCCL's contract: a communicator of `world` ranks made from an id
// one rank creates and distributes; collectives and grouped send/recv pairs that every rank must call in the same order;
// results in the receive buffers once the stream has passed the call.  What it drops: asynchrony (every call, or the
// ncclGroupEnd that closes a group, waits for the stream, moves the bytes through a POSIX shared-memory segment with two
// barriers, and returns when the data is in place) and speed.  A rank that never arrives -- or one that fails locally before a
// barrier and returns early -- makes the others fail with ncclSystemError after FAKE_RCCL_TIMEOUT_S (default 120) seconds
// instead of hanging.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Shared {  // the head of the segment; `world` slots of slot_bytes follow, one per rank
    std::atomic<uint32_t> ready;       // the creating rank has initialised the header
    std::atomic<uint32_t> arrived;     // barrier: ranks that have arrived in the current generation
    std::atomic<uint32_t> generation;  // barrier: bumped by the last arrival
    uint32_t world;
    uint64_t slot_bytes;
    uint64_t header_bytes;
};

struct Message {  // in a rank's slot: messages back to back, each 16-byte aligned
    uint32_t kind;  // 1 all-gather, 2 send, 3 all-reduce
    int32_t dst;    // send: the receiving rank
    uint64_t bytes;
};

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

constexpr size_t kAlign = 16;
size_t aligned(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

double timeout_seconds()
{
    const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
    return e ? atof(e) : 120.0;
}

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

}  // namespace

struct ncclComm {
    int rank = 0, world = 0;
    Shared *sh = nullptr;
    size_t map_bytes = 0;
    bool failed = false;
    bool local = false;  // ncclCommInitAll: all ranks in this process, the "segment" is plain memory shared by its communicators
    std::string last_error;
    char *slot(int r) const { return reinterpret_cast<char *>(sh) + sh->header_bytes + (size_t)r * sh->slot_bytes; }
};

namespace {

bool barrier(ncclComm *c)
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
    unsigned spins = 0;
    while (sh->generation.load(std::memory_order_acquire) == gen) {
        if (++spins < 2000)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            c->failed = true;
            c->last_error = "fake rccl: a rank did not arrive within " + std::to_string(limit) + " s";
            return false;
        }
    }
    return true;
}

ncclResult_t fail(ncclComm *c, ncclResult_t code, const std::string &what)
{
    c->failed = true;
    c->last_error = "fake rccl: " + what;
    return code;
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

// the k-th message of `kind` (for sends: addressed to `dst`) in rank r's slot; nullptr if there is none
const Message *find_message(const ncclComm *c, int r, uint32_t kind, int dst, int k)
{
    const char *p = c->slot(r);
    uint64_t n_msgs = 0;
    std::memcpy(&n_msgs, p, sizeof n_msgs);
    p += kAlign;
    for (uint64_t i = 0; i < n_msgs; ++i) {
        const Message *m = reinterpret_cast<const Message *>(p);
        if (m->kind == kind && (kind != kSend || m->dst == dst) && k-- == 0)
            return m;
        p += kAlign + aligned(m->bytes);
    }
    return nullptr;
}

// what rank c sends in this group, into its slot
ncclResult_t write_phase(ncclComm *c, const std::vector<Op> &ops)
{
    for (const Op &o : ops)  // everything the stream was given before the call has happened
        if (o.comm == c && hipStreamSynchronize(o.stream) != hipSuccess)
            return fail(c, ncclUnhandledCudaError, "hipStreamSynchronize");
    char *p = c->slot(c->rank);
    const char *end = p + c->sh->slot_bytes;
    uint64_t n_msgs = 0;
    char *q = p + kAlign;
    for (const Op &o : ops) {
        if (o.comm != c || o.kind == kRecv)
            continue;
        if (q + kAlign + aligned(o.bytes) > end)
            return fail(c, ncclInternalError, "the messages of one group exceed FAKE_RCCL_SLOT_MB");
        Message m{(uint32_t)o.kind, o.peer, o.bytes};
        std::memcpy(q, &m, sizeof m);
        if (o.bytes && hipMemcpy(q + kAlign, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(c, ncclUnhandledCudaError, "hipMemcpy (device to host)");
        q += kAlign + aligned(o.bytes);
        ++n_msgs;
    }
    std::memcpy(p, &n_msgs, sizeof n_msgs);
    return ncclSuccess;
}

// what rank c receives in this group, out of the slots of all ranks
ncclResult_t read_phase(ncclComm *c, const std::vector<Op> &ops)
{
    ncclResult_t rc = ncclSuccess;
    int n_gather = 0, n_reduce = 0;
    std::vector<int> n_recv((size_t)c->world, 0);
    for (const Op &o : ops) {
        if (o.comm != c)
            continue;
        if (o.kind == kAllGather) {
            for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
                const Message *m = find_message(c, r, kAllGather, 0, n_gather);
                if (!m || m->bytes != o.bytes)
                    rc = fail(c, ncclInvalidUsage, "ncclAllGather: rank " + std::to_string(r) + " did not enter the same call");
                else if (o.bytes && hipMemcpy(static_cast<char *>(o.recv) + (size_t)r * o.bytes,
                                              reinterpret_cast<const char *>(m) + kAlign, o.bytes, hipMemcpyHostToDevice) != hipSuccess)
                    rc = fail(c, ncclUnhandledCudaError, "hipMemcpy (host to device)");
            }
            ++n_gather;
        } else if (o.kind == kRecv) {
            const Message *m = find_message(c, o.peer, kSend, c->rank, n_recv[(size_t)o.peer]++);
            if (!m || m->bytes != o.bytes)
                rc = fail(c, ncclInvalidUsage, "ncclRecv: rank " + std::to_string(o.peer) + " posted no matching ncclSend");
            else if (o.bytes && hipMemcpy(o.recv, reinterpret_cast<const char *>(m) + kAlign, o.bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(c, ncclUnhandledCudaError, "hipMemcpy (host to device)");
        } else if (o.kind == kAllReduce) {
            std::vector<char> acc(o.bytes);
            for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {  // in rank order: every rank gets the same bits
                const Message *m = find_message(c, r, kAllReduce, 0, n_reduce);
                if (!m || m->bytes != o.bytes) {
                    rc = fail(c, ncclInvalidUsage, "ncclAllReduce: rank " + std::to_string(r) + " did not enter the same call");
                    break;
                }
                const char *src = reinterpret_cast<const char *>(m) + kAlign;
                if (r == 0) {
                    std::memcpy(acc.data(), src, o.bytes);
                } else if (o.type == ncclDouble && o.red == ncclSum) {
                    for (size_t i = 0; i < o.count; ++i)
                        reinterpret_cast<double *>(acc.data())[i] += reinterpret_cast<const double *>(src)[i];
                } else if (o.type == ncclUint64 && o.red == ncclMax) {
                    for (size_t i = 0; i < o.count; ++i) {
                        uint64_t &a = reinterpret_cast<uint64_t *>(acc.data())[i];
                        const uint64_t b = reinterpret_cast<const uint64_t *>(src)[i];
                        a = a > b ? a : b;
                    }
                } else {
                    rc = fail(c, ncclInvalidArgument, "ncclAllReduce: only double sums and uint64 maxima (what the library uses)");
                }
            }
            if (rc == ncclSuccess && o.bytes && hipMemcpy(o.recv, acc.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(c, ncclUnhandledCudaError, "hipMemcpy (host to device)");
            ++n_reduce;
        }
        if (rc != ncclSuccess)
            break;
    }
    return rc;
}

ncclResult_t run(std::vector<Op> &ops)
{
    if (ops.empty())
        return ncclSuccess;
    ncclComm *c = ops[0].comm;
    if (c->local) {
        // every rank of the communicator lives in this process (ncclCommInitAll) and one thread drives them all: a group must
        // hold the calls of ALL ranks (as a real collective would need), their messages are written first, then delivered
        std::vector<ncclComm *> comms;
        for (const Op &o : ops) {
            if (!o.comm->local || o.comm->sh != c->sh)
                return fail(c, ncclInvalidUsage, "a group mixes communicators of different worlds");
            bool seen = false;
            for (ncclComm *k : comms)
                seen |= k == o.comm;
            if (!seen)
                comms.push_back(o.comm);
        }
        if ((int)comms.size() != c->world)
            return fail(c, ncclInvalidUsage, "a group holds the calls of " + std::to_string(comms.size()) + " of " +
                                                 std::to_string(c->world) + " ranks: a real collective would wait for ever");
        for (ncclComm *k : comms)
            if (k->failed)
                return ncclSystemError;
        ncclResult_t rc = ncclSuccess;
        for (size_t i = 0; i < comms.size() && rc == ncclSuccess; ++i)
            rc = write_phase(comms[i], ops);
        for (size_t i = 0; i < comms.size() && rc == ncclSuccess; ++i)
            rc = read_phase(comms[i], ops);
        return rc;
    }
    for (const Op &o : ops)
        if (o.comm != c)  // several communicators of the one-rank-per-process kind in one group
            return fail(c, ncclInvalidUsage, "one communicator per group in the one-rank-per-process model");
    if (c->failed)
        return ncclSystemError;
    ncclResult_t rc = write_phase(c, ops);
    if (rc != ncclSuccess)
        return rc;
    if (!barrier(c))
        return ncclSystemError;
    rc = read_phase(c, ops);
    if (!barrier(c))  // nobody rewrites a slot before everybody has read it (entered even after a local error)
        return ncclSystemError;
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

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || rank < 0 || rank >= world)
        return ncclInvalidArgument;
    uint64_t token[2];
    std::memcpy(token, id.internal, sizeof token);
    char name[96];
    std::snprintf(name, sizeof name, "/fake_rccl_%016llx%016llx", (unsigned long long)token[0], (unsigned long long)token[1]);
    const char *mb = getenv("FAKE_RCCL_SLOT_MB");
    const uint64_t slot_bytes = (uint64_t)(mb ? atol(mb) : 64) << 20;
    const uint64_t header = aligned(sizeof(Shared)) + 4096;
    const size_t map_bytes = header + (size_t)world * slot_bytes;
    bool creator = true;
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) {
        creator = false;
        const auto t0 = std::chrono::steady_clock::now();
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {  // the creating rank is on its way
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds())
                return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    if (creator && ftruncate(fd, (off_t)map_bytes) != 0) {
        close(fd);
        shm_unlink(name);
        return ncclSystemError;
    }
    if (!creator) {  // the creator sizes the segment before anybody maps it
        struct stat st;
        const auto t0 = std::chrono::steady_clock::now();
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < map_bytes) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds()) {
                close(fd);
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void *mem = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) {
        if (creator)
            shm_unlink(name);
        return ncclSystemError;
    }
    ncclComm *c = new ncclComm;
    c->rank = rank;
    c->world = world;
    c->sh = static_cast<Shared *>(mem);
    c->map_bytes = map_bytes;
    if (creator) {  // a fresh segment is zero-filled: the atomics start at 0
        c->sh->world = (uint32_t)world;
        c->sh->slot_bytes = slot_bytes;
        c->sh->header_bytes = header;
        c->sh->ready.store(1, std::memory_order_release);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (c->sh->ready.load(std::memory_order_acquire) == 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds()) {
                munmap(mem, map_bytes);
                delete c;
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    const bool all_here = barrier(c);  // every rank has mapped the segment: its name can go (no litter in /dev/shm)
    if (creator)
        shm_unlink(name);
    if (!all_here) {
        munmap(mem, map_bytes);
        delete c;
        return ncclSystemError;
    }
    *out = c;
    return ncclSuccess;
}

// Every rank in this process (the library's nbody_multi_create with NBODY_TRANSPORT_RCCL): the communicators share a block of
// plain memory laid out like the segment; the block is freed with the last of them.
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *)
{
    if (!comms || n < 1)
        return ncclInvalidArgument;
    const char *mb = getenv("FAKE_RCCL_SLOT_MB");
    const uint64_t slot_bytes = (uint64_t)(mb ? atol(mb) : 64) << 20;
    const uint64_t header = aligned(sizeof(Shared)) + 4096;
    const size_t bytes = header + (size_t)n * slot_bytes;
    void *mem = std::calloc(1, bytes);
    if (!mem)
        return ncclSystemError;
    Shared *sh = new (mem) Shared;
    sh->world = (uint32_t)n;
    sh->slot_bytes = slot_bytes;
    sh->header_bytes = header;
    sh->arrived.store((uint32_t)n);  // here: the number of communicators still alive
    for (int i = 0; i < n; ++i) {
        ncclComm *c = new ncclComm;
        c->rank = i;
        c->world = n;
        c->sh = sh;
        c->map_bytes = bytes;
        c->local = true;
        comms[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (c) {
        if (!c->local)
            munmap(c->sh, c->map_bytes);
        else if (c->sh->arrived.fetch_sub(1) == 1)
            std::free(c->sh);
        delete c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }

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
    *async = c->failed ? ncclSystemError : ncclSuccess;
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
    default: return "error";
    }
}

const char *ncclGetLastError(ncclComm_t c) { return c ? c->last_error.c_str() : ""; }

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
