// rccl_nonblocking_probe.hip -- TEST INFRASTRUCTURE: the calling protocol csrc/nbody_multi.hip uses on its RCCL communicators,
// against the REAL librccl with the one rank a one-GPU box can have:
//   ncclCommInitRankConfig(blocking = 0) -> poll ncclCommGetAsyncError until it leaves ncclInProgress ->
//   ncclGroupStart . ncclAllGather (in place) . ncclSend / ncclRecv to the rank itself . ncclGroupEnd ->
//   poll again -> only THEN record an event on the stream -> the event implies the collective's result.
// Mode "absent": world size 2 with the peer missing, the creation in a helper thread with a 3 s limit (what the library does):
// reports whether this RCCL honours blocking = 0 for the creation (ROCm 7.2's RCCL 2.27.7 does not: the call sits in the
// bootstrap) -- either way the caller is back in time.
// Prints one line; exit code 0 = as expected.   usage: rccl_nonblocking_probe [one|absent]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <unistd.h>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static ncclResult_t settle(ncclComm_t comm, double limit_s, int *polls)
{
    const double t0 = now();
    ncclResult_t state = ncclInProgress;
    for (;;) {
        ncclResult_t q = ncclCommGetAsyncError(comm, &state);
        ++*polls;
        if (q != ncclSuccess)
            return q;
        if (state != ncclInProgress || now() - t0 > limit_s)
            return state;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

int main(int argc, char **argv)
{
    const bool absent = argc > 1 && std::strcmp(argv[1], "absent") == 0;
    if (hipSetDevice(0) != hipSuccess) {
        std::printf("no device\n");
        return 2;
    }
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) {
        std::printf("ncclGetUniqueId failed\n");
        return 2;
    }
    ncclConfig_t config = NCCL_CONFIG_INITIALIZER;
    config.blocking = 0;
    ncclComm_t comm = nullptr;
    int polls = 0;
    const double t0 = now();
    if (absent) {
        // The creation in a helper thread, as the library does it: whether RCCL honours blocking = 0 here (the call returns
        // ncclInProgress at once) or sits in its bootstrap until the peer arrives, the CALLER is back after its timeout.
        static std::atomic<int> returned{0};
        static ncclResult_t init_result = ncclInProgress;
        static ncclComm_t made = nullptr;
        std::thread([&id, &config] {
            (void)hipSetDevice(0);
            init_result = ncclCommInitRankConfig(&made, 2, id, 0, &config);
            returned.store(1);
        }).detach();
        const double limit = 3.0;
        while (!returned.load() && now() - t0 < limit)
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        if (!returned.load()) {
            std::printf("absent peer: ncclCommInitRankConfig(blocking = 0) has not returned after %.1f s: this RCCL's creation blocks in "
                        "its bootstrap whatever config.blocking says -- the helper thread is left behind and the caller reports\n", limit);
            std::fflush(stdout);
            _exit(0);
        }
        ncclResult_t state = ncclInProgress;
        if (init_result == ncclInProgress || init_result == ncclSuccess)
            state = settle(made, 3.0, &polls);
        std::printf("absent peer: init call returned after %.3f s (%s), state after 3 s of polling: %s -- non-blocking creation honoured\n",
                    now() - t0, ncclGetErrorString(init_result), ncclGetErrorString(state));
        std::fflush(stdout);
        _exit(state == ncclInProgress ? 0 : 3);
    }
    ncclResult_t r = ncclCommInitRankConfig(&comm, 1, id, 0, &config);
    const double t_call = now() - t0;
    if (r != ncclSuccess && r != ncclInProgress) {
        std::printf("ncclCommInitRankConfig: %s\n", ncclGetErrorString(r));
        return 2;
    }
    ncclResult_t state = settle(comm, 60.0, &polls);
    if (state != ncclSuccess) {
        std::printf("creation ended in %s\n", ncclGetErrorString(state));
        return 3;
    }
    const size_t n = 1 << 20;
    float *buf = nullptr, *out = nullptr;
    hipStream_t s;
    hipEvent_t ev;
    if (hipMalloc((void **)&buf, n * 4) != hipSuccess || hipMalloc((void **)&out, n * 4) != hipSuccess ||
        hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
        return 2;
    std::vector<float> host(n);
    for (size_t i = 0; i < n; ++i)
        host[i] = (float)(i & 0xffff);
    hipMemcpy(buf, host.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(out, 0, n * 4);
    int group_polls = 0, in_progress_answers = 0;
    for (int round = 0; round < 3; ++round) {
        ncclResult_t g = ncclGroupStart();
        if (g == ncclSuccess)
            g = ncclAllGather(buf, buf, n, ncclFloat, comm, s);           // one rank, in place
        if (g == ncclSuccess)
            g = ncclSend(buf, n, ncclFloat, 0, comm, s);                   // to itself
        if (g == ncclSuccess)
            g = ncclRecv(out, n, ncclFloat, 0, comm, s);
        ncclResult_t e = ncclGroupEnd();
        if (g != ncclSuccess || (e != ncclSuccess && e != ncclInProgress)) {
            std::printf("group: %s / %s\n", ncclGetErrorString(g), ncclGetErrorString(e));
            return 3;
        }
        in_progress_answers += e == ncclInProgress;
        state = settle(comm, 60.0, &group_polls);
        if (state != ncclSuccess) {
            std::printf("group ended in %s\n", ncclGetErrorString(state));
            return 3;
        }
        hipEventRecord(ev, s);   // behind the collective only because the group has settled
    }
    hipEventSynchronize(ev);
    std::vector<float> got(n);
    hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i)
        bad += got[i] != host[i];
    int count = 0;
    ncclCommCount(comm, &count);
    ncclResult_t d = ncclCommDestroy(comm);
    std::printf("one rank: init call %.3f s (%s), %d polls to ready; 3 groups: %d answered ncclInProgress, %d polls; %zu wrong words; "
                "ranks %d; destroy: %s\n", t_call, ncclGetErrorString(r), polls, in_progress_answers, group_polls, bad, count,
                ncclGetErrorString(d));
    return bad == 0 && count == 1 ? 0 : 3;
}
