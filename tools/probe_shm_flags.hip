// probe_shm_flags.hip -- does the mechanism the asynchronous RCCL test double (tests/fake_rccl) relies on work on this pool?
//   * a POSIX shared-memory segment mapped by two PROCESSES, registered with hipHostRegister in each (pinned, device-visible);
//   * hipMemcpyAsync device <-> that segment on a stream (asynchronous: the call returns before the copy has run);
//   * a one-wave kernel of process A spinning (bounded by wall_clock64) on a flag that a kernel of process B writes, both
//     processes on the SAME GPU at the same time.
// Usage: probe_shm_flags <rank 0|1> <name>      (start both; prints one line each)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#define CK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            std::printf("rank %d: %s failed: %s\n", g_rank, #x, hipGetErrorString(e_));              \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)

static int g_rank = 0;

__global__ void signal_kernel(uint32_t *flag, uint32_t value)
{
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void wait_kernel(uint32_t *flag, uint32_t want, uint32_t *status, long long timeout_ticks)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

__global__ void busy_kernel(float *x, int iters)  // something that takes a while, in front of the send
{
    float v = x[threadIdx.x];
    for (int i = 0; i < iters; ++i)
        v = v * 1.0000001f + 1e-7f;
    x[threadIdx.x] = v;
}

int main(int argc, char **argv)
{
    if (argc < 3)
        return 1;
    g_rank = atoi(argv[1]);
    const char *name = argv[2];
    const size_t bytes = 64u << 20, data_off = 4096, n = 4u << 20;  // 16 MiB of floats per rank
    int fd = -1;
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {
        fd = shm_open(name, g_rank == 0 ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
        if (fd < 0)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (fd < 0) {
        std::printf("rank %d: shm_open failed\n", g_rank);
        return 2;
    }
    if (g_rank == 0 && ftruncate(fd, (off_t)bytes) != 0)
        return 2;
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    void *mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (mem == MAP_FAILED) {
        std::printf("rank %d: mmap failed\n", g_rank);
        return 2;
    }
    CK(hipSetDevice(0));
    CK(hipHostRegister(mem, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    void *dmem = nullptr;
    CK(hipHostGetDevicePointer(&dmem, mem, 0));
    uint32_t *flags = static_cast<uint32_t *>(dmem);          // [0]: rank 0's data flag, [64]: rank 1's, [128]: status
    char *hslot = static_cast<char *>(mem) + data_off + (size_t)g_rank * (n * 4);
    char *hpeer = static_cast<char *>(mem) + data_off + (size_t)(1 - g_rank) * (n * 4);
    float *dsend = nullptr, *drecv = nullptr;
    CK(hipMalloc((void **)&dsend, n * 4));
    CK(hipMalloc((void **)&drecv, n * 4));
    std::vector<float> host(n);
    for (size_t i = 0; i < n; ++i)
        host[i] = (float)(g_rank * 1000000 + (int)(i & 0xffff));
    CK(hipMemcpy(dsend, host.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (g_rank == 0)
        std::this_thread::sleep_for(std::chrono::milliseconds(300));  // rank 1 gets to its wait kernel first
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t round = 1; round <= 3; ++round) {
        hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s, dsend, 2000000);                        // ~ms in front of the send
        CK(hipMemcpyAsync(hslot, dsend, n * 4, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, flags + 64 * g_rank, round);
        hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, s, flags + 64 * (1 - g_rank), round, flags + 128, 20ll * 100000000ll);
        CK(hipMemcpyAsync(drecv, hpeer, n * 4, hipMemcpyHostToDevice, s));
    }
    const double enqueue_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(s));
    const double total_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(host.data(), drecv, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {  // the peer's dsend after three busy kernels on its first 64 entries
        const float want = (float)((1 - g_rank) * 1000000 + (int)(i & 0xffff));
        if (i >= 64 && host[i] != want)
            ++bad;
    }
    uint32_t status = static_cast<uint32_t *>(mem)[128];
    std::printf("rank %d: enqueue of 3 rounds %.2f ms, complete after %.1f ms, %zu wrong words, timeout flag %u -> %s\n", g_rank,
                enqueue_ms, total_ms, bad, status, bad == 0 && status == 0 ? "OK" : "FAILED");
    CK(hipHostUnregister(mem));
    munmap(mem, bytes);
    if (g_rank == 0)
        shm_unlink(name);
    return bad == 0 && status == 0 ? 0 : 3;
}
