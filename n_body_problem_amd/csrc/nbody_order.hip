// nbody_order.hip -- the body order of nbody_morton_order (include/nbody.h) computed ON THE DEVICE, and the row gathers
// that apply it: a layout refresh without a host copy of the state.
//
// The reference keeps whatever order its loader produced (main_project/kernel.cu:190-556); order is not part of the
// physics.  The layout (bodies along a Morton curve, the bodies of one mass together when the set has few distinct masses)
// is this build's own: it lowers the operand toggling of the force kernels, and it decays as the bodies move.  The host
// helper needs the state on the host (download, threaded radix sort, upload: 0.13-0.3 s at N = 2^20); these kernels
// produce the SAME permutation from the positions where they are, in well under a millisecond of device time:
//
//   order_scan_kernel      bounding cube of the finite positions (atomic min / max on order-preserving integer images of
//                          the floats) and the set of distinct mass bit patterns (a 17-slot table filled with 64-bit
//                          compare-and-swap: at most NBODY_ORDER_MAX_SPECIES + 1 insertions ever succeed)
//   order_params_kernel    one thread: cube -> origin and scale in double (the host's arithmetic), species table sorted
//                          by value as the host sorts it
//   order_keys_kernel      key = species rank << 60 | 3 x 20 interleaved bits, in double exactly as the host computes it
//   rocprim radix sort     stable, 64-bit keys, 32-bit payload = the body's index: ties keep the index order, like the
//                          host's LSD sort (HBM-bound: 8 passes over 12 B x N)
//   gather kernels         dst[k] = src[perm[k]] for float4 rows, floats and the composition of two permutations
//
// Every step is a pure function of the n float4 {x, y, z, m}: each rank of a multi-GPU run sorts its own replica and
// gets the identical permutation (tests/test_body_order.py compares with nbody_morton_order bit for bit).
#include "nbody_kernels.h"

#include <cstring>  // rocprim's headers call memset without including it

#include <rocprim/device/device_radix_sort.hpp>

namespace nbody {

namespace {

constexpr int kSpeciesSlots = 17;  // NBODY_ORDER_MAX_SPECIES + 1: the 17th distinct mass says "too many"

struct OrderScan {  // filled by order_scan_kernel; zero-initialised except lo/hi
    unsigned lo[3], hi[3];                   // ordered images of the smallest / largest finite coordinate
    unsigned any;                            // a body with three finite coordinates exists
    unsigned overflow;                       // more than kSpeciesSlots distinct masses met
    unsigned long long slot[kSpeciesSlots];  // 1 << 32 | mass bits, 0 = empty
};

struct OrderParams {  // what order_keys_kernel needs
    double lo[3], scale;
    unsigned species[16];
    int n_species;  // 0: more than 16 distinct masses, the rank is 0 for every body
};

// order-preserving map float -> unsigned (finite values and infinities; NaNs never get here)
__device__ __forceinline__ unsigned ordered(float f)
{
    const unsigned b = __builtin_bit_cast(unsigned, f);
    return b & 0x80000000u ? ~b : b | 0x80000000u;
}
__device__ __forceinline__ float unordered(unsigned u)
{
    return __builtin_bit_cast(float, u & 0x80000000u ? u & 0x7fffffffu : ~u);
}

__device__ __forceinline__ bool finite3(const float4 &p)
{
    return __builtin_isfinite(p.x) && __builtin_isfinite(p.y) && __builtin_isfinite(p.z);
}

__global__ __launch_bounds__(kTile) void order_scan_init_kernel(OrderScan *s)
{
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            s->lo[a] = 0xffffffffu;
            s->hi[a] = 0u;
        }
        s->any = s->overflow = 0u;
        for (int j = 0; j < kSpeciesSlots; ++j)
            s->slot[j] = 0ull;
    }
}

__global__ __launch_bounds__(kTile) void order_scan_kernel(const float4 *pos, int n, OrderScan *s)
{
    __shared__ unsigned w_lo[3], w_hi[3], w_any;
    if (threadIdx.x < 3) {
        w_lo[threadIdx.x] = 0xffffffffu;
        w_hi[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0)
        w_any = 0u;
    __syncthreads();
    const int stride = gridDim.x * kTile;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < n; i += stride) {
        const float4 p = pos[i];
        if (finite3(p)) {
            atomicMin(&w_lo[0], ordered(p.x));
            atomicMax(&w_hi[0], ordered(p.x));
            atomicMin(&w_lo[1], ordered(p.y));
            atomicMax(&w_hi[1], ordered(p.y));
            atomicMin(&w_lo[2], ordered(p.z));
            atomicMax(&w_hi[2], ordered(p.z));
            w_any = 1u;
        }
        // the set of distinct masses (a set: the order of insertion does not matter, order_params_kernel sorts it)
        const unsigned bits = __builtin_bit_cast(unsigned, p.w);
        if (i > 0 && __builtin_bit_cast(unsigned, pos[i - 1].w) == bits)
            continue;  // a run of one mass is entered by its first body (speed only)
        if (__hip_atomic_load(&s->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            continue;
        const unsigned long long want = 1ull << 32 | bits;
        int j = 0;
        for (; j < kSpeciesSlots; ++j) {
            unsigned long long have = __hip_atomic_load(&s->slot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (have == 0ull) {
                have = atomicCAS(&s->slot[j], 0ull, want);
                if (have == 0ull)
                    break;  // inserted
            }
            if (have == want)
                break;  // known
        }
        if (j == kSpeciesSlots)
            __hip_atomic_store(&s->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x < 3 && w_any) {
        atomicMin(&s->lo[threadIdx.x], w_lo[threadIdx.x]);
        atomicMax(&s->hi[threadIdx.x], w_hi[threadIdx.x]);
    }
    if (threadIdx.x == 0 && w_any)
        __hip_atomic_store(&s->any, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the host's comparison of two mass bit patterns: by value; NaNs and signed zeros by bit pattern
__device__ __forceinline__ bool species_less(unsigned a, unsigned b)
{
    const float fa = __builtin_bit_cast(float, a), fb = __builtin_bit_cast(float, b);
    return fa < fb || (!(fb < fa) && a < b);
}

__global__ void order_params_kernel(const OrderScan *s, OrderParams *p)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    double extent = 0.0;
    for (int a = 0; a < 3; ++a) {
        const float lo = s->any ? unordered(s->lo[a]) : 0.f, hi = s->any ? unordered(s->hi[a]) : 0.f;
        p->lo[a] = (double)lo;
        const double e = (double)hi - (double)lo;
        extent = e > extent ? e : extent;
    }
    p->scale = extent > 0.0 ? 1048575.0 / extent : 0.0;  // 20 bits per axis
    int count = 0;
    for (int j = 0; j < kSpeciesSlots; ++j)
        count += s->slot[j] != 0ull;
    const bool few = !s->overflow && count <= 16;
    p->n_species = 0;
    if (few) {
        unsigned v[kSpeciesSlots];  // one entry per slot: with v[16] the compiler, free to assume that the 17th possible
        int m = 0;                  // store never happens, dropped the whole branch for exactly 16 species
        for (int j = 0; j < kSpeciesSlots; ++j)
            if (s->slot[j] != 0ull)
                v[m++] = (unsigned)s->slot[j];
        for (int i = 1; i < m; ++i) {  // insertion sort of at most 16 values
            const unsigned x = v[i];
            int k = i;
            for (; k > 0 && species_less(x, v[k - 1]); --k)
                v[k] = v[k - 1];
            v[k] = x;
        }
        for (int i = 0; i < m; ++i)
            p->species[i] = v[i];
        p->n_species = m;
    }
}

__device__ __forceinline__ unsigned long long spread3(unsigned long long v)  // bit k of v to bit 3k
{
    v &= 0xfffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Entry i of the sequence to sort is the body in slot index[i] when `by_index` (the sequence then runs in the order of the
// caller's indices: the stable sort breaks ties by them, whatever order the slots are in), else the body in slot i.
__global__ __launch_bounds__(kTile) void order_keys_kernel(const float4 *pos, int n, const OrderParams *params,
                                                           unsigned long long *keys, unsigned *index, int by_index)
{
    __shared__ OrderParams P;
    if (threadIdx.x == 0)
        P = *params;
    __syncthreads();
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= n)
        return;
    const unsigned slot = by_index ? index[i] : (unsigned)i;
    const float4 p = pos[slot];
    unsigned long long curve = (1ull << 60) - 1;  // bodies without a finite position last within their species
    if (finite3(p)) {
        const float c[3] = {p.x, p.y, p.z};
        unsigned long long q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double t = ((double)c[a] - P.lo[a]) * P.scale;
            t = t > 0.0 ? t : 0.0;  // std::max(0.0, t): 0.0 when t is NaN cannot happen (finite inputs)
            t = t < 1048575.0 ? t : 1048575.0;
            q[a] = (unsigned long long)t;
        }
        curve = spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2;
    }
    unsigned long long rank = 0;
    if (P.n_species > 0) {
        const unsigned bits = __builtin_bit_cast(unsigned, p.w);
        int r = 0;
        while (r < P.n_species && P.species[r] != bits)
            ++r;
        rank = (unsigned long long)r;
    }
    keys[i] = rank << 60 | curve;
    index[i] = slot;
}

template <typename T>
__global__ __launch_bounds__(kTile) void gather_kernel(T *dst, const T *src, const unsigned *perm, int first, int count)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < count)
        dst[j] = src[perm[first + j]];
}

__global__ __launch_bounds__(kTile) void identity_kernel(unsigned *perm, int first, int count)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < count)
        perm[first + j] = (unsigned)(first + j);
}

__global__ __launch_bounds__(kTile) void widen_kernel(int64_t *dst, const unsigned *src, int n)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < n)
        dst[j] = (int64_t)src[j];
}

__global__ __launch_bounds__(kTile) void iota64_kernel(int64_t *dst, int n)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < n)
        dst[j] = j;
}

__global__ __launch_bounds__(kTile) void set_perm_kernel(unsigned *perm, const int64_t *order, int n, int inverse)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j >= n)
        return;
    const int64_t o = order[j];
    if (!inverse)
        perm[j] = (unsigned)o;
    else if (o >= 0 && o < n)
        perm[o] = (unsigned)j;
}

inline dim3 blocks_for(int n) { return dim3((unsigned)((n + kTile - 1) / kTile)); }

}  // namespace

size_t order_scratch_bytes(int n)
{
    // OrderScan + OrderParams (256 B each is plenty), two key arrays, two index arrays, the sort's own temporary storage
    size_t sort_bytes = 0;
    unsigned long long *k = nullptr;
    unsigned *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, k, k, v, v, (unsigned)(n > 0 ? n : 1), 0u, 64u, (hipStream_t) nullptr);
    const size_t n8 = ((size_t)(n > 0 ? n : 1) * 8 + 255) / 256 * 256, n4 = ((size_t)(n > 0 ? n : 1) * 4 + 255) / 256 * 256;
    return 512 + 2 * n8 + 2 * n4 + (sort_bytes + 255) / 256 * 256;
}

// perm[k] (k < n) = the slot of `pos` whose body belongs at slot k: nbody_morton_order of the n bodies, as 32-bit indices.
// `order` (n int64 on the device, or NULL = the identity) names the caller's index of the body in each slot: equal keys
// keep the order of the CALLER's indices, so the result is nbody_morton_order of the bodies in the caller's order whatever
// order the slots are in -- a pure function of the body set.  `scratch` holds order_scratch_bytes(n) bytes.  Asynchronous
// on `stream`; `perm` may not alias the scratch.
hipError_t launch_morton_order(const float4 *pos, int n, unsigned *perm, void *scratch, hipStream_t stream, const int64_t *order)
{
    if (n <= 0)
        return hipSuccess;
    char *base = static_cast<char *>(scratch);
    OrderScan *scan = reinterpret_cast<OrderScan *>(base);
    OrderParams *params = reinterpret_cast<OrderParams *>(base + 256);
    static_assert(sizeof(OrderScan) <= 256 && sizeof(OrderParams) <= 256, "scratch header");
    const size_t n8 = ((size_t)n * 8 + 255) / 256 * 256, n4 = ((size_t)n * 4 + 255) / 256 * 256;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(base + 512);
    unsigned long long *keys2 = reinterpret_cast<unsigned long long *>(base + 512 + n8);
    unsigned *index = reinterpret_cast<unsigned *>(base + 512 + 2 * n8);
    void *sort_tmp = base + 512 + 2 * n8 + 2 * n4;
    (void)n4;
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, index, perm, (unsigned)n, 0u, 64u, stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(order_scan_init_kernel, dim3(1), dim3(kTile), 0, stream, scan);
    const int scan_blocks = (n + kTile - 1) / kTile < 1024 ? (n + kTile - 1) / kTile : 1024;
    hipLaunchKernelGGL(order_scan_kernel, dim3((unsigned)scan_blocks), dim3(kTile), 0, stream, pos, n, scan);
    hipLaunchKernelGGL(order_params_kernel, dim3(1), dim3(64), 0, stream, scan, params);
    if (order)  // index[c] = the slot that holds the caller's body c
        hipLaunchKernelGGL(set_perm_kernel, blocks_for(n), dim3(kTile), 0, stream, index, order, n, 1);
    hipLaunchKernelGGL(order_keys_kernel, blocks_for(n), dim3(kTile), 0, stream, pos, n, params, keys, index, order ? 1 : 0);
    e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    return rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys2, index, perm, (unsigned)n, 0u, 64u, stream);
}

hipError_t launch_identity_perm(unsigned *perm, int first, int count, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(identity_kernel, blocks_for(count), dim3(kTile), 0, stream, perm, first, count);
    return hipGetLastError();
}

hipError_t launch_gather_float4(float4 *dst, const float4 *src, const unsigned *perm, int first, int count, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(gather_kernel<float4>, blocks_for(count), dim3(kTile), 0, stream, dst, src, perm, first, count);
    return hipGetLastError();
}

hipError_t launch_gather_float(float *dst, const float *src, const unsigned *perm, int first, int count, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(gather_kernel<float>, blocks_for(count), dim3(kTile), 0, stream, dst, src, perm, first, count);
    return hipGetLastError();
}

hipError_t launch_gather_int64(int64_t *dst, const int64_t *src, const unsigned *perm, int first, int count, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(gather_kernel<int64_t>, blocks_for(count), dim3(kTile), 0, stream, dst, src, perm, first, count);
    return hipGetLastError();
}

hipError_t launch_widen_perm(int64_t *dst, const unsigned *src, int n, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(widen_kernel, blocks_for(n), dim3(kTile), 0, stream, dst, src, n);
    return hipGetLastError();
}

hipError_t launch_set_perm(unsigned *perm, const int64_t *order, int n, bool inverse, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(set_perm_kernel, blocks_for(n), dim3(kTile), 0, stream, perm, order, n, inverse ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_iota64(int64_t *dst, int n, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(iota64_kernel, blocks_for(n), dim3(kTile), 0, stream, dst, n);
    return hipGetLastError();
}

}  // namespace nbody
