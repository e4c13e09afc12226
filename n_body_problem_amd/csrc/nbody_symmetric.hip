// nbody_symmetric.hip -- EXPERIMENTAL pair-once force kernel.
//
// The reference's own contribution ("method C", main_project/kernel.cu:703-774) evaluates each unordered pair
// once on an upper-triangular grid of 256 x 256 tiles and applies it to both bodies (Newton's third law) through
// shared-memory and global float atomics -- which its author names as the bottleneck (kernel.cu:757) and which
// make the result order-dependent.  This kernel keeps the idea and drops the atomics:
//
//   * tiles are pairs of column splits (I <= J), split_len bodies each; ONE 1024-thread workgroup (16 wave64) per
//     tile, so every partial sum P[split][body] is produced by exactly one workgroup:
//        rows b in I, columns c in J:  P[J][b] = sum_c m_c f(b,c)   (row side, registers)
//                                      P[I][c] = -sum_b m_b f(b,c)  (column side, LDS)
//     which is the SAME partial-sum array the one-sided kernel fills -- the update kernel does not change;
//   * inside a wave, lane l owns R rows and at step s meets column (l+s) mod 64 of the wave's current 64-column
//     group; the three column accumulators travel with the column, one lane per step (DPP wave_rol:1), so after 64
//     steps column c's sum sits in lane c and is added to the LDS array without conflicts;
//   * the 16 waves walk the column groups in a rotated order, G/16 groups apart, with a barrier every G/16 groups, so
//     no two waves touch the same LDS entries at a time and every entry receives its terms in a fixed order:
//     bit-reproducible.
//   * diagonal tiles (I == J) visit every (row, column) combination and keep the pairs with row index < column index.
//
// Per unordered pair: 3 sub, 3 fma, rsq, 4 mul, 6 fma = 16 VALU + 1 transcendental (+ 3 DPP moves per 64 x R pairs)
// against 2 x (12 + 1) for the two ordered interactions it replaces.
#include "nbody_kernels.h"

namespace nbody {

constexpr int kSymThreads = 1024;
constexpr int kSymWaves = kSymThreads / 64;
constexpr int kSymRows = 4;  // rows per lane

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    const int i = __builtin_bit_cast(int, v);  // every lane is written by a wave rotate: "old" is never used
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
// lane i receives the value of lane (i + 1) mod 64
__device__ __forceinline__ float wave_rol1(float v) { return dpp_move<0x134>(v); }

template <bool DIAG, bool GUARD>
__device__ __forceinline__ void sym_tile(const SymArgs &a, int I, int J, float *sx, float *sy, float *sz, float4 *stage)
{
    const int L = a.split_len, G = L / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rowbase = I * L, colbase = J * L;
    const int rows_per_pass = kSymWaves * 64 * kSymRows;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float eps2;  // in a VGPR: an SGPR source operand costs an fp32 instruction two extra cycles on gfx950
    asm volatile("v_mov_b32 %0, %1" : "=v"(eps2) : "s"(a.eps2));

    for (int c = tid; c < L; c += kSymThreads)
        sx[c] = sy[c] = sz[c] = 0.f;
    __syncthreads();

    for (int pass0 = 0; pass0 < L; pass0 += rows_per_pass) {
        float x[kSymRows], y[kSymRows], z[kSymRows], m[kSymRows], ax[kSymRows], ay[kSymRows], az[kSymRows];
        int rl[kSymRows];  // row index inside the split, or -1
#pragma unroll
        for (int k = 0; k < kSymRows; ++k) {
            const int r = pass0 + (wave * kSymRows + k) * 64 + lane;
            float4 p = zero4;
            rl[k] = -1;
            if (r < L && rowbase + r < a.n_total) {
                p = a.pos[rowbase + r];
                rl[k] = r;
            }
            x[k] = p.x; y[k] = p.y; z[k] = p.z; m[k] = p.w;
            ax[k] = ay[k] = az[k] = 0.f;
        }

        // Wave w starts `spacing` groups after wave w-1 and all walk the groups in the same direction, so two waves
        // can only meet on a group if one gets `spacing` rounds ahead: a barrier every `spacing` rounds rules that
        // out and fixes the order in which the waves' terms reach each LDS entry.
        const int spacing = G / kSymWaves;  // >= 1 (the host requires split_len >= 1024)
        for (int g = 0; g < G; ++g) {
            int cg = g + spacing * wave;
            if (cg >= G)
                cg -= G;
            const int gc = colbase + cg * 64 + lane;
            float4 c = zero4;
            if (gc < a.n_total)
                c = a.pos[gc];
            stage[lane] = c;       // twice, so that lane + s never wraps
            stage[lane + 64] = c;
            float cx = 0.f, cy = 0.f, cz = 0.f;  // accumulators of column (lane + s) mod 64, travelling

#pragma unroll 4
            for (int s = 0; s < 64; ++s) {
                // (prefetching the next step's column into a second register set measured 5 % slower)
                const float4 pj = stage[lane + s];
#pragma unroll
                for (int k = 0; k < kSymRows; ++k) {
                    const float dx = pj.x - x[k], dy = pj.y - y[k], dz = pj.z - z[k];
                    float r2 = __builtin_fmaf(dx, dx, eps2);
                    r2 = __builtin_fmaf(dy, dy, r2);
                    r2 = __builtin_fmaf(dz, dz, r2);
                    if (GUARD)
                        r2 = __builtin_fmaxf(r2, 1.0e-24f);
                    const float inv = __builtin_amdgcn_rsqf(r2);
                    float inv3 = inv * (inv * inv);
                    if (DIAG) {  // rows and columns are the same bodies: keep row < column (drops the self pair too)
                        const int col = cg * 64 + ((lane + s) & 63);
                        inv3 = (rl[k] >= 0 && rl[k] < col) ? inv3 : 0.f;
                    }
                    const float sr = pj.w * inv3, sc = m[k] * inv3;
                    ax[k] = __builtin_fmaf(dx, sr, ax[k]);
                    ay[k] = __builtin_fmaf(dy, sr, ay[k]);
                    az[k] = __builtin_fmaf(dz, sr, az[k]);
                    cx = __builtin_fmaf(dx, sc, cx);
                    cy = __builtin_fmaf(dy, sc, cy);
                    cz = __builtin_fmaf(dz, sc, cz);
                    // keep each pair's dependent chain contiguous: fp32 instructions issued between a v_rsq_f32 and
                    // the use of its result run at half rate on gfx950, so interleaving the
                    // rows, which the scheduler would otherwise do, is slower than one chain after the other
                    __builtin_amdgcn_sched_barrier(0);
                }
                cx = wave_rol1(cx);
                cy = wave_rol1(cy);
                cz = wave_rol1(cz);
            }
            // after 64 rotations lane l holds column l of the group; force on the column body is -m_row * d * inv3
            sx[cg * 64 + lane] -= cx;
            sy[cg * 64 + lane] -= cy;
            sz[cg * 64 + lane] -= cz;
            if ((g + 1) % spacing == 0)
                __syncthreads();
        }
        __syncthreads();

        // row sums of this pass
        if (DIAG) {
#pragma unroll
            for (int k = 0; k < kSymRows; ++k)
                if (rl[k] >= 0) {
                    sx[rl[k]] += ax[k];
                    sy[rl[k]] += ay[k];
                    sz[rl[k]] += az[k];
                }
        } else {
            float4 *out = a.partials + (size_t)J * a.n_total;
#pragma unroll
            for (int k = 0; k < kSymRows; ++k)
                if (rl[k] >= 0)
                    out[rowbase + rl[k]] = make_float4(ax[k], ay[k], az[k], 0.f);
        }
        __syncthreads();
    }

    float4 *out = a.partials + (size_t)I * a.n_total;
    for (int c = tid; c < L; c += kSymThreads)
        if (colbase + c < a.n_total)
            out[colbase + c] = make_float4(sx[c], sy[c], sz[c], 0.f);
}

template <bool GUARD>
__global__ __launch_bounds__(kSymThreads) void force_sym_kernel(SymArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.split_len;
    float *sx = smem, *sy = sx + L, *sz = sy + L;
    float4 *stage = reinterpret_cast<float4 *>(sz + L) + (threadIdx.x >> 6) * 128;
    const int2 t = a.tiles[blockIdx.x];
    if (t.x == t.y)
        sym_tile<true, GUARD>(a, t.x, t.y, sx, sy, sz, stage);
    else
        sym_tile<false, GUARD>(a, t.x, t.y, sx, sy, sz, stage);
}

size_t symmetric_lds_bytes(int split_len) { return (size_t)split_len * 12 + (size_t)kSymWaves * 128 * sizeof(float4); }

hipError_t launch_forces_symmetric(const SymArgs &a, hipStream_t stream)
{
    if (a.n_tiles <= 0)
        return hipSuccess;
    const size_t lds = symmetric_lds_bytes(a.split_len);
    hipError_t e;
    if (a.eps2 > 0.f) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&force_sym_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(force_sym_kernel<false>, dim3(a.n_tiles), dim3(kSymThreads), lds, stream, a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&force_sym_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(force_sym_kernel<true>, dim3(a.n_tiles), dim3(kSymThreads), lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace nbody
