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

// ---- hand-allocated 64-step loop of an off-diagonal tile (same arithmetic and order as the C++ loop below) -----
// Phases per step as in force_kernel_r4 (nbody_kernels.hip): 4 x (sub,sub,sub,fma,fma,fma) . 4 x v_rsq_f32 . next
// ds_read_b128 . 4 x (4 mul, 6 fma) . rotate the column accumulators one lane.  No idle gap here: a 1024-thread
// workgroup leaves 4 waves per SIMD, too few to hide it (measured: 0/8/16/24/32 wait states = 220/220/230/246/244 ms).
// VGPR banks (index mod 4; src0/src1 never share one):
//   v[2:5] / v[6:9]   column body {x,y,z,m} = banks 2,3,0,1, double-buffered        v10 = 1e-24 (GUARD)  v11 = eps^2
//   row k=0..3        {x,y,z,m} = v[12+4k : 15+4k] = banks 0,1,2,3
//   temps k           {dx,dy,dz, r2/inv/s_row} = v[28+4k : 31+4k] = banks 0,1,2,3
//   shared            inv^2 = v44/v48 (bank 0)   inv^3 = v46/v50 (bank 2)   s_col = v47/v51 (bank 3)
//   row sums          v52..v63      column sums v64,v65,v66 (travelling)     v0 = LDS byte address
typedef float nb_f4 __attribute__((ext_vector_type(4)));
#define SY_PRE(PX, PY, PZ, X, Y, Z, D0, D1, D2, R, GRD)                                                          \
    "v_sub_f32_e32 " D0 ", " PX ", " X "\n\tv_sub_f32_e32 " D1 ", " PY ", " Y "\n\tv_sub_f32_e32 " D2 ", " PZ ", " Z "\n\t" \
    "v_fma_f32 " R ", " D0 ", " D0 ", v11\n\tv_fmac_f32_e32 " R ", " D1 ", " D1 "\n\tv_fmac_f32_e32 " R ", " D2 ", " D2 "\n\t" GRD(R)
#define SY_NOGUARD(R) ""
#define SY_GUARD(R) "v_max_f32_e32 " R ", v10, " R "\n\t"
#define SY_POST(PM, M, AX, AY, AZ, D0, D1, D2, R, Q, T, SC)                                                      \
    "v_mul_f32_e32 " Q ", " R ", " R "\n\tv_mul_f32_e32 " T ", " R ", " Q "\n\t"                                      \
    "v_mul_f32_e32 " SC ", " M ", " T "\n\tv_mul_f32_e32 " R ", " PM ", " T "\n\t"                                    \
    "v_fmac_f32_e32 " AX ", " D0 ", " R "\n\tv_fmac_f32_e32 " AY ", " D1 ", " R "\n\tv_fmac_f32_e32 " AZ ", " D2 ", " R "\n\t" \
    "v_fmac_f32_e32 v64, " D0 ", " SC "\n\tv_fmac_f32_e32 v65, " D1 ", " SC "\n\tv_fmac_f32_e32 v66, " D2 ", " SC "\n\t"
#define SY_STEP(PX, PY, PZ, PM, NEXT, GRD)                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    SY_PRE(PX, PY, PZ, "v12", "v13", "v14", "v28", "v29", "v30", "v31", GRD)                                     \
    SY_PRE(PX, PY, PZ, "v16", "v17", "v18", "v32", "v33", "v34", "v35", GRD)                                     \
    SY_PRE(PX, PY, PZ, "v20", "v21", "v22", "v36", "v37", "v38", "v39", GRD)                                     \
    SY_PRE(PX, PY, PZ, "v24", "v25", "v26", "v40", "v41", "v42", "v43", GRD)                                     \
    "v_rsq_f32_e32 v31, v31\n\tv_rsq_f32_e32 v35, v35\n\tv_rsq_f32_e32 v39, v39\n\tv_rsq_f32_e32 v43, v43\n\t"       \
    NEXT                                                                                                         \
    SY_POST(PM, "v15", "v52", "v53", "v54", "v28", "v29", "v30", "v31", "v44", "v46", "v47")                     \
    SY_POST(PM, "v19", "v55", "v56", "v57", "v32", "v33", "v34", "v35", "v48", "v50", "v51")                     \
    SY_POST(PM, "v23", "v58", "v59", "v60", "v36", "v37", "v38", "v39", "v44", "v46", "v47")                     \
    SY_POST(PM, "v27", "v61", "v62", "v63", "v40", "v41", "v42", "v43", "v48", "v50", "v51")                     \
    "s_nop 1\n\t"                                                                                                \
    "v_mov_b32_dpp v64, v64 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"                                           \
    "v_mov_b32_dpp v65, v65 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"                                           \
    "v_mov_b32_dpp v66, v66 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
#define SY_GROUP_LOOP(GRD)                                                                                       \
    "ds_read_b128 v[2:5], v0\n\t"                                                                                \
    "s_mov_b32 %[cnt], 16\n"                                                                                     \
    "1:\n\t"                                                                                                     \
    SY_STEP("v2", "v3", "v4", "v5", "ds_read_b128 v[6:9], v0 offset:16\n\t", GRD)                                \
    SY_STEP("v6", "v7", "v8", "v9", "ds_read_b128 v[2:5], v0 offset:32\n\t", GRD)                                \
    SY_STEP("v2", "v3", "v4", "v5", "ds_read_b128 v[6:9], v0 offset:48\n\t", GRD)                                \
    SY_STEP("v6", "v7", "v8", "v9", "v_add_u32_e32 v0, 64, v0\n\tds_read_b128 v[2:5], v0\n\t", GRD)               \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_waitcnt lgkmcnt(0)\n"

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

            if (!DIAG) {
                unsigned lds = (unsigned)(size_t)(&stage[lane]);
                unsigned cnt;
                const nb_f4 r0 = {x[0], y[0], z[0], m[0]}, r1 = {x[1], y[1], z[1], m[1]};
                const nb_f4 r2v = {x[2], y[2], z[2], m[2]}, r3 = {x[3], y[3], z[3], m[3]};
                const float tiny = 1.0e-24f;
#define SY_OPERANDS                                                                                                   \
                : "+{v52}"(ax[0]), "+{v53}"(ay[0]), "+{v54}"(az[0]), "+{v55}"(ax[1]), "+{v56}"(ay[1]), "+{v57}"(az[1]),      \
                  "+{v58}"(ax[2]), "+{v59}"(ay[2]), "+{v60}"(az[2]), "+{v61}"(ax[3]), "+{v62}"(ay[3]), "+{v63}"(az[3]),      \
                  "+{v64}"(cx), "+{v65}"(cy), "+{v66}"(cz), "+{v0}"(lds), [cnt] "=&s"(cnt)                                 \
                : "{v[12:15]}"(r0), "{v[16:19]}"(r1), "{v[20:23]}"(r2v), "{v[24:27]}"(r3), "{v11}"(eps2), "{v10}"(tiny)      \
                : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",      \
                  "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v46", "v47", "v48", "v50", "v51", "scc",    \
                  "memory"
                if (GUARD)
                    asm volatile(SY_GROUP_LOOP(SY_GUARD) SY_OPERANDS);
                else
                    asm volatile(SY_GROUP_LOOP(SY_NOGUARD) SY_OPERANDS);
#undef SY_OPERANDS
            } else {
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
                    {  // rows and columns are the same bodies: keep row < column (drops the self pair too)
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
                    // keep each pair's dependent chain contiguous
                    __builtin_amdgcn_sched_barrier(0);
                }
                cx = wave_rol1(cx);
                cy = wave_rol1(cy);
                cz = wave_rol1(cz);
            }
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
