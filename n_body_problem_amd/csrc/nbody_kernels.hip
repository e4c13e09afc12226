// nbody_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the all-pairs step.
//
// What the reference computes on this path (main_project/kernel.cu):
//   cal_acc_advanced          :703-774  per-body acceleration from all other bodies
//   use_acc_update_position   :777-801  v += a*dt; x += v*dt (fp64 FMA, rounded to fp32)
//   simple_update_all         :828-884  the GPU-Gems shape: 1 body per thread, 256-body shared tile
// Written from scratch for CDNA4: 64-lane wavefronts, column tiles staged in LDS and read back as
// wave-uniform (broadcast) ds_read_b128, several rows register-blocked per lane so each LDS read
// feeds rows_per_lane interactions, v_rsq_f32 for the inverse square root, issued in batches with an
// idle gap behind them (see force_kernel), no atomics (the reference's shared/global float atomics,
// kernel.cu:758-773, are what it names as its bottleneck).
// The kernel is VALU-bound; HBM traffic is the O(N) state plus the partials.
#include "nbody_kernels.h"

#include <type_traits>

namespace nbody {

// One body-body interaction: 3 sub, 3 fma (r^2+eps^2), 1 rsq, 3 mul, 3 fma = 12 fp32 VALU instructions + 1
// transcendental, counted as 20 flop.  The softening is folded into the r^2 FMA chain
// (kernel.cu:679 adds EPSILON separately, in double).
//
// Scheduling (measured on gfx950, tools/ubench_spec.hip and tools/gen_sched2.py): the
// fp32 instructions a wave issues right after its own v_rsq_f32 run at about half rate for a few tens of
// cycles, whatever they depend on.  So the RPL interactions of one column are issued in three phases --
// [RPL x (sub, sub, sub, fma, fma, fma)] [RPL x v_rsq_f32] [idle kGap wait states] [RPL x (mul, mul, mul, fma,
// fma, fma)] -- and the wave idles through the slow window while the SIMD's other waves issue at full rate:
// 33.7 SIMD cycles per interaction instead of the 40.2 of the natural row-after-row order (floor: 12 x 2.05 + 8).
template <int RPL> __device__ __forceinline__ void idle_gap() { asm volatile("s_nop 15\n\ts_nop 11"); }  // 28 wait states
template <> __device__ __forceinline__ void idle_gap<2>() { asm volatile("s_nop 15"); }                      // 16
template <> __device__ __forceinline__ void idle_gap<1>() { asm volatile("s_nop 15\n\ts_nop 15"); }         // 32

// grid.x = row tiles of kTile*RPL rows, grid.y = splits of this launch.  Each lane owns RPL rows
// (row = tile_base + k*kTile + lane-in-block, so row loads/stores are coalesced float4) and walks the
// split's columns in ascending order, one fp32 FMA chain per row: the order bit-exactly defines the
// partial sum whatever the grid, RPL or sharding.
// PPS: per-particle softening (the eps the reference loads into vel.w, kernel.cu:223, and never uses):
// eps_ij^2 = eps^2 + eps_i^2 + eps_j^2, one extra add per interaction and a second, scalar LDS tile.
template <int RPL, bool GUARD, bool PPS = false>
__global__ __launch_bounds__(kTile) void force_kernel(ForceArgs a)
{
    __shared__ float4 tile[2][kTile];
    __shared__ float etile[PPS ? 2 : 1][PPS ? kTile : 1];

    const int tid = threadIdx.x;
    int split = a.split_first + blockIdx.y;
    if (split >= a.skip_first)
        split += a.skip_count;
    const int j0 = split * a.split_len;
    const int j1 = min(j0 + a.split_len, a.n_total);
    const int ntiles = (j1 - j0 + kTile - 1) / kTile;
    const int row_base = blockIdx.x * (kTile * RPL) + tid;

    float xi[RPL], yi[RPL], zi[RPL], ax[RPL], ay[RPL], az[RPL], ei2[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = row_base + k * kTile;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        float e = 0.f;
        if (r < a.row_count) {
            p = a.pos[a.row_lo + r];
            if (PPS)
                e = a.eps_pp[a.row_lo + r];
        }
        xi[k] = p.x;
        yi[k] = p.y;
        zi[k] = p.z;
        ei2[k] = __builtin_fmaf(e, e, a.eps2);  // eps^2 + eps_i^2
        ax[k] = ay[k] = az[k] = 0.f;
    }
    float eps2;  // kept in a VGPR: an SGPR source costs an fp32 instruction two extra cycles on gfx950
    asm volatile("v_mov_b32 %0, %1" : "=v"(eps2) : "s"(a.eps2));

    // an out-of-range column is staged as a zero-mass body at the origin: it adds exactly 0
    float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
    float stage_e = 0.f;
    if (j0 + tid < j1) {
        stage = a.pos[j0 + tid];
        if (PPS)
            stage_e = a.eps_pp[j0 + tid];
    }
    tile[0][tid] = stage;
    if (PPS)
        etile[0][tid] = stage_e * stage_e;
    __syncthreads();

    // A split whose bodies all have one mass (ForceArgs::split_mass; every split of an equal-mass system): the mass
    // leaves the loop -- the sums collect d * inv^3 and are multiplied by it once at the end -- 11 fp32 instructions + 1
    // transcendental per interaction instead of 12 + 1.  Decided per split from the data, so every register blocking
    // and every sharding takes the same path for the same split: the variants stay bit-identical among themselves.
    const float split_mass = GUARD ? __builtin_nanf("") : a.split_mass[split];
    const bool uniform = split_mass == split_mass;

    auto tiles = [&](auto uniform_tag) {
    constexpr bool UNIFORM = decltype(uniform_tag)::value;
    for (int t = 0; t < ntiles; ++t) {
        const int jn = j0 + (t + 1) * kTile + tid;
        if (t + 1 < ntiles) {  // in flight under the tile's arithmetic
            stage = make_float4(0.f, 0.f, 0.f, 0.f);
            stage_e = 0.f;
            if (jn < j1) {
                stage = a.pos[jn];
                if (PPS)
                    stage_e = a.eps_pp[jn];
            }
        }

        const float4 *cur = tile[t & 1];
        const float *ecur = etile[PPS ? (t & 1) : 0];
        float4 pj = cur[0];  // wave-uniform address: broadcast ds_read_b128
        const int ncols = ((min(kTile, j1 - j0 - t * kTile) + 3) >> 2) << 2;  // a partial last tile: see force_kernel_r4
#pragma unroll 4
        for (int jj = 0; jj < ncols; ++jj) {
            float dx[RPL], dy[RPL], dz[RPL], w[RPL];
#pragma unroll
            for (int k = 0; k < RPL; ++k) {  // phase 1: separations and r^2 + eps^2, one chain per row
                dx[k] = pj.x - xi[k];
                dy[k] = pj.y - yi[k];
                dz[k] = pj.z - zi[k];
                float r2 = __builtin_fmaf(dx[k], dx[k], PPS ? ei2[k] + ecur[jj] : eps2);
                r2 = __builtin_fmaf(dy[k], dy[k], r2);
                r2 = __builtin_fmaf(dz[k], dz[k], r2);
                if (GUARD)  // eps == 0: a pair at zero distance (the self pair) must contribute 0, not NaN
                    r2 = guard_r2(r2);
                w[k] = r2;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < RPL; ++k)  // phase 2: the transcendentals, back to back
                w[k] = __builtin_amdgcn_rsqf(w[k]);
            __builtin_amdgcn_sched_barrier(0);
            const float4 pn = cur[(jj + 1) & (kTile - 1)];  // the next column's LDS read travels during the gap
            __builtin_amdgcn_sched_barrier(0);
            idle_gap<RPL>();  // phase 3: sit out the slow window
            __builtin_amdgcn_s_setprio(0);  // phases 1-2 run at priority 2, phase 4 at 0 (see NB_R4_PRIO_* below)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < RPL; ++k) {  // phase 4: m_j / (r^2+eps^2)^(3/2) and the accumulation
                const float inv = w[k];
                const float s = UNIFORM ? inv * (inv * inv) : (pj.w * inv) * (inv * inv);
                ax[k] = __builtin_fmaf(dx[k], s, ax[k]);
                ay[k] = __builtin_fmaf(dy[k], s, ay[k]);
                az[k] = __builtin_fmaf(dz[k], s, az[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(2);
            __builtin_amdgcn_sched_barrier(0);
            pj = pn;
        }
        __builtin_amdgcn_s_setprio(0);

        if (t + 1 < ntiles) {
            tile[(t + 1) & 1][tid] = stage;
            if (PPS)
                etile[(t + 1) & 1][tid] = stage_e * stage_e;
        }
        __syncthreads();
    }
    };
    if (uniform)
        tiles(std::true_type{});
    else
        tiles(std::false_type{});
    const float scale = uniform ? split_mass : 1.f;

    float4 *out = a.partials + (size_t)split * a.row_count;
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = row_base + k * kTile;
        if (r < a.row_count)
            out[r] = make_float4(ax[k] * scale, ay[k] * scale, az[k] * scale, 0.f);
    }
}

// ---- the 4-rows-per-lane kernel with a hand-allocated inner loop ------------------------------------------------
// Same arithmetic, same phases and the same per-row operation order as force_kernel<4, GUARD> (the results are
// bit-identical; tests/test_parity_gpu.py checks it), but the 256-column tile loop is one asm block with fixed VGPRs,
// because hipcc's register allocation is blind to the gfx950 VGPR banks: an fp32 VOP2/VOP3 instruction whose src0 and
// src1 sit in the same bank (register index mod 4) costs two extra cycles, and the compiled loop has four to five
// such instructions per interaction.  Allocation (bank = index mod 4):
//   v0-3 / v4-7   column body {x,y,z,m} (banks 0,1,2,3), double-buffered: the next ds_read_b128 travels in the gap
//   v8            eps^2 (only ever src2)            v35  2^-84, v53 +inf (GUARD only)          v52  LDS byte address
//   row k=0..3    x,y,z = v(9+4k), v(10+4k), v(11+4k) (banks 1,2,3)   ax = v(12+4k)   ay,az = v(25+2k), v(26+2k)
//   temps k       r2/inv/s = v(36+4k) (bank 0)   dx,dy,dz = v(37+4k)..v(39+4k) (banks 1,2,3)   inv^2 = v33/v34
#define NB_PRE(PX, PY, PZ, X, Y, Z, R, D0, D1, D2, GRD)                                                          \
    "v_sub_f32_e32 " D0 ", " PX ", " X "\n\tv_sub_f32_e32 " D1 ", " PY ", " Y "\n\tv_sub_f32_e32 " D2 ", " PZ ", " Z "\n\t" \
    "v_fma_f32 " R ", " D0 ", " D0 ", v8\n\tv_fmac_f32_e32 " R ", " D1 ", " D1 "\n\tv_fmac_f32_e32 " R ", " D2 ", " D2 "\n\t" GRD(R)
#define NB_NOGUARD(R) ""
#define NB_GUARD(R) "v_cmp_le_f32_e32 vcc, v35, " R "\n\tv_cndmask_b32_e32 " R ", v53, " R ", vcc\n\t"
#define NB_POST(PM, AX, AY, AZ, R, D0, D1, D2, Q)                                                                \
    "v_mul_f32_e32 " Q ", " R ", " R "\n\tv_mul_f32_e32 " R ", " PM ", " R "\n\tv_mul_f32_e32 " R ", " R ", " Q "\n\t"     \
    "v_fmac_f32_e32 " AX ", " D0 ", " R "\n\tv_fmac_f32_e32 " AY ", " D1 ", " R "\n\tv_fmac_f32_e32 " AZ ", " D2 ", " R "\n\t"
#define NB_POSTU(PM, AX, AY, AZ, R, D0, D1, D2, Q) /* a split of equal masses: see force_kernel */                  \
    "v_mul_f32_e32 " Q ", " R ", " R "\n\tv_mul_f32_e32 " R ", " R ", " Q "\n\t"                                      \
    "v_fmac_f32_e32 " AX ", " D0 ", " R "\n\tv_fmac_f32_e32 " AY ", " D1 ", " R "\n\tv_fmac_f32_e32 " AZ ", " D2 ", " R "\n\t"
// one column: wait for its LDS read, phase 1 for the four rows, the four rsq, issue the NEXT read, idle 24 wait
// states (20-28 measured equally good on the kernel, 12 and 32 about 2 % worse), phase 4
// Wave priority (measured after the gap was tuned): priority 2 for the PRE + v_rsq_f32 phases, 0 for the POST phase --
// the SIMD then issues a waiting wave's rsq batch before another wave's POST stretch, which does by arbitration what the
// idle gap does by waiting: 256.2-257.1 ms per N = 2^20 pass against 260.7 ms, and the gap length stops mattering
// (0 / 6 / 24 wait states: 256.2 / 256.8 / 257.1 ms).
#ifndef NB_R4_GAP
#define NB_R4_GAP "s_nop 5\n\t"
#endif
#ifndef NB_R4_PRIO_POST
#define NB_R4_PRIO_POST "s_setprio 0\n\t"
#define NB_R4_PRIO_PRE "s_setprio 2\n\t"
#endif
#define NB_COLUMN(PX, PY, PZ, PM, NEXT, GRD, POST)                                                                     \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    NB_PRE(PX, PY, PZ, "v9", "v10", "v11", "v36", "v37", "v38", "v39", GRD)                                      \
    NB_PRE(PX, PY, PZ, "v13", "v14", "v15", "v40", "v41", "v42", "v43", GRD)                                     \
    NB_PRE(PX, PY, PZ, "v17", "v18", "v19", "v44", "v45", "v46", "v47", GRD)                                     \
    NB_PRE(PX, PY, PZ, "v21", "v22", "v23", "v48", "v49", "v50", "v51", GRD)                                     \
    "v_rsq_f32_e32 v36, v36\n\tv_rsq_f32_e32 v40, v40\n\tv_rsq_f32_e32 v44, v44\n\tv_rsq_f32_e32 v48, v48\n\t"       \
    NEXT NB_R4_GAP NB_R4_PRIO_POST                                                                               \
    POST(PM, "v12", "v25", "v26", "v36", "v37", "v38", "v39", "v33")                                          \
    POST(PM, "v16", "v27", "v28", "v40", "v41", "v42", "v43", "v34")                                          \
    POST(PM, "v20", "v29", "v30", "v44", "v45", "v46", "v47", "v33")                                          \
    POST(PM, "v24", "v31", "v32", "v48", "v49", "v50", "v51", "v34")                                          \
    NB_R4_PRIO_PRE
#define NB_TILE_LOOP(GRD, POST)                                                                                        \
    NB_R4_PRIO_PRE                                                                                               \
    "ds_read_b128 v[0:3], v52\n\t"                                                                               \
    "s_mov_b32 %[cnt], %[n4]\n"                                                                                     \
    "1:\n\t"                                                                                                     \
    NB_COLUMN("v0", "v1", "v2", "v3", "ds_read_b128 v[4:7], v52 offset:16\n\t", GRD, POST)                             \
    NB_COLUMN("v4", "v5", "v6", "v7", "ds_read_b128 v[0:3], v52 offset:32\n\t", GRD, POST)                             \
    NB_COLUMN("v0", "v1", "v2", "v3", "ds_read_b128 v[4:7], v52 offset:48\n\t", GRD, POST)                             \
    NB_COLUMN("v4", "v5", "v6", "v7", "v_add_u32_e32 v52, 64, v52\n\tds_read_b128 v[0:3], v52\n\t", GRD, POST)          \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

template <bool GUARD>
__global__ __launch_bounds__(kTile) void force_kernel_r4(ForceArgs a)
{
    __shared__ float4 tile[2 * kTile + 1];  // + 1: the loop's last read-ahead lands one body past the second tile

    const int tid = threadIdx.x;
    int split = a.split_first + blockIdx.y;
    if (split >= a.skip_first)
        split += a.skip_count;
    const int j0 = split * a.split_len;
    const int j1 = min(j0 + a.split_len, a.n_total);
    const int ntiles = (j1 - j0 + kTile - 1) / kTile;
    const int row_base = blockIdx.x * (kTile * 4) + tid;
    const float split_mass = GUARD ? __builtin_nanf("") : a.split_mass[split];  // see force_kernel
    const bool uniform = split_mass == split_mass;

    // pinned for the whole kernel (local register variables): no copies around the asm block, 8 waves per SIMD
    register float x0 asm("v9"), y0 asm("v10"), z0 asm("v11"), ax0 asm("v12");
    register float x1 asm("v13"), y1 asm("v14"), z1 asm("v15"), ax1 asm("v16");
    register float x2 asm("v17"), y2 asm("v18"), z2 asm("v19"), ax2 asm("v20");
    register float x3 asm("v21"), y3 asm("v22"), z3 asm("v23"), ax3 asm("v24");
    register float ay0 asm("v25"), az0 asm("v26"), ay1 asm("v27"), az1 asm("v28");
    register float ay2 asm("v29"), az2 asm("v30"), ay3 asm("v31"), az3 asm("v32");
    register float eps2 asm("v8");
    register float tiny asm("v35");
    register float pinf asm("v53");
    {
        float4 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = row_base + k * kTile;
            p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.row_count)
                p[k] = a.pos[a.row_lo + r];
        }
        x0 = p[0].x; y0 = p[0].y; z0 = p[0].z;
        x1 = p[1].x; y1 = p[1].y; z1 = p[1].z;
        x2 = p[2].x; y2 = p[2].y; z2 = p[2].z;
        x3 = p[3].x; y3 = p[3].y; z3 = p[3].z;
    }
    ax0 = ay0 = az0 = ax1 = ay1 = az1 = ax2 = ay2 = az2 = ax3 = ay3 = az3 = 0.f;
    eps2 = a.eps2;
    tiny = kGuardMin;
    pinf = __builtin_inff();

    float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 + tid < j1)
        stage = a.pos[j0 + tid];
    tile[tid] = stage;
    if (tid == 0)
        tile[2 * kTile] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int jn = j0 + (t + 1) * kTile + tid;
        if (t + 1 < ntiles) {  // in flight under the tile's arithmetic
            stage = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jn < j1)
                stage = a.pos[jn];
        }
        register unsigned lds asm("v52") = (unsigned)(size_t)(&tile[(t & 1) * kTile]);  // LDS byte address of the tile
        // four-column iterations of this tile: all 64 but for the last tile of a split whose length is no multiple of the
        // 256-body tile (split lengths are multiples of 64; a ragged last split's padding columns are zero-mass bodies)
        const unsigned n4 = (unsigned)__builtin_amdgcn_readfirstlane((min(kTile, j1 - j0 - t * kTile) + 3) >> 2);
        unsigned cnt;
#define NB_OPERANDS                                                                                                   \
        : "+v"(ax0), "+v"(ay0), "+v"(az0), "+v"(ax1), "+v"(ay1), "+v"(az1), "+v"(ax2), "+v"(ay2), "+v"(az2), "+v"(ax3),    \
          "+v"(ay3), "+v"(az3), "+v"(lds), [cnt] "=&s"(cnt)                                                             \
        : "v"(x0), "v"(y0), "v"(z0), "v"(x1), "v"(y1), "v"(z1), "v"(x2), "v"(y2), "v"(z2), "v"(x3), "v"(y3), "v"(z3),      \
          "v"(eps2), "v"(tiny), "v"(pinf), [n4] "s"(n4)                                                                 \
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v33", "v34", "v36", "v37", "v38", "v39", "v40", "v41", "v42",   \
          "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "scc", "vcc", "memory"
        if (GUARD)
            asm volatile(NB_TILE_LOOP(NB_GUARD, NB_POST) NB_OPERANDS);
        else if (uniform)
            asm volatile(NB_TILE_LOOP(NB_NOGUARD, NB_POSTU) NB_OPERANDS);
        else
            asm volatile(NB_TILE_LOOP(NB_NOGUARD, NB_POST) NB_OPERANDS);
#undef NB_OPERANDS
        if (t + 1 < ntiles)
            tile[((t + 1) & 1) * kTile + tid] = stage;
        __syncthreads();
    }

    float4 *out = a.partials + (size_t)split * a.row_count;
    const float sc = uniform ? split_mass : 1.f;
    const float4 r0 = make_float4(ax0 * sc, ay0 * sc, az0 * sc, 0.f), r1 = make_float4(ax1 * sc, ay1 * sc, az1 * sc, 0.f);
    const float4 r2 = make_float4(ax2 * sc, ay2 * sc, az2 * sc, 0.f), r3 = make_float4(ax3 * sc, ay3 * sc, az3 * sc, 0.f);
    if (row_base < a.row_count) out[row_base] = r0;
    if (row_base + kTile < a.row_count) out[row_base + kTile] = r1;
    if (row_base + 2 * kTile < a.row_count) out[row_base + 2 * kTile] = r2;
    if (row_base + 3 * kTile < a.row_count) out[row_base + 3 * kTile] = r3;
}


// ---- the same loop with packed fp32 instructions, two rows per instruction (the default 4-rows-per-lane kernel) ------
// Both force kernels run at the package power limit, so what counts is energy per interaction:
// v_pk_{add,mul,fma}_f32 do two rows' worth per issued instruction -- the cycles are the same (4 per packed instruction),
// the results bit-identical, and 26 -> 14 issued instructions per 2 interactions measured 2 % less time (259.6 -> 254.6 ms
// per N = 2^20 pass, same box).  force_kernel_r4 above (rows_per_lane = 40) is kept for the comparison.
// 64-bit operands sit in even-aligned pairs; pair classes ((reg / 2) mod 2) of src0 and src1 differ in every instruction:
//   v[0:3] / v[4:7]  column body: (x,y) class 0, (z,m) class 1       v[8:9] = (eps^2, -) class 0    v10 = 2^-84, v11 = +inf (GUARD)
//   rows 0,1 / 2,3:  X v[14:15] / v[22:23], Y v[18:19] / v[26:27] (class 1), Z v[12:13] / v[16:17] (class 0)
//   temps:           DX,DY,DZ v[30:31],v[34:35],v[38:39] / v[42:43],v[46:47],v[50:51] (class 1), R v[20:21] / v[24:25] (class 0),
//                    Q v[54:55] (class 1)          sums: AX,AY,AZ v[28:29],v[32:33],v[36:37] / v[40:41],v[44:45],v[48:49]
#define PK_NEG " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define PK_PRE(PXY, PZM, X, Y, Z, DX, DY, DZ, R, RLO, RHI, GRD)                                                  \
    "v_pk_add_f32 " DX ", " PXY ", " X " op_sel_hi:[0,1]" PK_NEG                                                    \
    "v_pk_add_f32 " DY ", " PXY ", " Y " op_sel:[1,0] op_sel_hi:[1,1]" PK_NEG                                       \
    "v_pk_add_f32 " DZ ", " PZM ", " Z " op_sel_hi:[0,1]" PK_NEG                                                    \
    "v_pk_fma_f32 " R ", " DX ", " DX ", v[8:9] op_sel_hi:[1,1,0]\n\t"                                              \
    "v_pk_fma_f32 " R ", " DY ", " DY ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DZ ", " DZ ", " R "\n\t" GRD(RLO) GRD(RHI)
#define PK_NOGUARD(R) ""
#define PK_GUARD(R) "v_cmp_le_f32_e32 vcc, v10, " R "\n\tv_cndmask_b32_e32 " R ", v11, " R ", vcc\n\t"
#define PK_POST(PZM, AX, AY, AZ, DX, DY, DZ, R)                                                                  \
    "v_pk_mul_f32 v[54:55], " R ", " R "\n\t"                                                                       \
    "v_pk_mul_f32 " R ", " PZM ", " R " op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                           \
    "v_pk_mul_f32 " R ", " R ", v[54:55]\n\t"                                                                       \
    "v_pk_fma_f32 " AX ", " DX ", " R ", " AX "\n\t"                                                                 \
    "v_pk_fma_f32 " AY ", " DY ", " R ", " AY "\n\t"                                                                 \
    "v_pk_fma_f32 " AZ ", " DZ ", " R ", " AZ "\n\t"
#define PK_POSTU(PZM, AX, AY, AZ, DX, DY, DZ, R) /* a split of equal masses: see force_kernel */                   \
    "v_pk_mul_f32 v[54:55], " R ", " R "\n\t"                                                                       \
    "v_pk_mul_f32 " R ", " R ", v[54:55]\n\t"                                                                       \
    "v_pk_fma_f32 " AX ", " DX ", " R ", " AX "\n\t"                                                                 \
    "v_pk_fma_f32 " AY ", " DY ", " R ", " AY "\n\t"                                                                 \
    "v_pk_fma_f32 " AZ ", " DZ ", " R ", " AZ "\n\t"
#define PK_COLUMN(PXY, PZM, NEXT, GRD, POST)                                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    PK_PRE(PXY, PZM, "v[14:15]", "v[18:19]", "v[12:13]", "v[30:31]", "v[34:35]", "v[38:39]", "v[20:21]", "v20", "v21", GRD) \
    PK_PRE(PXY, PZM, "v[22:23]", "v[26:27]", "v[16:17]", "v[42:43]", "v[46:47]", "v[50:51]", "v[24:25]", "v24", "v25", GRD) \
    "v_rsq_f32_e32 v20, v20\n\tv_rsq_f32_e32 v21, v21\n\tv_rsq_f32_e32 v24, v24\n\tv_rsq_f32_e32 v25, v25\n\t"       \
    NEXT NB_R4_GAP NB_R4_PRIO_POST                                                                               \
    POST(PZM, "v[28:29]", "v[32:33]", "v[36:37]", "v[30:31]", "v[34:35]", "v[38:39]", "v[20:21]")             \
    POST(PZM, "v[40:41]", "v[44:45]", "v[48:49]", "v[42:43]", "v[46:47]", "v[50:51]", "v[24:25]")             \
    NB_R4_PRIO_PRE
#define PK_TILE_LOOP(GRD, POST)                                                                                        \
    NB_R4_PRIO_PRE                                                                                               \
    "ds_read_b128 v[0:3], v52\n\t"                                                                               \
    "s_mov_b32 %[cnt], %[n4]\n"                                                                                     \
    "1:\n\t"                                                                                                     \
    PK_COLUMN("v[0:1]", "v[2:3]", "ds_read_b128 v[4:7], v52 offset:16\n\t", GRD, POST)                                 \
    PK_COLUMN("v[4:5]", "v[6:7]", "ds_read_b128 v[0:3], v52 offset:32\n\t", GRD, POST)                                 \
    PK_COLUMN("v[0:1]", "v[2:3]", "ds_read_b128 v[4:7], v52 offset:48\n\t", GRD, POST)                                 \
    PK_COLUMN("v[4:5]", "v[6:7]", "v_add_u32_e32 v52, 64, v52\n\tds_read_b128 v[0:3], v52\n\t", GRD, POST)              \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// The same loop under PER-PARTICLE SOFTENING (eps_ij^2 = eps^2 + eps_i^2 + eps_j^2): the first addend of the r^2
// chain is formed per interaction -- (eps^2 + eps_i^2) of the two rows (ER01 = v[58:59], ER23 = v[62:63], class 1) plus the
// column's eps_j^2, read two columns at a time from a second, scalar LDS tile (v[56:57] / v[64:65], class 0; address v66) --
// the sum force_kernel<RPL, GUARD, true> forms, so not a bit differs.  6 (equal-mass splits) or 6.5 packed instructions + 1
// transcendental per interaction instead of 5.5 / 6.
#define PK_ELO " op_sel:[0,0] op_sel_hi:[1,0]\n\t"
#define PK_EHI " op_sel:[0,1] op_sel_hi:[1,1]\n\t"
#define PK_PRE_E(PXY, PZM, X, Y, Z, DX, DY, DZ, R, RLO, RHI, GRD, ER, EJ, ESEL)                                  \
    "v_pk_add_f32 " DX ", " PXY ", " X " op_sel_hi:[0,1]" PK_NEG                                                    \
    "v_pk_add_f32 " DY ", " PXY ", " Y " op_sel:[1,0] op_sel_hi:[1,1]" PK_NEG                                       \
    "v_pk_add_f32 " DZ ", " PZM ", " Z " op_sel_hi:[0,1]" PK_NEG                                                    \
    "v_pk_add_f32 " R ", " ER ", " EJ ESEL                                                                          \
    "v_pk_fma_f32 " R ", " DX ", " DX ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DY ", " DY ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DZ ", " DZ ", " R "\n\t" GRD(RLO) GRD(RHI)
#define PK_COLUMN_E(PXY, PZM, NEXT, GRD, POST, EJ, ESEL)                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    PK_PRE_E(PXY, PZM, "v[14:15]", "v[18:19]", "v[12:13]", "v[30:31]", "v[34:35]", "v[38:39]", "v[20:21]", "v20", "v21", GRD, \
             "v[58:59]", EJ, ESEL)                                                                               \
    PK_PRE_E(PXY, PZM, "v[22:23]", "v[26:27]", "v[16:17]", "v[42:43]", "v[46:47]", "v[50:51]", "v[24:25]", "v24", "v25", GRD, \
             "v[62:63]", EJ, ESEL)                                                                               \
    "v_rsq_f32_e32 v20, v20\n\tv_rsq_f32_e32 v21, v21\n\tv_rsq_f32_e32 v24, v24\n\tv_rsq_f32_e32 v25, v25\n\t"       \
    NEXT NB_R4_GAP NB_R4_PRIO_POST                                                                               \
    POST(PZM, "v[28:29]", "v[32:33]", "v[36:37]", "v[30:31]", "v[34:35]", "v[38:39]", "v[20:21]")             \
    POST(PZM, "v[40:41]", "v[44:45]", "v[48:49]", "v[42:43]", "v[46:47]", "v[50:51]", "v[24:25]")             \
    NB_R4_PRIO_PRE
#define PK_TILE_LOOP_E(GRD, POST)                                                                                \
    NB_R4_PRIO_PRE                                                                                               \
    "ds_read_b128 v[0:3], v52\n\t"                                                                               \
    "ds_read2_b32 v[56:57], v66 offset1:1\n\t"                                                                   \
    "s_mov_b32 %[cnt], %[n4]\n"                                                                                     \
    "1:\n\t"                                                                                                     \
    PK_COLUMN_E("v[0:1]", "v[2:3]", "ds_read_b128 v[4:7], v52 offset:16\n\tds_read2_b32 v[64:65], v66 offset0:2 offset1:3\n\t", \
                GRD, POST, "v[56:57]", PK_ELO)                                                                   \
    PK_COLUMN_E("v[4:5]", "v[6:7]", "ds_read_b128 v[0:3], v52 offset:32\n\t", GRD, POST, "v[56:57]", PK_EHI)       \
    PK_COLUMN_E("v[0:1]", "v[2:3]", "ds_read_b128 v[4:7], v52 offset:48\n\tds_read2_b32 v[56:57], v66 offset0:4 offset1:5\n\t", \
                GRD, POST, "v[64:65]", PK_ELO)                                                                   \
    PK_COLUMN_E("v[4:5]", "v[6:7]", "v_add_u32_e32 v52, 64, v52\n\tv_add_u32_e32 v66, 16, v66\n\tds_read_b128 v[0:3], v52\n\t", \
                GRD, POST, "v[64:65]", PK_EHI)                                                                   \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// operands of PK_TILE_LOOP_E in both packed kernels
#define PK_OPERANDS_E                                                                                                 \
        : "+{v[28:29]}"(ax01), "+{v[32:33]}"(ay01), "+{v[36:37]}"(az01), "+{v[40:41]}"(ax23), "+{v[44:45]}"(ay23),      \
          "+{v[48:49]}"(az23), "+{v52}"(lds), "+{v66}"(elds), [cnt] "=&s"(cnt)                                        \
        : "{v[14:15]}"(x01), "{v[18:19]}"(y01), "{v[12:13]}"(z01), "{v[22:23]}"(x23), "{v[26:27]}"(y23),              \
          "{v[16:17]}"(z23), "{v[58:59]}"(er01), "{v[62:63]}"(er23), "{v10}"(tiny), "{v11}"(pinf), [n4] "s"(n4)      \
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v20", "v21", "v24", "v25", "v30", "v31", "v34", "v35",     \
          "v38", "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v56", "v57", "v64", "v65", "scc",    \
          "vcc", "memory"

typedef float nb_f2 __attribute__((ext_vector_type(2)));

template <bool GUARD, bool PPS = false>
__global__ __launch_bounds__(kTile) void force_kernel_r4pk(ForceArgs a)
{
    __shared__ float4 tile[2 * kTile + 1];
    __shared__ float etile[PPS ? 2 * kTile + 2 : 1];  // eps_j^2 of the tile's columns (+ 2: the loop's last read-ahead)

    const int tid = threadIdx.x;
    int split = a.split_first + blockIdx.y;
    if (split >= a.skip_first)
        split += a.skip_count;
    const int j0 = split * a.split_len;
    const int j1 = min(j0 + a.split_len, a.n_total);
    const int ntiles = (j1 - j0 + kTile - 1) / kTile;
    // Which row of its wave's 64 a lane takes: 4 (lane mod 16) + lane / 16, so that the four lanes one ALU lane serves in
    // consecutive cycles hold four consecutive bodies (bodies along a space-filling curve: fewer operand bits toggle, see
    // nbody_symmetric.hip).  A row's sum does not depend on the lane that forms it: not a bit changes.
    const int row_base = blockIdx.x * (kTile * 4) + (tid & ~63) + 4 * (tid & 15) + ((tid >> 4) & 3);
    const float split_mass = GUARD ? __builtin_nanf("") : a.split_mass[split];  // see force_kernel
    const bool uniform = split_mass == split_mass;

    float4 p[4];
    float er[4];  // PPS: eps^2 + eps_i^2 of the rows
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = row_base + k * kTile;
        p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        float e = 0.f;
        if (r < a.row_count) {
            p[k] = a.pos[a.row_lo + r];
            if (PPS)
                e = a.eps_pp[a.row_lo + r];
        }
        er[k] = __builtin_fmaf(e, e, a.eps2);
    }
    const nb_f2 x01 = {p[0].x, p[1].x}, y01 = {p[0].y, p[1].y}, z01 = {p[0].z, p[1].z};
    const nb_f2 x23 = {p[2].x, p[3].x}, y23 = {p[2].y, p[3].y}, z23 = {p[2].z, p[3].z};
    const nb_f2 er01 = {er[0], er[1]}, er23 = {er[2], er[3]};
    nb_f2 ax01 = {0.f, 0.f}, ay01 = ax01, az01 = ax01, ax23 = ax01, ay23 = ax01, az23 = ax01;
    const nb_f2 epsv = {a.eps2, 0.f};
    const float tiny = kGuardMin, pinf = __builtin_inff();

    float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
    float stage_e = 0.f;
    if (j0 + tid < j1) {
        stage = a.pos[j0 + tid];
        if (PPS)
            stage_e = a.eps_pp[j0 + tid];
    }
    tile[tid] = stage;
    if (PPS)
        etile[tid] = stage_e * stage_e;
    if (tid == 0)
        tile[2 * kTile] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PPS && tid < 2)
        etile[2 * kTile + tid] = 0.f;
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int jn = j0 + (t + 1) * kTile + tid;
        if (t + 1 < ntiles) {
            stage = make_float4(0.f, 0.f, 0.f, 0.f);
            stage_e = 0.f;
            if (jn < j1) {
                stage = a.pos[jn];
                if (PPS)
                    stage_e = a.eps_pp[jn];
            }
        }
        unsigned lds = (unsigned)(size_t)(&tile[(t & 1) * kTile]);
        // four-column iterations of this tile: all 64 but for the last tile of a split whose length is no multiple of the
        // 256-body tile (split lengths are multiples of 64; a ragged last split's padding columns are zero-mass bodies)
        const unsigned n4 = (unsigned)__builtin_amdgcn_readfirstlane((min(kTile, j1 - j0 - t * kTile) + 3) >> 2);
        unsigned cnt;
        if constexpr (PPS) {
            unsigned elds = (unsigned)(size_t)(&etile[(t & 1) * kTile]);
            if (GUARD)
                asm volatile(PK_TILE_LOOP_E(PK_GUARD, PK_POST) PK_OPERANDS_E);
            else if (uniform)
                asm volatile(PK_TILE_LOOP_E(PK_NOGUARD, PK_POSTU) PK_OPERANDS_E);
            else
                asm volatile(PK_TILE_LOOP_E(PK_NOGUARD, PK_POST) PK_OPERANDS_E);
            if (t + 1 < ntiles) {
                tile[((t + 1) & 1) * kTile + tid] = stage;
                etile[((t + 1) & 1) * kTile + tid] = stage_e * stage_e;
            }
            __syncthreads();
            continue;
        }
#define PK_OPERANDS                                                                                                   \
        : "+{v[28:29]}"(ax01), "+{v[32:33]}"(ay01), "+{v[36:37]}"(az01), "+{v[40:41]}"(ax23), "+{v[44:45]}"(ay23),          \
          "+{v[48:49]}"(az23), "+{v52}"(lds), [cnt] "=&s"(cnt)                                                            \
        : "{v[14:15]}"(x01), "{v[18:19]}"(y01), "{v[12:13]}"(z01), "{v[22:23]}"(x23), "{v[26:27]}"(y23), "{v[16:17]}"(z23),  \
          "{v[8:9]}"(epsv), "{v10}"(tiny), "{v11}"(pinf), [n4] "s"(n4)                                                                  \
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v20", "v21", "v24", "v25", "v30", "v31", "v34", "v35", "v38",   \
          "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "scc", "vcc", "memory"
        if (GUARD)
            asm volatile(PK_TILE_LOOP(PK_GUARD, PK_POST) PK_OPERANDS);
        else if (uniform)
            asm volatile(PK_TILE_LOOP(PK_NOGUARD, PK_POSTU) PK_OPERANDS);
        else
            asm volatile(PK_TILE_LOOP(PK_NOGUARD, PK_POST) PK_OPERANDS);
#undef PK_OPERANDS
        if (t + 1 < ntiles)
            tile[((t + 1) & 1) * kTile + tid] = stage;
        __syncthreads();
    }

    float4 *out = a.partials + (size_t)split * a.row_count;
    const float sc = uniform ? split_mass : 1.f;
    if (row_base < a.row_count) out[row_base] = make_float4(ax01.x * sc, ay01.x * sc, az01.x * sc, 0.f);
    if (row_base + kTile < a.row_count) out[row_base + kTile] = make_float4(ax01.y * sc, ay01.y * sc, az01.y * sc, 0.f);
    if (row_base + 2 * kTile < a.row_count) out[row_base + 2 * kTile] = make_float4(ax23.x * sc, ay23.x * sc, az23.x * sc, 0.f);
    if (row_base + 3 * kTile < a.row_count) out[row_base + 3 * kTile] = make_float4(ax23.y * sc, ay23.y * sc, az23.y * sc, 0.f);
}

// ---- the same packed loop with ONE wave per workgroup: systems too small to fill the chip with 1024-row workgroups ------
// A workgroup of the kernel above is 4 waves x 4 rows x 64 lanes = 1024 rows: at the reference's own size (20 225 bodies
// padded, 79 splits of 256 columns) that is 20 x 79 = 1580 workgroups, six or seven to a CU, and the compiler-allocated
// one-row kernel that filled the chip better ran at 3.3e12 interactions/s (124 us per pass, profiles/r03_small_n_kernels.txt)
// against the 4.8e12 this loop reaches on large systems.  Here a workgroup is one wave: 256 rows x one split, 80 x 79 = 6320
// waves, six to a SIMD, dealt as slots free up; the wave stages its own 256-column tiles (four float4 a lane, no barrier
// partner) and runs the identical hand-allocated loop, so each row's sum is the same FMA chain: not a bit changes.
// Splits of up to 512 columns (every system below 32 768 bodies) need no split_mass_kernel launch in front: the wave holds
// the split's masses in registers and forms the same flag itself (one launch less per step).
template <bool GUARD, int QT, bool PPS = false>
__global__ __launch_bounds__(64) void force_kernel_r4pk_w1(ForceArgs a)
{
    // QT != 0: the WHOLE split -- at most 64 QT columns, QT = 4 ... 8 -- is staged at once: no second buffer, no columns in
    // flight under the loop.  4 KiB of LDS for 256-column splits, 5 KiB for the 320-column splits of the reference's own size:
    // with the double-buffered 8 KiB only 19 of the 20 one-wave workgroups a CU gets there were resident (4.75 waves per SIMD:
    // a sixth round, measured 118 us per pass against 110 with 256-column splits).  QT = 0: longer splits, two 256-column
    // buffers used in turn.
    constexpr bool WHOLE = QT != 0;
    constexpr int kStageCols = WHOLE ? 64 * QT : 2 * kTile, NS = WHOLE ? QT : 4;
    __shared__ float4 tile[kStageCols + 1];
    __shared__ float etile[PPS ? kStageCols + 2 : 1];  // per-particle softening: eps_j^2 of the columns

    const int lane = threadIdx.x;
    int split = a.split_first + blockIdx.y;
    if (split >= a.skip_first)
        split += a.skip_count;
    const int j0 = split * a.split_len;
    const int j1 = min(j0 + a.split_len, a.n_total);
    const int ntiles = (j1 - j0 + kTile - 1) / kTile;
    const int row_base = blockIdx.x * kTile + 4 * (lane & 15) + (lane >> 4);  // rows row_base + 64 k: see force_kernel_r4pk

    float4 p[4];
    float er[4];  // PPS: eps^2 + eps_i^2 of the rows
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = row_base + k * 64;
        p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        float e = 0.f;
        if (r < a.row_count) {
            p[k] = a.pos[a.row_lo + r];
            if (PPS)
                e = a.eps_pp[a.row_lo + r];
        }
        er[k] = __builtin_fmaf(e, e, a.eps2);
    }
    const nb_f2 x01 = {p[0].x, p[1].x}, y01 = {p[0].y, p[1].y}, z01 = {p[0].z, p[1].z};
    const nb_f2 x23 = {p[2].x, p[3].x}, y23 = {p[2].y, p[3].y}, z23 = {p[2].z, p[3].z};
    const nb_f2 er01 = {er[0], er[1]}, er23 = {er[2], er[3]};
    nb_f2 ax01 = {0.f, 0.f}, ay01 = ax01, az01 = ax01, ax23 = ax01, ay23 = ax01, az23 = ax01;
    const nb_f2 epsv = {a.eps2, 0.f};
    const float tiny = kGuardMin, pinf = __builtin_inff();

    float4 stage[NS];  // NS columns per lane (the split, or its first tile); an out-of-range column is a zero-mass body at the origin
    float stage_e[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int c = j0 + k * 64 + lane;
        stage[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        stage_e[k] = 0.f;
        if (c < j1) {
            stage[k] = a.pos[c];
            if (PPS)
                stage_e[k] = a.eps_pp[c];
        }
    }
    // the one mass of the split's bodies, as split_mass_kernel defines it (nbody_symmetric.hip): from the tile in registers
    // when the split is this one tile, else from the flags the launch in front has written
    float split_mass = __builtin_nanf("");
    if (!GUARD) {
        if (a.own_split_mass) {
            const unsigned m0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, stage[0].w));
            bool differs = false;
#pragma unroll
            for (int k = 0; k < NS; ++k)
                differs |= __builtin_bit_cast(unsigned, stage[k].w) != m0;
            if (!WHOLE)  // the columns beyond the first tile: their masses straight from memory
                for (int c = kTile + lane; c < a.split_len; c += 64) {
                    const float m = j0 + c < a.n_total ? a.pos[j0 + c].w : 0.f;  // a ragged split's missing bodies: mass 0
                    differs |= __builtin_bit_cast(unsigned, m) != m0;
                }
            const float mf = __builtin_bit_cast(float, m0);
            if (!__builtin_amdgcn_ballot_w64(differs) && fabsf(mf) <= 3.4e38f)
                split_mass = mf;
        } else {
            split_mass = a.split_mass[split];
        }
    }
    const bool uniform = split_mass == split_mass;

#pragma unroll
    for (int k = 0; k < NS; ++k) {
        tile[k * 64 + lane] = stage[k];
        if (PPS)
            etile[k * 64 + lane] = stage_e[k] * stage_e[k];
    }
    if (lane == 0)
        tile[kStageCols] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PPS && lane < 2)
        etile[kStageCols + lane] = 0.f;
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        if (!WHOLE && t + 1 < ntiles) {  // in flight under the tile's arithmetic
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = j0 + (t + 1) * kTile + k * 64 + lane;
                stage[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                stage_e[k] = 0.f;
                if (c < j1) {
                    stage[k] = a.pos[c];
                    if (PPS)
                        stage_e[k] = a.eps_pp[c];
                }
            }
        }
        unsigned lds = (unsigned)(size_t)(&tile[WHOLE ? t * kTile : (t & 1) * kTile]);
        // four-column iterations of this tile: all 64 but for the last tile of a split whose length is no multiple of the
        // 256-body tile (split lengths are multiples of 64; a ragged last split's padding columns are zero-mass bodies)
        const unsigned n4 = (unsigned)__builtin_amdgcn_readfirstlane((min(kTile, j1 - j0 - t * kTile) + 3) >> 2);
        unsigned cnt;
        if constexpr (PPS) {
            unsigned elds = (unsigned)(size_t)(&etile[WHOLE ? t * kTile : (t & 1) * kTile]);
            if (GUARD)
                asm volatile(PK_TILE_LOOP_E(PK_GUARD, PK_POST) PK_OPERANDS_E);
            else if (uniform)
                asm volatile(PK_TILE_LOOP_E(PK_NOGUARD, PK_POSTU) PK_OPERANDS_E);
            else
                asm volatile(PK_TILE_LOOP_E(PK_NOGUARD, PK_POST) PK_OPERANDS_E);
            if (!WHOLE && t + 1 < ntiles) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tile[((t + 1) & 1) * kTile + k * 64 + lane] = stage[k];
                    etile[((t + 1) & 1) * kTile + k * 64 + lane] = stage_e[k] * stage_e[k];
                }
            }
            if (!WHOLE)
                __syncthreads();
            continue;
        }
#define PK_OPERANDS                                                                                                   \
        : "+{v[28:29]}"(ax01), "+{v[32:33]}"(ay01), "+{v[36:37]}"(az01), "+{v[40:41]}"(ax23), "+{v[44:45]}"(ay23),          \
          "+{v[48:49]}"(az23), "+{v52}"(lds), [cnt] "=&s"(cnt)                                                            \
        : "{v[14:15]}"(x01), "{v[18:19]}"(y01), "{v[12:13]}"(z01), "{v[22:23]}"(x23), "{v[26:27]}"(y23), "{v[16:17]}"(z23),  \
          "{v[8:9]}"(epsv), "{v10}"(tiny), "{v11}"(pinf), [n4] "s"(n4)                                                                  \
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v20", "v21", "v24", "v25", "v30", "v31", "v34", "v35", "v38",   \
          "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "scc", "vcc", "memory"
        if (GUARD)
            asm volatile(PK_TILE_LOOP(PK_GUARD, PK_POST) PK_OPERANDS);
        else if (uniform)
            asm volatile(PK_TILE_LOOP(PK_NOGUARD, PK_POSTU) PK_OPERANDS);
        else
            asm volatile(PK_TILE_LOOP(PK_NOGUARD, PK_POST) PK_OPERANDS);
#undef PK_OPERANDS
        if (!WHOLE && t + 1 < ntiles) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tile[((t + 1) & 1) * kTile + k * 64 + lane] = stage[k];
        }
        if (!WHOLE)
            __syncthreads();
    }

    float4 *out = a.partials + (size_t)split * a.row_count;
    const float sc = uniform ? split_mass : 1.f;
    if (row_base < a.row_count) out[row_base] = make_float4(ax01.x * sc, ay01.x * sc, az01.x * sc, 0.f);
    if (row_base + 64 < a.row_count) out[row_base + 64] = make_float4(ax01.y * sc, ay01.y * sc, az01.y * sc, 0.f);
    if (row_base + 128 < a.row_count) out[row_base + 128] = make_float4(ax23.x * sc, ay23.x * sc, az23.x * sc, 0.f);
    if (row_base + 192 < a.row_count) out[row_base + 192] = make_float4(ax23.y * sc, ay23.y * sc, az23.y * sc, 0.f);
}

template <int QT>
static void launch_w1(const ForceArgs &a, dim3 grid, hipStream_t stream)
{
    if (a.eps_pp) {  // per-particle softening (a particle may have eps = 0: the guard stays on when eps = 0)
        if (a.eps2 > 0.f)
            hipLaunchKernelGGL((force_kernel_r4pk_w1<false, QT, true>), grid, dim3(64), 0, stream, a);
        else
            hipLaunchKernelGGL((force_kernel_r4pk_w1<true, QT, true>), grid, dim3(64), 0, stream, a);
    } else if (a.eps2 > 0.f) {
        hipLaunchKernelGGL((force_kernel_r4pk_w1<false, QT>), grid, dim3(64), 0, stream, a);
    } else {
        hipLaunchKernelGGL((force_kernel_r4pk_w1<true, QT>), grid, dim3(64), 0, stream, a);
    }
}

static hipError_t launch_forces_r4pk_w1(const ForceArgs &a, hipStream_t stream)
{
    dim3 grid((a.row_count + kTile - 1) / kTile, a.split_count, 1);
    switch (a.split_len <= 2 * kTile ? (a.split_len + 63) / 64 : 0) {  // splits of up to 512 columns are staged whole
    case 0: launch_w1<0>(a, grid, stream); break;
    case 5: launch_w1<5>(a, grid, stream); break;
    case 6: launch_w1<6>(a, grid, stream); break;
    case 7: launch_w1<7>(a, grid, stream); break;
    case 8: launch_w1<8>(a, grid, stream); break;
    default: launch_w1<4>(a, grid, stream); break;  // 256 columns or fewer
    }
    return hipGetLastError();
}

static hipError_t launch_forces_r4pk(const ForceArgs &a, hipStream_t stream)
{
    dim3 grid((a.row_count + kTile * 4 - 1) / (kTile * 4), a.split_count, 1);
    if (a.eps_pp) {
        if (a.eps2 > 0.f)
            hipLaunchKernelGGL((force_kernel_r4pk<false, true>), grid, dim3(kTile), 0, stream, a);
        else
            hipLaunchKernelGGL((force_kernel_r4pk<true, true>), grid, dim3(kTile), 0, stream, a);
    } else if (a.eps2 > 0.f) {
        hipLaunchKernelGGL((force_kernel_r4pk<false, false>), grid, dim3(kTile), 0, stream, a);
    } else {
        hipLaunchKernelGGL((force_kernel_r4pk<true, false>), grid, dim3(kTile), 0, stream, a);
    }
    return hipGetLastError();
}

template <int RPL>
static hipError_t launch_forces_rpl(const ForceArgs &a, hipStream_t stream)
{
    const int rows_per_block = kTile * RPL;
    dim3 grid((a.row_count + rows_per_block - 1) / rows_per_block, a.split_count, 1);
    if (a.eps_pp) {  // per-particle softening: a particle may have eps = 0, so the guard stays on when eps = 0
        if (a.eps2 > 0.f)
            hipLaunchKernelGGL((force_kernel<RPL, false, true>), grid, dim3(kTile), 0, stream, a);
        else
            hipLaunchKernelGGL((force_kernel<RPL, true, true>), grid, dim3(kTile), 0, stream, a);
        return hipGetLastError();
    }
    if (a.eps2 > 0.f)
        hipLaunchKernelGGL((force_kernel<RPL, false>), grid, dim3(kTile), 0, stream, a);
    else
        hipLaunchKernelGGL((force_kernel<RPL, true>), grid, dim3(kTile), 0, stream, a);
    return hipGetLastError();
}

static hipError_t launch_forces_r4_asm(const ForceArgs &a, hipStream_t stream)
{
    dim3 grid((a.row_count + kTile * 4 - 1) / (kTile * 4), a.split_count, 1);
    if (a.eps2 > 0.f)
        hipLaunchKernelGGL(force_kernel_r4<false>, grid, dim3(kTile), 0, stream, a);
    else
        hipLaunchKernelGGL(force_kernel_r4<true>, grid, dim3(kTile), 0, stream, a);
    return hipGetLastError();
}

// rows_per_lane: 4 = the hand-allocated kernel (default), 41 = the same with one wave per workgroup (small systems), 1/2/8 and
// -4 = the compiler-allocated template
hipError_t launch_forces(const ForceArgs &a, int rows_per_lane, hipStream_t stream)
{
    if (a.row_count <= 0 || a.split_count <= 0)
        return hipSuccess;
    switch (rows_per_lane) {
    case 1: return launch_forces_rpl<1>(a, stream);
    case 2: return launch_forces_rpl<2>(a, stream);
    case 4: return launch_forces_r4pk(a, stream);     // packed fp32 (default); per-particle softening: its own loop
    case 40: return a.eps_pp ? launch_forces_rpl<4>(a, stream) : launch_forces_r4_asm(a, stream);  // one row per instruction
    case 41: return launch_forces_r4pk_w1(a, stream);  // packed, one wave per workgroup
    case -4: return launch_forces_rpl<4>(a, stream);
    case 8: return launch_forces_rpl<8>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

// The splits' partial sums of row r, added in ascending split order (one fp32 chain: the order defines the bits).  The
// loads of 16 splits are issued together and the adds follow in order: a lane's loads are independent, its adds are not,
// and with few rows (the reference's N = 20 000: 79 splits x 20 225 rows, 80 workgroups of 256) the kernel is bound by
// load latency, not bandwidth -- 21.7 us with one load in flight per lane, profiles/r03_small_n_kernels.txt.  (32 at a time
// measured slower: 13.7 against 11.0 us at N = 20 225.)
template <int BATCH = 16>
__device__ __forceinline__ float4 sum_partials(const float4 *partials, int r, int row_count, int n_splits)
{
    float4 acc = partials[r];
    int s = 1;
    for (; s + BATCH <= n_splits; s += BATCH) {
        float4 p[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            p[k] = partials[(size_t)(s + k) * row_count + r];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            acc.x += p[k].x;
            acc.y += p[k].y;
            acc.z += p[k].z;
        }
    }
    for (; s < n_splits; ++s) {
        const float4 p = partials[(size_t)s * row_count + r];
        acc.x += p.x;
        acc.y += p.y;
        acc.z += p.z;
    }
    return acc;
}

// use_acc_update_position, kernel.cu:777-801, with the reference's fp64 FMA (TIME_TICK is a double
// literal there) and the partial sums of the splits added first, in ascending split order.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void update_kernel(float4 *pos_all, float4 *vel_rows, const float4 *partials,
                                                       int row_lo, int row_count, int n_splits, float dt)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= row_count)
        return;
    const float4 acc = sum_partials(partials, r, row_count, n_splits);
    float4 v = vel_rows[r];
    float4 x = pos_all[row_lo + r];
    const double h = (double)dt;
    v.x = (float)__builtin_fma((double)acc.x, h, (double)v.x);
    v.y = (float)__builtin_fma((double)acc.y, h, (double)v.y);
    v.z = (float)__builtin_fma((double)acc.z, h, (double)v.z);
    x.x = (float)__builtin_fma((double)v.x, h, (double)x.x);
    x.y = (float)__builtin_fma((double)v.y, h, (double)x.y);
    x.z = (float)__builtin_fma((double)v.z, h, (double)x.z);
    vel_rows[r] = v;       // .w (the unused per-particle eps) written back unchanged
    pos_all[row_lo + r] = x;  // .w (mass) written back unchanged
}

hipError_t launch_update(float4 *pos_all, float4 *vel_rows, const float4 *partials, int row_lo, int row_count,
                         int n_splits, float dt, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    // few rows: one wave per workgroup, so that every CU gets some (20 225 rows: 317 workgroups instead of 80)
    if (row_count < 256 * kTile)
        hipLaunchKernelGGL(update_kernel<64>, dim3((row_count + 63) / 64), dim3(64), 0, stream, pos_all, vel_rows, partials,
                           row_lo, row_count, n_splits, dt);
    else
        hipLaunchKernelGGL(update_kernel<kTile>, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, pos_all,
                           vel_rows, partials, row_lo, row_count, n_splits, dt);
    return hipGetLastError();
}

// The pair-once mode's last two passes in one: acc[b] = sum over the groups g (ascending) of ( rowsum[g][b] + colparts[g][b] )
// -- sym_combine_kernel's association (nbody_symmetric.hip), so not a bit changes -- and the kick-drift of update_kernel,
// without the round trip of the summed accelerations through memory and one launch less behind the force pass.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void update_sym_kernel(float4 *pos_all, float4 *vel_rows, const float4 *rowsum,
                                                           const float4 *colparts, int row_lo, int row_count, int n_total,
                                                           int n_groups, float dt)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= row_count)
        return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int g = 0; g < n_groups; ++g) {
        const float4 rs = rowsum[(size_t)g * row_count + r];
        const float4 cp = colparts[(size_t)g * n_total + row_lo + r];
        ax += rs.x + cp.x;
        ay += rs.y + cp.y;
        az += rs.z + cp.z;
    }
    float4 v = vel_rows[r];
    float4 x = pos_all[row_lo + r];
    const double h = (double)dt;
    v.x = (float)__builtin_fma((double)ax, h, (double)v.x);
    v.y = (float)__builtin_fma((double)ay, h, (double)v.y);
    v.z = (float)__builtin_fma((double)az, h, (double)v.z);
    x.x = (float)__builtin_fma((double)v.x, h, (double)x.x);
    x.y = (float)__builtin_fma((double)v.y, h, (double)x.y);
    x.z = (float)__builtin_fma((double)v.z, h, (double)x.z);
    vel_rows[r] = v;
    pos_all[row_lo + r] = x;
}

hipError_t launch_update_sym(float4 *pos_all, float4 *vel_rows, const float4 *rowsum, const float4 *colparts, int row_lo,
                             int row_count, int n_total, int n_groups, float dt, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    if (row_count < 256 * kTile)
        hipLaunchKernelGGL(update_sym_kernel<64>, dim3((row_count + 63) / 64), dim3(64), 0, stream, pos_all, vel_rows, rowsum,
                           colparts, row_lo, row_count, n_total, n_groups, dt);
    else
        hipLaunchKernelGGL(update_sym_kernel<kTile>, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, pos_all,
                           vel_rows, rowsum, colparts, row_lo, row_count, n_total, n_groups, dt);
    return hipGetLastError();
}

// mode 0: acc = sum ; mode 1: acc = sum, v += acc*dt/2
template <int MODE>
__global__ __launch_bounds__(kTile) void kdk_kick_kernel(float4 *vel_rows, float4 *acc_out, const float4 *partials,
                                                         int row_count, int n_splits, float dt)
{
    const int r = blockIdx.x * kTile + threadIdx.x;
    if (r >= row_count)
        return;
    const float4 acc = sum_partials(partials, r, row_count, n_splits);
    acc_out[r] = acc;
    if (MODE == 1) {
        float4 v = vel_rows[r];
        const double h = 0.5 * (double)dt;
        v.x = (float)__builtin_fma((double)acc.x, h, (double)v.x);
        v.y = (float)__builtin_fma((double)acc.y, h, (double)v.y);
        v.z = (float)__builtin_fma((double)acc.z, h, (double)v.z);
        vel_rows[r] = v;
    }
}

__global__ __launch_bounds__(kTile) void kdk_kick_drift_kernel(float4 *pos_all, float4 *vel_rows, const float4 *acc_in,
                                                               int row_lo, int row_count, float dt)
{
    const int r = blockIdx.x * kTile + threadIdx.x;
    if (r >= row_count)
        return;
    const float4 acc = acc_in[r];
    float4 v = vel_rows[r];
    float4 x = pos_all[row_lo + r];
    const double h = (double)dt, hh = 0.5 * (double)dt;
    v.x = (float)__builtin_fma((double)acc.x, hh, (double)v.x);
    v.y = (float)__builtin_fma((double)acc.y, hh, (double)v.y);
    v.z = (float)__builtin_fma((double)acc.z, hh, (double)v.z);
    x.x = (float)__builtin_fma((double)v.x, h, (double)x.x);
    x.y = (float)__builtin_fma((double)v.y, h, (double)x.y);
    x.z = (float)__builtin_fma((double)v.z, h, (double)x.z);
    vel_rows[r] = v;
    pos_all[row_lo + r] = x;
}

hipError_t launch_kdk_reduce(float4 *acc, const float4 *partials, int row_count, int n_splits, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(kdk_kick_kernel<0>, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, nullptr, acc,
                       partials, row_count, n_splits, 0.f);
    return hipGetLastError();
}

hipError_t launch_kdk_kick(float4 *vel_rows, float4 *acc, const float4 *partials, int row_count, int n_splits, float dt,
                           hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(kdk_kick_kernel<1>, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, vel_rows, acc,
                       partials, row_count, n_splits, dt);
    return hipGetLastError();
}

hipError_t launch_kdk_kick_drift(float4 *pos_all, float4 *vel_rows, const float4 *acc, int row_lo, int row_count, float dt,
                                 hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(kdk_kick_drift_kernel, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, pos_all,
                       vel_rows, acc, row_lo, row_count, dt);
    return hipGetLastError();
}

__global__ __launch_bounds__(kTile) void scatter_mass_kernel(float4 *pos_all, const float *masses, int n_total)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i < n_total)
        reinterpret_cast<float *>(pos_all)[4 * (size_t)i + 3] = masses[i];
}

hipError_t launch_scatter_mass(float4 *pos_all, const float *masses, int n_total, hipStream_t stream)
{
    if (n_total <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(scatter_mass_kernel, dim3((n_total + kTile - 1) / kTile), dim3(kTile), 0, stream, pos_all,
                       masses, n_total);
    return hipGetLastError();
}

// ---- diagnostics ---------------------------------------------------------------------------------

template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double *block_out)
{
    __shared__ double red[NV][kTile / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        double x = v[c];
        for (int off = 32; off > 0; off >>= 1)
            x += __shfl_down(x, off, 64);
        if (lane == 0)
            red[c][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            double x = 0;
            for (int w = 0; w < kTile / 64; ++w)
                x += red[c][w];
            block_out[(size_t)blockIdx.x * NV + c] = x;
        }
    }
}

// Potential of each row against ALL columns (self pair excluded by index), fp32 pair terms, fp32 sum inside a tile,
// fp64 across tiles and across rows; kinetic energy of the rows.  Four rows per lane feed on each broadcast LDS read;
// only the (at most four) tiles that hold a workgroup's own rows run the loop with the self-pair test, every other tile
// the plain one: 7 fp32 instructions + 1 transcendental per pair (the eps = 0 guard adds a compare and a select).
constexpr int kEnergyRows = 4;

template <bool PPS, bool SELF, bool GUARD>
__device__ __forceinline__ void energy_tile(const float4 *tile, const float *etile, const float4 (&pi)[kEnergyRows],
                                            const float (&ei2)[kEnergyRows], const int (&gi)[kEnergyRows], int j0, float eps2,
                                            double (&phi)[kEnergyRows])
{
    float s[kEnergyRows] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int jj = 0; jj < kTile; ++jj) {
        const float4 pj = tile[jj];
#pragma unroll
        for (int k = 0; k < kEnergyRows; ++k) {
            const float dx = pj.x - pi[k].x, dy = pj.y - pi[k].y, dz = pj.z - pi[k].z;
            const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, PPS ? ei2[k] + etile[jj] : eps2)));
            float inv = __builtin_amdgcn_rsqf(GUARD ? guard_r2(r2) : r2);  // guarded: as the forces, pairs closer than 2.3e-13 drop out
            if (SELF)
                inv = (j0 + jj != gi[k]) ? inv : 0.f;
            s[k] = __builtin_fmaf(pj.w, inv, s[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < kEnergyRows; ++k)
        phi[k] += (double)s[k];
}

template <bool PPS, bool GUARD>
__global__ __launch_bounds__(kTile) void energy_kernel(const float4 *pos_all, const float4 *vel_rows,
                                                       double *block_out, int row_lo, int row_count, int n_total,
                                                       float eps2, const float *eps_pp)
{
    __shared__ float4 tile[kTile];
    __shared__ float etile[PPS ? kTile : 1];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * (kTile * kEnergyRows);
    float4 pi[kEnergyRows];
    float ei2[kEnergyRows];
    int gi[kEnergyRows];
    double phi[kEnergyRows];
#pragma unroll
    for (int k = 0; k < kEnergyRows; ++k) {
        const int r = row0 + k * kTile + tid;
        gi[k] = r < row_count ? row_lo + r : -1;
        pi[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        ei2[k] = eps2;
        phi[k] = 0.0;
        if (gi[k] >= 0) {
            pi[k] = pos_all[gi[k]];
            if (PPS)
                ei2[k] = __builtin_fmaf(eps_pp[gi[k]], eps_pp[gi[k]], eps2);
        }
    }
    // the tiles that hold this workgroup's rows (row_lo is a multiple of the tile: a split boundary)
    const int self_lo = row_lo + row0, self_hi = self_lo + kTile * kEnergyRows;
    for (int j0 = 0; j0 < n_total; j0 += kTile) {
        __syncthreads();
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        float et = 0.f;
        if (j0 + tid < n_total) {
            pt = pos_all[j0 + tid];
            if (PPS)
                et = eps_pp[j0 + tid];
        }
        tile[tid] = pt;
        if (PPS)
            etile[tid] = et * et;
        __syncthreads();
        if (j0 + kTile > self_lo && j0 < self_hi)
            energy_tile<PPS, true, GUARD>(tile, etile, pi, ei2, gi, j0, eps2, phi);
        else
            energy_tile<PPS, false, GUARD>(tile, etile, pi, ei2, gi, j0, eps2, phi);
    }
    double v[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kEnergyRows; ++k)
        if (gi[k] >= 0) {
            const float4 w = vel_rows[gi[k] - row_lo];
            v[0] += 0.5 * (double)pi[k].w * ((double)w.x * w.x + (double)w.y * w.y + (double)w.z * w.z);
            v[1] += -0.5 * (double)pi[k].w * phi[k];
        }
    block_reduce_store<2>(v, block_out);
}

int energy_kernel_blocks(int row_count) { return (row_count + kTile * kEnergyRows - 1) / (kTile * kEnergyRows); }

int energy_blocks(int row_count) { return (row_count + kTile - 1) / kTile; }

hipError_t launch_energy(const float4 *pos_all, const float4 *vel_rows, double *block_out, int row_lo,
                         int row_count, int n_total, float eps2, const float *eps_pp, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    const dim3 grid(energy_kernel_blocks(row_count)), block(kTile);
    // a particle may have eps = 0: with per-particle softening the guard stays on when the global eps is 0
    const bool guard = !(eps2 > 0.f);
    if (eps_pp && guard)
        hipLaunchKernelGGL((energy_kernel<true, true>), grid, block, 0, stream, pos_all, vel_rows, block_out, row_lo, row_count, n_total, eps2, eps_pp);
    else if (eps_pp)
        hipLaunchKernelGGL((energy_kernel<true, false>), grid, block, 0, stream, pos_all, vel_rows, block_out, row_lo, row_count, n_total, eps2, eps_pp);
    else if (guard)
        hipLaunchKernelGGL((energy_kernel<false, true>), grid, block, 0, stream, pos_all, vel_rows, block_out, row_lo, row_count, n_total, eps2, eps_pp);
    else
        hipLaunchKernelGGL((energy_kernel<false, false>), grid, block, 0, stream, pos_all, vel_rows, block_out, row_lo, row_count, n_total, eps2, eps_pp);
    return hipGetLastError();
}

__global__ __launch_bounds__(kTile) void momentum_kernel(const float4 *pos_all, const float4 *vel_rows,
                                                         double *block_out, int row_lo, int row_count)
{
    const int r = blockIdx.x * kTile + threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < row_count) {
        const float4 p = pos_all[row_lo + r];
        const float4 w = vel_rows[r];
        v[0] = (double)p.w * w.x;
        v[1] = (double)p.w * w.y;
        v[2] = (double)p.w * w.z;
        v[3] = (double)p.w;
    }
    block_reduce_store<4>(v, block_out);
}

hipError_t launch_momentum(const float4 *pos_all, const float4 *vel_rows, double *block_out, int row_lo,
                           int row_count, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(momentum_kernel, dim3(energy_blocks(row_count)), dim3(kTile), 0, stream, pos_all, vel_rows,
                       block_out, row_lo, row_count);
    return hipGetLastError();
}

}  // namespace nbody
