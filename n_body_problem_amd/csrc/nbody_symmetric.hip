// nbody_symmetric.hip -- the pair-once force kernels: the headline force mode of bench.py.
//
// The reference's own contribution ("method C", main_project/kernel.cu:703-774) evaluates each unordered pair
// once on an upper-triangular grid of 256 x 256 tiles and applies it to both bodies (Newton's third law) through
// shared-memory and global float atomics -- which its author names as the bottleneck (kernel.cu:757) and which
// make the result order-dependent.  This kernel keeps the idea and drops the atomics:
//
//   * tiles are pairs of splits (R, C), split_len bodies each; ONE workgroup (4 wave64; 2 or 1 for short splits) per tile, so every
//     partial sum is produced by exactly one workgroup (d = (C - R) mod n_splits, the tile's ring distance):
//        rows b in R, columns c in C:  P_row[d][b]    = sum_c m_c f(b,c)   (row side, registers)
//                                      P_col[R][d][c] = -sum_b m_b f(b,c)  (column side, LDS)
//     each unordered pair {R, C} is one tile, with the side sym_rows_side() names as rows -- so that the tiles of a
//     context that owns the rows of some splits are exactly the ones with R among them (sharding over GPUs: the
//     P_col of remote bodies are summed per group of splits, exchanged once per step and added in a fixed order);
//   * inside a wave, lane l owns 4 or 8 rows and meets the columns of the wave's current 64-column group one step at a
//     time -- column (l+s) mod 64 at step s in the one-column loops, columns (l+s) and (l+s+32) mod 64 at step s = 0..31
//     in the packed loops (the defaults), where every v_pk_*_f32 serves both; the column accumulators travel with their
//     columns, one lane per step (ds_bpermute_b32), so after the last step a column's sum sits in one lane and is added to
//     the LDS array without conflicts;
//   * the 4 waves walk the column groups in a rotated order, G/4 groups apart, with a barrier every G/4 groups, so
//     no two waves touch the same LDS entries at a time and every entry receives its terms in a fixed order:
//     bit-reproducible;
//   * small workgroups and 33 KiB of LDS (split_len 2048: 8 KiB of per-wave stages + 24 KiB of column sums) put 4-5
//     workgroups = 4-5 waves per SIMD on a CU (3 for the eight-row loop with arbitrary masses), enough for the phased
//     schedule of force_kernel_r4 to hide each wave's slow window after its v_rsq_f32 batch;
//   * diagonal tiles (R == C) are a separate, compiler-scheduled kernel that visits every (row, column) combination
//     and keeps the pairs with row index < column index (1/n_splits of the work); the same kernel, without the mask,
//     computes every tile when per-particle softening is on.
//
// Per unordered pair: 3 sub, 3 fma, rsq, 4 mul, 6 fma = 16 fp32 operations + 1 transcendental (14 + 1 on tiles whose two
// splits each carry one mass), issued as 8 (7) packed instructions in the default loops, plus three LDS reads and six
// ds_bpermute_b32 per step of 8 or 16 pairs -- against 2 x (12 + 1) for the two ordered interactions it replaces.
// Loops, in the order they appear: SY_* one column per step (kept for A/B and eps = 0) . S2_* packed, equal-mass tiles .
// S8_* the same with eight rows per lane (the headline loop) . S9_* eight rows, arbitrary masses . S3_* packed four rows,
// arbitrary masses.
#include "nbody_kernels.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace nbody {

// A tile's workgroup has W wave64 (W x 256 rows per pass): 4 for splits of 1024 bodies or more, fewer for the shorter
// splits of small systems, so that no wave is left without rows (sym_waves()).
constexpr int kSymRows = 4;  // rows per lane
constexpr int kSymStageFloatsPerWave = 512;  // per wave 2 KiB, 1 KiB-aligned: one 64-body group as float4[64] (or SoA, below)

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    const int i = __builtin_bit_cast(int, v);  // every lane is written by a wave rotate: "old" is never used
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
// lane i receives the value of lane (i + 1) mod 64
__device__ __forceinline__ float wave_rol1(float v) { return dpp_move<0x134>(v); }

// ---- hand-allocated 64-step loop of an off-diagonal tile (same arithmetic and order as the C++ loop below) -----
// Phases per step as in force_kernel_r4 (nbody_kernels.hip): the next step's LDS address . 4 x (sub,sub,sub,fma,fma,
// fma) . 4 x v_rsq_f32 . next ds_read_b128 . idle gap . 4 x (4 mul, 6 fma) . send the column accumulators one lane on.
// VGPR banks (index mod 4): no instruction reads two sources from one bank, and an accumulating v_fmac counts its
// destination as a source -- with the sums in the banks of dx,dy,dz this loop measured 18 conflicts x 2 cycles a step.
//   v[2:5] / v[6:9]   column body {x,y,z,m} = banks 2,3,0,1, double-buffered
//   row k=0..3        {x,y,z,m} = v[12+4k : 15+4k] = banks 0,1,2,3
//   temps k           {dx,dy,dz, r2/inv/s_row} = v[28+4k : 31+4k] = banks 0,1,2,3
//   shared            inv^2 = v44/v48 (bank 0)   inv^3 = v46/v50 (bank 2)   s_col = v47/v51 (bank 3)
//   row sums k        az = v(52+4k) (bank 0)  ax = v(53+4k) (bank 1)  ay = v(54+4k) (bank 2)
//   column sums       cx = v45 (bank 1)  cy = v68 (bank 0)  cz = v49 (bank 1)   (travelling)
//   v11 = eps^2 (bank 3)   v69 = 2^-84, v70 = +inf (GUARD)   v59 = 4 * ((lane + 1) mod 64), the permute's source lane
//   v0 = LDS byte address of the next read = v10 | (v1 & v55): v1 counts 16 bytes per step from 16*lane, v55 = 1023,
//   v10 = the wave's 1 KiB-aligned group buffer -- column (lane + s) mod 64 without a second copy of the group.
typedef float nb_f4 __attribute__((ext_vector_type(4)));
typedef float nb_f2 __attribute__((ext_vector_type(2)));
#define SY_PRE(PX, PY, PZ, X, Y, Z, D0, D1, D2, R, GRD)                                                          \
    "v_sub_f32_e32 " D0 ", " PX ", " X "\n\tv_sub_f32_e32 " D1 ", " PY ", " Y "\n\tv_sub_f32_e32 " D2 ", " PZ ", " Z "\n\t" \
    "v_fma_f32 " R ", " D0 ", " D0 ", v11\n\tv_fmac_f32_e32 " R ", " D1 ", " D1 "\n\tv_fmac_f32_e32 " R ", " D2 ", " D2 "\n\t" GRD(R)
#define SY_NOGUARD(R) ""
#define SY_GUARD(R) "v_cmp_le_f32_e32 vcc, v69, " R "\n\tv_cndmask_b32_e32 " R ", v70, " R ", vcc\n\t"
#define SY_POST(PM, M, AX, AY, AZ, D0, D1, D2, R, Q, T, SC)                                                      \
    "v_mul_f32_e32 " Q ", " R ", " R "\n\tv_mul_f32_e32 " T ", " R ", " Q "\n\t"                                      \
    "v_mul_f32_e32 " SC ", " M ", " T "\n\tv_mul_f32_e32 " R ", " PM ", " T "\n\t"                                    \
    "v_fmac_f32_e32 " AX ", " D0 ", " R "\n\tv_fmac_f32_e32 " AY ", " D1 ", " R "\n\tv_fmac_f32_e32 " AZ ", " D2 ", " R "\n\t" \
    "v_fmac_f32_e32 v45, " D0 ", " SC "\n\tv_fmac_f32_e32 v68, " D1 ", " SC "\n\tv_fmac_f32_e32 v49, " D2 ", " SC "\n\t"
// Tiles whose row bodies all have one mass and whose column bodies all have one mass (SymArgs::split_mass: every tile of
// an equal-mass system, most tiles of a few-species one): the masses leave the loop -- the sums collect d * inv^3 and
// are multiplied by the other side's mass once, when they are written out -- which drops the two mass multiplies of a
// pair: 14 fp32 instructions + 1 transcendental instead of 16 + 1.  inv^3 is built in the register of inv (bank 3).
#define SY_POSTU(PM, M, AX, AY, AZ, D0, D1, D2, R, Q, T, SC)                                                     \
    "v_mul_f32_e32 " Q ", " R ", " R "\n\tv_mul_f32_e32 " R ", " R ", " Q "\n\t"                                      \
    "v_fmac_f32_e32 " AX ", " D0 ", " R "\n\tv_fmac_f32_e32 " AY ", " D1 ", " R "\n\tv_fmac_f32_e32 " AZ ", " D2 ", " R "\n\t" \
    "v_fmac_f32_e32 v45, " D0 ", " R "\n\tv_fmac_f32_e32 v68, " D1 ", " R "\n\tv_fmac_f32_e32 v49, " D2 ", " R "\n\t"
// Wave priority: a wave runs its PRE + v_rsq_f32 phases at priority 2 and drops to 0 for the POST phase, so the SIMD
// issues a waiting wave's rsq batch before another wave's long POST stretch and the wave then sits out its slow window
//  while the others issue -- measured 1.7-2.3 % faster than equal priorities, and with it the
// explicit idle gap hardly matters any more (0 / 6 / 12 / 20 / 28 wait states: 194.0 / 193.7 / 194.3 / 194.5 / 195.1 ms
// against 198.3 ms without priorities, same box).
#ifndef NB_SYM_PRIO_POST
#define NB_SYM_PRIO_POST "s_setprio 0\n\t"
#define NB_SYM_PRIO_PRE "s_setprio 2\n\t"
#endif
#ifndef NB_SYM_GAP
#define NB_SYM_GAP "s_nop 5\n\t"
#endif
// The column sums move one lane per step through the LDS crossbar (ds_bpermute_b32, v59 = 4 * ((lane + 1) mod 64)):
// three v_mov_b32_dpp wave_rol:1 measured 38 VALU cycles a step (tools/gen_sched2.py), the permutes 6.  LDS operations
// of a wave complete in order, so "lgkmcnt(3)" at the top of a step means the column body has arrived (the three
// permutes issued after it may still be in flight) and "lgkmcnt(1)" before the POST phase means the permutes have (the
// next column's read may not).
#define SY_ROTATE                                                                                                \
    "ds_bpermute_b32 v45, v59, v45\n\tds_bpermute_b32 v68, v59, v68\n\tds_bpermute_b32 v49, v59, v49\n\t"
#define SY_STEP(PX, PY, PZ, PM, NEXT, GRD, POST)                                                                       \
    "v_add_u32_e32 v1, 16, v1\n\t"                                                                               \
    "v_and_or_b32 v0, v1, v55, v10\n\t"                                                                          \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                   \
    SY_PRE(PX, PY, PZ, "v12", "v13", "v14", "v28", "v29", "v30", "v31", GRD)                                     \
    SY_PRE(PX, PY, PZ, "v16", "v17", "v18", "v32", "v33", "v34", "v35", GRD)                                     \
    SY_PRE(PX, PY, PZ, "v20", "v21", "v22", "v36", "v37", "v38", "v39", GRD)                                     \
    SY_PRE(PX, PY, PZ, "v24", "v25", "v26", "v40", "v41", "v42", "v43", GRD)                                     \
    "v_rsq_f32_e32 v31, v31\n\tv_rsq_f32_e32 v35, v35\n\tv_rsq_f32_e32 v39, v39\n\tv_rsq_f32_e32 v43, v43\n\t"       \
    NEXT                                                                                                         \
    NB_SYM_GAP                                                                                                   \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                   \
    NB_SYM_PRIO_POST                                                                                             \
    POST(PM, "v15", "v53", "v54", "v52", "v28", "v29", "v30", "v31", "v44", "v46", "v47")                     \
    POST(PM, "v19", "v57", "v58", "v56", "v32", "v33", "v34", "v35", "v48", "v50", "v51")                     \
    POST(PM, "v23", "v61", "v62", "v60", "v36", "v37", "v38", "v39", "v44", "v46", "v47")                     \
    POST(PM, "v27", "v65", "v66", "v64", "v40", "v41", "v42", "v43", "v48", "v50", "v51")                     \
    NB_SYM_PRIO_PRE                                                                                              \
    SY_ROTATE
#define SY_GROUP_LOOP(GRD, POST)                                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t" /* nothing of the compiler's may be counted by the waits below */                 \
    "v_and_or_b32 v0, v1, v55, v10\n\t"                                                                          \
    "ds_read_b128 v[2:5], v0\n\t"                                                                                \
    SY_ROTATE /* of zeros: primes the in-order LDS queue so that every step sees the same pattern */             \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 32\n"                                                                                     \
    "1:\n\t"                                                                                                     \
    SY_STEP("v2", "v3", "v4", "v5", "ds_read_b128 v[6:9], v0\n\t", GRD, POST)                                          \
    SY_STEP("v6", "v7", "v8", "v9", "ds_read_b128 v[2:5], v0\n\t", GRD, POST)                                          \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- equal-mass tiles, packed: TWO columns per step, one packed fp32 instruction for both (SymArgs::packed, the default) ----
// Lane l meets columns (l + s) and (l + s + 32) mod 64 at step s = 0..31, the pair sharing every v_pk_{add,mul,fma}_f32:
// 7 packed instructions + 1 transcendental per pair evaluation instead of 14 + 1 (the cycles are the same, 4 per packed
// instruction; what halves is the number of instructions issued -- the kernel sits on the power limit).  The group is
// staged as three arrays x[128], y[128], z[128] (the 64 columns twice, so neither the step nor column + 32 needs a wrap)
// and read with ds_read2_b32; both columns' sums travel one lane per step (six permutes for eight pairs); the row sums are
// kept per column of the pair and added when the pass ends.  Two batches of 2 rows x 2 columns per step keep the batch of
// four v_rsq_f32 of the other loops.  64-bit operands in even pairs; pair classes ((reg / 2) mod 2) of src0 and src1 differ:
//   column pair  PX = v[2:3] (1)  PZ = v[4:5] (0)  PY = v[6:7] (1)        v[8:9] = (eps^2, -)
//   rows         v[12:27] as above: (x, y) pairs class 0, (z, m) pairs class 1
//   temps A / B  R = v[28:29] / v[32:33] (0)   DX, DY, DZ = v[30:31], v[34:35], v[38:39] / v[42:43], v[46:47], v[50:51] (1)
//   Q = v[54:55] (1)     column sums CX, CY, CZ = v[36:37], v[40:41], v[44:45]      row sums v[56:79]: row k, component c at
//   v[56 + 6k + 2c : 57 + 6k + 2c]          v0 / v1 / v53: address of x and y, address of z (and m), permute source
#define S2_NEG " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define S2_PRE(RXY, RZM, DX, DY, DZ, R)                                                                          \
    "v_pk_add_f32 " DX ", v[2:3], " RXY " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                      \
    "v_pk_add_f32 " DY ", v[6:7], " RXY " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                      \
    "v_pk_add_f32 " DZ ", v[4:5], " RZM " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                      \
    "v_pk_fma_f32 " R ", " DX ", " DX ", v[8:9] op_sel_hi:[1,1,0]\n\t"                                              \
    "v_pk_fma_f32 " R ", " DY ", " DY ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DZ ", " DZ ", " R "\n\t"
#define S2_POST(AX, AY, AZ, DX, DY, DZ, R)                                                                       \
    "v_pk_mul_f32 v[54:55], " R ", " R "\n\tv_pk_mul_f32 " R ", " R ", v[54:55]\n\t"                                 \
    "v_pk_fma_f32 " AX ", " DX ", " R ", " AX "\n\tv_pk_fma_f32 " AY ", " DY ", " R ", " AY "\n\t"                     \
    "v_pk_fma_f32 " AZ ", " DZ ", " R ", " AZ "\n\t"                                                                 \
    "v_pk_fma_f32 v[36:37], " DX ", " R ", v[36:37]\n\tv_pk_fma_f32 v[40:41], " DY ", " R ", v[40:41]\n\t"             \
    "v_pk_fma_f32 v[44:45], " DZ ", " R ", v[44:45]\n\t"
#ifdef NB_EXP_NOROT  /* timing experiment only (wrong sums): what the six permutes of a step cost */
#define S2_ROTATE ""
#define S2_WAIT_TOP "s_waitcnt lgkmcnt(0)\n\t"
#else
#define S2_ROTATE                                                                                                \
    "ds_bpermute_b32 v36, v53, v36\n\tds_bpermute_b32 v37, v53, v37\n\tds_bpermute_b32 v40, v53, v40\n\t"            \
    "ds_bpermute_b32 v41, v53, v41\n\tds_bpermute_b32 v44, v53, v44\n\tds_bpermute_b32 v45, v53, v45\n\t"
#define S2_WAIT_TOP "s_waitcnt lgkmcnt(6)\n\t" /* the column pair has arrived; the six permutes behind it may be in flight */
#endif
// The group is staged as x[128], y[128], z[128] (, m[128]): the 64 columns twice, so lane l reads columns l + s and
// l + s + 32 at dword l + s (+ 32) without wrapping and the step number is an immediate offset: four steps per loop
// iteration, the two addresses (v0: x at +0, y at +128 dwords; v1 = v0 + 1024 bytes: z at +0, m at +128) advance once per
// iteration instead of an add and a mask per step.
#ifdef NB_EXP_NOREAD  /* timing experiment only (wrong sums): what the three reads of a step cost */
#define S2_READ(X0, X1, Y0, Y1) ""
#else
#define S2_READ(X0, X1, Y0, Y1)                                                                                  \
    "ds_read2_b32 v[2:3], v0 offset0:" X0 " offset1:" X1 "\n\tds_read2_b32 v[6:7], v0 offset0:" Y0 " offset1:" Y1 "\n\t" \
    "ds_read2_b32 v[4:5], v1 offset0:" X0 " offset1:" X1 "\n\t"
#endif
#define S2_ADVANCE "v_add_u32_e32 v0, 16, v0\n\tv_add_u32_e32 v1, 16, v1\n\t"
#define S2_STEP(NEXT)                                                                                            \
    S2_WAIT_TOP                                                                                                  \
    S2_PRE("v[12:13]", "v[14:15]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")                               \
    S2_PRE("v[16:17]", "v[18:19]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")                               \
    "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"       \
    NB_SYM_GAP                                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t" /* the column sums have arrived from the next lane */                             \
    NB_SYM_PRIO_POST                                                                                             \
    S2_POST("v[56:57]", "v[58:59]", "v[60:61]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")                  \
    S2_POST("v[62:63]", "v[64:65]", "v[66:67]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")                  \
    NB_SYM_PRIO_PRE                                                                                              \
    S2_PRE("v[20:21]", "v[22:23]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")                               \
    S2_PRE("v[24:25]", "v[26:27]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")                               \
    NEXT /* the next step's column pair: the current one has been consumed by the four PRE blocks */             \
    "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"       \
    NB_SYM_GAP                                                                                                   \
    NB_SYM_PRIO_POST                                                                                             \
    S2_POST("v[68:69]", "v[70:71]", "v[72:73]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")                  \
    S2_POST("v[74:75]", "v[76:77]", "v[78:79]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")                  \
    NB_SYM_PRIO_PRE                                                                                              \
    S2_ROTATE
#define S2_GROUP_LOOP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160")                                                                             \
    S2_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S2_STEP(S2_READ("1", "33", "129", "161"))                                                                    \
    S2_STEP(S2_READ("2", "34", "130", "162"))                                                                    \
    S2_STEP(S2_READ("3", "35", "131", "163"))                                                                    \
    S2_STEP(S2_ADVANCE S2_READ("0", "32", "128", "160"))                                                         \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- equal-mass tiles, EIGHT rows per lane (SymArgs::packed == 2): the same two-columns-per-step loop with twice the rows
// behind every column pair, so the LDS traffic of a step (three reads, six permutes: 8 % of the pass, measured by leaving
// them out, profiles/r02_ab_lds_cost.txt) serves 16 pair evaluations instead of 8.  A wave covers 512 rows (two waves per
// 1024-body split), 4 waves per SIMD instead of 5.  Rows as (x, y) pairs (class 0) and (z, z') pairs of two rows (class 1):
//   rows 0..3   v[12:13] v[16:17] v[20:21] v[24:25]   rows 4..7   v[48:49] v[52:53] v[56:57] v[60:61]
//   (z0, z1) v[14:15]   (z2, z3) v[18:19]   (z4, z5) v[22:23]   (z6, z7) v[26:27]
//   row sums: row k, component c at v[64 + 6k + 2c : 65 + 6k + 2c]      v10: permute source      temps, column pair, column
//   sums, eps^2, addresses as in the four-row loop.  Four batches of 2 rows x 2 columns per step.
#define S8_ZLO " op_sel:[0,0] op_sel_hi:[1,0]"
#define S8_ZHI " op_sel:[0,1] op_sel_hi:[1,1]"
#define S8_PRE(...) S8_PRE_I(__VA_ARGS__) /* S8_TA / S8_TB expand to four arguments */
#define S8_POST(...) S2_POST(__VA_ARGS__)
#define S8_PRE_I(RXY, RZZ, ZSEL, DX, DY, DZ, R)                                                                  \
    "v_pk_add_f32 " DX ", v[2:3], " RXY " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                      \
    "v_pk_add_f32 " DY ", v[6:7], " RXY " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                      \
    "v_pk_add_f32 " DZ ", v[4:5], " RZZ ZSEL S2_NEG                                                                 \
    "v_pk_fma_f32 " R ", " DX ", " DX ", v[8:9] op_sel_hi:[1,1,0]\n\t"                                              \
    "v_pk_fma_f32 " R ", " DY ", " DY ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DZ ", " DZ ", " R "\n\t"
#define S8_ROTATE                                                                                                \
    "ds_bpermute_b32 v36, v10, v36\n\tds_bpermute_b32 v37, v10, v37\n\tds_bpermute_b32 v40, v10, v40\n\t"            \
    "ds_bpermute_b32 v41, v10, v41\n\tds_bpermute_b32 v44, v10, v44\n\tds_bpermute_b32 v45, v10, v45\n\t"
#define S8_RSQ "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"
#define S8_TA "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]"
#define S8_TB "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]"
// A batch = two rows (temps A and B) x two columns, its instructions interleaved A, B: consecutive instructions then work
// on the same kind of operand (the column pair's x against row A's x, then row B's x: few bits toggle) and no result is used
// by the next instruction.  Each sum still receives its terms in the order A, B.  Q of row B in v[58:59].
#ifdef NB_S8_SEQUENTIAL /* the batch as two rows one after the other (A/B measurement) */
#define S8_PRE2(RXYA, RXYB, RZZ) S8_PRE(RXYA, RZZ, S8_ZLO, S8_TA) S8_PRE(RXYB, RZZ, S8_ZHI, S8_TB)
#define S8_POST2(AXA, AYA, AZA, AXB, AYB, AZB) S8_POST(AXA, AYA, AZA, S8_TA) S8_POST(AXB, AYB, AZB, S8_TB)
#else
#define S8_PRE2(RXYA, RXYB, RZZ)                                                                                 \
    "v_pk_add_f32 v[30:31], v[2:3], " RXYA " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                   \
    "v_pk_add_f32 v[42:43], v[2:3], " RXYB " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                   \
    "v_pk_add_f32 v[34:35], v[6:7], " RXYA " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                   \
    "v_pk_add_f32 v[46:47], v[6:7], " RXYB " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                   \
    "v_pk_add_f32 v[38:39], v[4:5], " RZZ S8_ZLO S2_NEG                                                             \
    "v_pk_add_f32 v[50:51], v[4:5], " RZZ S8_ZHI S2_NEG                                                             \
    "v_pk_fma_f32 v[28:29], v[30:31], v[30:31], v[8:9] op_sel_hi:[1,1,0]\n\t"                                       \
    "v_pk_fma_f32 v[32:33], v[42:43], v[42:43], v[8:9] op_sel_hi:[1,1,0]\n\t"                                       \
    "v_pk_fma_f32 v[28:29], v[34:35], v[34:35], v[28:29]\n\tv_pk_fma_f32 v[32:33], v[46:47], v[46:47], v[32:33]\n\t" \
    "v_pk_fma_f32 v[28:29], v[38:39], v[38:39], v[28:29]\n\tv_pk_fma_f32 v[32:33], v[50:51], v[50:51], v[32:33]\n\t"
#define S8_POST2(AXA, AYA, AZA, AXB, AYB, AZB)                                                                   \
    "v_pk_mul_f32 v[54:55], v[28:29], v[28:29]\n\tv_pk_mul_f32 v[58:59], v[32:33], v[32:33]\n\t"                     \
    "v_pk_mul_f32 v[28:29], v[28:29], v[54:55]\n\tv_pk_mul_f32 v[32:33], v[32:33], v[58:59]\n\t"                     \
    "v_pk_fma_f32 " AXA ", v[30:31], v[28:29], " AXA "\n\tv_pk_fma_f32 " AXB ", v[42:43], v[32:33], " AXB "\n\t"     \
    "v_pk_fma_f32 " AYA ", v[34:35], v[28:29], " AYA "\n\tv_pk_fma_f32 " AYB ", v[46:47], v[32:33], " AYB "\n\t"     \
    "v_pk_fma_f32 " AZA ", v[38:39], v[28:29], " AZA "\n\tv_pk_fma_f32 " AZB ", v[50:51], v[32:33], " AZB "\n\t"     \
    "v_pk_fma_f32 v[36:37], v[30:31], v[28:29], v[36:37]\n\tv_pk_fma_f32 v[40:41], v[34:35], v[28:29], v[40:41]\n\t" \
    "v_pk_fma_f32 v[44:45], v[38:39], v[28:29], v[44:45]\n\t"                                                       \
    "v_pk_fma_f32 v[36:37], v[42:43], v[32:33], v[36:37]\n\tv_pk_fma_f32 v[40:41], v[46:47], v[32:33], v[40:41]\n\t" \
    "v_pk_fma_f32 v[44:45], v[50:51], v[32:33], v[44:45]\n\t"
#endif
#define S8_STEP(NEXT)                                                                                            \
    "s_waitcnt lgkmcnt(6)\n\t" /* the column pair has arrived; the six permutes behind it may be in flight */    \
    S8_PRE2("v[12:13]", "v[16:17]", "v[14:15]")                                                                  \
    S8_RSQ NB_SYM_GAP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t" /* the column sums have arrived from the next lane */                             \
    NB_SYM_PRIO_POST                                                                                             \
    S8_POST2("v[64:65]", "v[66:67]", "v[68:69]", "v[70:71]", "v[72:73]", "v[74:75]")                             \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_PRE2("v[20:21]", "v[24:25]", "v[18:19]")                                                                  \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S8_POST2("v[76:77]", "v[78:79]", "v[80:81]", "v[82:83]", "v[84:85]", "v[86:87]")                             \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_PRE2("v[48:49]", "v[52:53]", "v[22:23]")                                                                  \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S8_POST2("v[88:89]", "v[90:91]", "v[92:93]", "v[94:95]", "v[96:97]", "v[98:99]")                             \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_PRE2("v[56:57]", "v[60:61]", "v[26:27]")                                                                  \
    NEXT /* the next step's column pair: the current one has been consumed by the eight PRE blocks */            \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S8_POST2("v[100:101]", "v[102:103]", "v[104:105]", "v[106:107]", "v[108:109]", "v[110:111]")                 \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_ROTATE
#define S8_GROUP_LOOP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160")                                                                             \
    S8_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S8_STEP(S2_READ("1", "33", "129", "161"))                                                                    \
    S8_STEP(S2_READ("2", "34", "130", "162"))                                                                    \
    S8_STEP(S2_READ("3", "35", "131", "163"))                                                                    \
    S8_STEP(S2_ADVANCE S2_READ("0", "32", "128", "160"))                                                         \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- EIGHT rows per lane for tiles with arbitrary masses (round 3): the loop above with the masses back in -- 8 packed
// instructions + 1 transcendental per pair, the LDS traffic of a step (four reads, six permutes) behind 16 pair evaluations.
// The row masses (8), the column pair's masses (2) and a second product register per row of a batch (4) do not fit the 128
// VGPRs of four waves per SIMD: this loop lives in its own kernel instantiation allocated for THREE waves per SIMD, which
// costs the eight-row schedule next to nothing (0-0.9 % on the equal-mass loop with the workgroups per CU cut from 4 to 3 by
// LDS padding, profiles/r03_ab_eight_rows_three_waves_lds_pad.txt) against the 3.4 % the eight rows buy.
//   PM = v[114:115] (class 1) the column pair's masses, read through v1 (z at +0, m at +128 dwords) at the END of a step
//   row masses (m0,m1) v[118:119]  (m2,m3) v[122:123]  (m4,m5) v[126:127]  (m6,m7) v[130:131]  (class 1: they multiply R, class 0)
//   S_A = v[112:113], S_B = v[116:117] (class 0) = m_col * inv^3, the factor of the row sums (R becomes m_row * inv^3)
#define S9_POST2(AXA, AYA, AZA, AXB, AYB, AZB, MROW)                                                             \
    "v_pk_mul_f32 v[54:55], v[28:29], v[28:29]\n\tv_pk_mul_f32 v[58:59], v[32:33], v[32:33]\n\t"                     \
    "v_pk_mul_f32 v[28:29], v[28:29], v[54:55]\n\tv_pk_mul_f32 v[32:33], v[32:33], v[58:59]\n\t"                     \
    "v_pk_mul_f32 v[112:113], v[114:115], v[28:29]\n\tv_pk_mul_f32 v[116:117], v[114:115], v[32:33]\n\t"             \
    "v_pk_mul_f32 v[28:29], v[28:29], " MROW " op_sel:[0,0] op_sel_hi:[1,0]\n\t"                                    \
    "v_pk_mul_f32 v[32:33], v[32:33], " MROW " op_sel:[0,1] op_sel_hi:[1,1]\n\t"                                    \
    "v_pk_fma_f32 " AXA ", v[30:31], v[112:113], " AXA "\n\tv_pk_fma_f32 " AXB ", v[42:43], v[116:117], " AXB "\n\t" \
    "v_pk_fma_f32 " AYA ", v[34:35], v[112:113], " AYA "\n\tv_pk_fma_f32 " AYB ", v[46:47], v[116:117], " AYB "\n\t" \
    "v_pk_fma_f32 " AZA ", v[38:39], v[112:113], " AZA "\n\tv_pk_fma_f32 " AZB ", v[50:51], v[116:117], " AZB "\n\t" \
    "v_pk_fma_f32 v[36:37], v[30:31], v[28:29], v[36:37]\n\tv_pk_fma_f32 v[40:41], v[34:35], v[28:29], v[40:41]\n\t" \
    "v_pk_fma_f32 v[44:45], v[38:39], v[28:29], v[44:45]\n\t"                                                       \
    "v_pk_fma_f32 v[36:37], v[42:43], v[32:33], v[36:37]\n\tv_pk_fma_f32 v[40:41], v[46:47], v[32:33], v[40:41]\n\t" \
    "v_pk_fma_f32 v[44:45], v[50:51], v[32:33], v[44:45]\n\t"
#define S9_MASSES(M0, M1) "ds_read2_b32 v[114:115], v1 offset0:" M0 " offset1:" M1 "\n\t"
#define S9_STEP(NEXT, NEXT_MASSES)                                                                               \
    "s_waitcnt lgkmcnt(7)\n\t" /* the positions of the column pair have arrived (masses and permutes may be in flight) */ \
    S8_PRE2("v[12:13]", "v[16:17]", "v[14:15]")                                                                  \
    S8_RSQ NB_SYM_GAP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t" /* the masses and the column sums have arrived */                                 \
    NB_SYM_PRIO_POST                                                                                             \
    S9_POST2("v[64:65]", "v[66:67]", "v[68:69]", "v[70:71]", "v[72:73]", "v[74:75]", "v[118:119]")               \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_PRE2("v[20:21]", "v[24:25]", "v[18:19]")                                                                  \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S9_POST2("v[76:77]", "v[78:79]", "v[80:81]", "v[82:83]", "v[84:85]", "v[86:87]", "v[122:123]")               \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_PRE2("v[48:49]", "v[52:53]", "v[22:23]")                                                                  \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S9_POST2("v[88:89]", "v[90:91]", "v[92:93]", "v[94:95]", "v[96:97]", "v[98:99]", "v[126:127]")               \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_PRE2("v[56:57]", "v[60:61]", "v[26:27]")                                                                  \
    NEXT /* the next step's positions: the current ones have been consumed by the eight PRE blocks */            \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S9_POST2("v[100:101]", "v[102:103]", "v[104:105]", "v[106:107]", "v[108:109]", "v[110:111]", "v[130:131]")   \
    NB_SYM_PRIO_PRE                                                                                              \
    NEXT_MASSES /* the next step's masses: the current ones have been consumed by the four POST blocks */        \
    S8_ROTATE
#define S9_GROUP_LOOP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160")                                                                             \
    S9_MASSES("128", "160")                                                                                      \
    S8_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S9_STEP(S2_READ("1", "33", "129", "161"), S9_MASSES("129", "161"))                                           \
    S9_STEP(S2_READ("2", "34", "130", "162"), S9_MASSES("130", "162"))                                           \
    S9_STEP(S2_READ("3", "35", "131", "163"), S9_MASSES("131", "163"))                                           \
    S9_STEP(S2_ADVANCE S2_READ("0", "32", "128", "160"), S9_MASSES("128", "160"))                                \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- EIGHT rows per lane with PER-PARTICLE SOFTENING (round 3): eps_ij^2 = eps^2 + eps_i^2 + eps_j^2 (the eps the
// reference loads into vel.w, kernel.cu:223, and never reads).  S9 plus one packed add per (row, column pair): the column
// pair's eps_j^2 (staged per wave, read with the positions) plus the row's eps^2 + eps_i^2 becomes the first addend of the
// r^2 chain -- the order of force_sym_general_kernel: fma(dx, dx, (eps^2 + eps_i^2) + eps_j^2).  8.5 packed instructions + 1
// transcendental per pair; its own kernel instantiation (ROWS8 = 3), used for splits of whole 512 bodies and eps > 0 (a
// particle may have eps_i = 0: with eps = 0 the guarded compiler-scheduled kernel keeps running).
//   EC = v[134:135] (class 1) the column pair's eps_j^2, read through v146 (its own staging array, the 64 columns twice)
//   row terms (e0,e1) v[132:133]  (e2,e3) v[136:137]  (e4,e5) v[140:141]  (e6,e7) v[144:145]  (class 0), e_k = eps^2 + eps_k^2
#define S10_PRE2(RXYA, RXYB, RZZ, ER)                                                                            \
    "v_pk_add_f32 v[30:31], v[2:3], " RXYA " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                   \
    "v_pk_add_f32 v[42:43], v[2:3], " RXYB " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                   \
    "v_pk_add_f32 v[34:35], v[6:7], " RXYA " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                   \
    "v_pk_add_f32 v[46:47], v[6:7], " RXYB " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                   \
    "v_pk_add_f32 v[38:39], v[4:5], " RZZ S8_ZLO S2_NEG                                                             \
    "v_pk_add_f32 v[50:51], v[4:5], " RZZ S8_ZHI S2_NEG                                                             \
    "v_pk_add_f32 v[28:29], v[134:135], " ER " op_sel:[0,0] op_sel_hi:[1,0]\n\t"                                     \
    "v_pk_add_f32 v[32:33], v[134:135], " ER " op_sel:[0,1] op_sel_hi:[1,1]\n\t"                                     \
    "v_pk_fma_f32 v[28:29], v[30:31], v[30:31], v[28:29]\n\tv_pk_fma_f32 v[32:33], v[42:43], v[42:43], v[32:33]\n\t" \
    "v_pk_fma_f32 v[28:29], v[34:35], v[34:35], v[28:29]\n\tv_pk_fma_f32 v[32:33], v[46:47], v[46:47], v[32:33]\n\t" \
    "v_pk_fma_f32 v[28:29], v[38:39], v[38:39], v[28:29]\n\tv_pk_fma_f32 v[32:33], v[50:51], v[50:51], v[32:33]\n\t"
#define S10_EPS(E0, E1) "ds_read2_b32 v[134:135], v146 offset0:" E0 " offset1:" E1 "\n\t"
#define S10_ADVANCE S2_ADVANCE "v_add_u32_e32 v146, 16, v146\n\t"
#define S10_STEP(NEXT, NEXT_MASSES)                                                                              \
    "s_waitcnt lgkmcnt(7)\n\t" /* positions and eps_j^2 of the column pair have arrived (masses, permutes may be in flight) */ \
    S10_PRE2("v[12:13]", "v[16:17]", "v[14:15]", "v[132:133]")                                                   \
    S8_RSQ NB_SYM_GAP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    NB_SYM_PRIO_POST                                                                                             \
    S9_POST2("v[64:65]", "v[66:67]", "v[68:69]", "v[70:71]", "v[72:73]", "v[74:75]", "v[118:119]")               \
    NB_SYM_PRIO_PRE                                                                                              \
    S10_PRE2("v[20:21]", "v[24:25]", "v[18:19]", "v[136:137]")                                                   \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S9_POST2("v[76:77]", "v[78:79]", "v[80:81]", "v[82:83]", "v[84:85]", "v[86:87]", "v[122:123]")               \
    NB_SYM_PRIO_PRE                                                                                              \
    S10_PRE2("v[48:49]", "v[52:53]", "v[22:23]", "v[140:141]")                                                   \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S9_POST2("v[88:89]", "v[90:91]", "v[92:93]", "v[94:95]", "v[96:97]", "v[98:99]", "v[126:127]")               \
    NB_SYM_PRIO_PRE                                                                                              \
    S10_PRE2("v[56:57]", "v[60:61]", "v[26:27]", "v[144:145]")                                                   \
    NEXT /* the next step's positions and eps_j^2 */                                                             \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S9_POST2("v[100:101]", "v[102:103]", "v[104:105]", "v[106:107]", "v[108:109]", "v[110:111]", "v[130:131]")   \
    NB_SYM_PRIO_PRE                                                                                              \
    NEXT_MASSES                                                                                                  \
    S8_ROTATE
#define S10_GROUP_LOOP                                                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160") S10_EPS("0", "32")                                                          \
    S9_MASSES("128", "160")                                                                                      \
    S8_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S10_STEP(S2_READ("1", "33", "129", "161") S10_EPS("1", "33"), S9_MASSES("129", "161"))                       \
    S10_STEP(S2_READ("2", "34", "130", "162") S10_EPS("2", "34"), S9_MASSES("130", "162"))                       \
    S10_STEP(S2_READ("3", "35", "131", "163") S10_EPS("3", "35"), S9_MASSES("131", "163"))                       \
    S10_STEP(S10_ADVANCE S2_READ("0", "32", "128", "160") S10_EPS("0", "32"), S9_MASSES("128", "160"))           \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- EIGHT rows per lane, per-particle softening, EQUAL-MASS tiles (round 4): S8 with S10's first addend -- the masses leave the
// loop again (the sums collect d * inv^3 and are multiplied by the other side's mass when they are written), 7.5 packed
// instructions + 1 transcendental per pair instead of 8.5 + 1: what a body set with ONE mass and individual softening lengths
// gets (the benchmark's sphere with vel.w in use).  Lives in the ROWS8 = 3 instantiation beside S10, chosen per tile.
#define S12_STEP(NEXT)                                                                                           \
    "s_waitcnt lgkmcnt(6)\n\t" /* positions and eps_j^2 of the column pair have arrived (the permutes may be in flight) */ \
    S10_PRE2("v[12:13]", "v[16:17]", "v[14:15]", "v[132:133]")                                                   \
    S8_RSQ NB_SYM_GAP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    NB_SYM_PRIO_POST                                                                                             \
    S8_POST2("v[64:65]", "v[66:67]", "v[68:69]", "v[70:71]", "v[72:73]", "v[74:75]")                             \
    NB_SYM_PRIO_PRE                                                                                              \
    S10_PRE2("v[20:21]", "v[24:25]", "v[18:19]", "v[136:137]")                                                   \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S8_POST2("v[76:77]", "v[78:79]", "v[80:81]", "v[82:83]", "v[84:85]", "v[86:87]")                             \
    NB_SYM_PRIO_PRE                                                                                              \
    S10_PRE2("v[48:49]", "v[52:53]", "v[22:23]", "v[140:141]")                                                   \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S8_POST2("v[88:89]", "v[90:91]", "v[92:93]", "v[94:95]", "v[96:97]", "v[98:99]")                             \
    NB_SYM_PRIO_PRE                                                                                              \
    S10_PRE2("v[56:57]", "v[60:61]", "v[26:27]", "v[144:145]")                                                   \
    NEXT /* the next step's positions and eps_j^2 */                                                             \
    S8_RSQ NB_SYM_GAP NB_SYM_PRIO_POST                                                                           \
    S8_POST2("v[100:101]", "v[102:103]", "v[104:105]", "v[106:107]", "v[108:109]", "v[110:111]")                 \
    NB_SYM_PRIO_PRE                                                                                              \
    S8_ROTATE
#define S12_GROUP_LOOP                                                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160") S10_EPS("0", "32")                                                          \
    S8_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S12_STEP(S2_READ("1", "33", "129", "161") S10_EPS("1", "33"))                                                \
    S12_STEP(S2_READ("2", "34", "130", "162") S10_EPS("2", "34"))                                                \
    S12_STEP(S2_READ("3", "35", "131", "163") S10_EPS("3", "35"))                                                \
    S12_STEP(S10_ADVANCE S2_READ("0", "32", "128", "160") S10_EPS("0", "32"))                                    \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- the same two-columns-per-step loop for tiles with arbitrary masses: 8 packed instructions + 1 transcendental per pair
// (one-column loop: 16 + 1).  The group's masses are staged as a fourth array m[128] behind x, y, z and read at the END of
// a step (the positions are consumed by the PRE blocks, the masses by the POST blocks of both batches).
//   PM = v[10:11] (1) the pair's masses, read through v1 (z at +0, m at +128 dwords)   S = v[80:81] (0) m_col * inv^3
#define S3_POST(RZM, AX, AY, AZ, DX, DY, DZ, R)                                                                  \
    "v_pk_mul_f32 v[54:55], " R ", " R "\n\tv_pk_mul_f32 " R ", " R ", v[54:55]\n\t"                                 \
    "v_pk_mul_f32 v[80:81], v[10:11], " R "\n\t"                                                                   \
    "v_pk_mul_f32 " R ", " R ", " RZM " op_sel:[0,1] op_sel_hi:[1,1]\n\t"                                           \
    "v_pk_fma_f32 " AX ", " DX ", v[80:81], " AX "\n\tv_pk_fma_f32 " AY ", " DY ", v[80:81], " AY "\n\t"             \
    "v_pk_fma_f32 " AZ ", " DZ ", v[80:81], " AZ "\n\t"                                                              \
    "v_pk_fma_f32 v[36:37], " DX ", " R ", v[36:37]\n\tv_pk_fma_f32 v[40:41], " DY ", " R ", v[40:41]\n\t"             \
    "v_pk_fma_f32 v[44:45], " DZ ", " R ", v[44:45]\n\t"
#define S3_MASSES(M0, M1) "ds_read2_b32 v[10:11], v1 offset0:" M0 " offset1:" M1 "\n\t"
#define S3_STEP(NEXT, NEXT_MASSES)                                                                               \
    "s_waitcnt lgkmcnt(7)\n\t" /* the positions of the column pair have arrived (masses and permutes may be in flight) */ \
    S2_PRE("v[12:13]", "v[14:15]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")                               \
    S2_PRE("v[16:17]", "v[18:19]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")                               \
    "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"       \
    NB_SYM_GAP                                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t" /* the masses and the column sums have arrived */                                 \
    NB_SYM_PRIO_POST                                                                                             \
    S3_POST("v[14:15]", "v[56:57]", "v[58:59]", "v[60:61]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")      \
    S3_POST("v[18:19]", "v[62:63]", "v[64:65]", "v[66:67]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")      \
    NB_SYM_PRIO_PRE                                                                                              \
    S2_PRE("v[20:21]", "v[22:23]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")                               \
    S2_PRE("v[24:25]", "v[26:27]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")                               \
    NEXT /* the next step's positions */                                                                         \
    "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"       \
    NB_SYM_GAP                                                                                                   \
    NB_SYM_PRIO_POST                                                                                             \
    S3_POST("v[22:23]", "v[68:69]", "v[70:71]", "v[72:73]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")      \
    S3_POST("v[26:27]", "v[74:75]", "v[76:77]", "v[78:79]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")      \
    NB_SYM_PRIO_PRE                                                                                              \
    NEXT_MASSES /* the next step's masses */                                                                     \
    S2_ROTATE
#define S3_GROUP_LOOP                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160")                                                                             \
    S3_MASSES("128", "160")                                                                                      \
    S2_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S3_STEP(S2_READ("1", "33", "129", "161"), S3_MASSES("129", "161"))                                           \
    S3_STEP(S2_READ("2", "34", "130", "162"), S3_MASSES("130", "162"))                                           \
    S3_STEP(S2_READ("3", "35", "131", "163"), S3_MASSES("131", "163"))                                           \
    S3_STEP(S2_ADVANCE S2_READ("0", "32", "128", "160"), S3_MASSES("128", "160"))                                \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ---- the four-row loop for arbitrary masses under PER-PARTICLE SOFTENING (splits that are not whole multiples of 512 bodies:
// systems below 65 536 bodies): S3 plus one packed add per (row, column pair) -- (eps_j^2 of the two columns, read with the
// positions from a per-wave staging array through v86) + (eps^2 + eps_i^2 of the row, from the pairs (e0, e1) = v[84:85] and
// (e2, e3) = v[88:89], class 0) becomes the first addend of the r^2 chain, EC = v[82:83] (class 1).  9 packed instructions +
// 1 transcendental per pair; its own kernel instantiation (ROWS8 = 4), eps > 0 only.
#define S11_PRE(RXY, RZM, DX, DY, DZ, R, ER, ESEL)                                                               \
    "v_pk_add_f32 " DX ", v[2:3], " RXY " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                      \
    "v_pk_add_f32 " DY ", v[6:7], " RXY " op_sel:[0,1] op_sel_hi:[1,1]" S2_NEG                                      \
    "v_pk_add_f32 " DZ ", v[4:5], " RZM " op_sel:[0,0] op_sel_hi:[1,0]" S2_NEG                                      \
    "v_pk_add_f32 " R ", v[82:83], " ER ESEL "\n\t"                                                                \
    "v_pk_fma_f32 " R ", " DX ", " DX ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DY ", " DY ", " R "\n\t"                                                                  \
    "v_pk_fma_f32 " R ", " DZ ", " DZ ", " R "\n\t"
#define S11_LO " op_sel:[0,0] op_sel_hi:[1,0]"
#define S11_HI " op_sel:[0,1] op_sel_hi:[1,1]"
#define S11_EPS(E0, E1) "ds_read2_b32 v[82:83], v86 offset0:" E0 " offset1:" E1 "\n\t"
#define S11_ADVANCE S2_ADVANCE "v_add_u32_e32 v86, 16, v86\n\t"
#define S11_STEP(NEXT, NEXT_MASSES)                                                                              \
    "s_waitcnt lgkmcnt(7)\n\t" /* positions and eps_j^2 of the column pair have arrived (masses, permutes may be in flight) */ \
    S11_PRE("v[12:13]", "v[14:15]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]", "v[84:85]", S11_LO)          \
    S11_PRE("v[16:17]", "v[18:19]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]", "v[84:85]", S11_HI)          \
    "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"       \
    NB_SYM_GAP                                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t" /* the masses and the column sums have arrived */                                 \
    NB_SYM_PRIO_POST                                                                                             \
    S3_POST("v[14:15]", "v[56:57]", "v[58:59]", "v[60:61]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")      \
    S3_POST("v[18:19]", "v[62:63]", "v[64:65]", "v[66:67]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")      \
    NB_SYM_PRIO_PRE                                                                                              \
    S11_PRE("v[20:21]", "v[22:23]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]", "v[88:89]", S11_LO)          \
    S11_PRE("v[24:25]", "v[26:27]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]", "v[88:89]", S11_HI)          \
    NEXT /* the next step's positions and eps_j^2 */                                                             \
    "v_rsq_f32_e32 v28, v28\n\tv_rsq_f32_e32 v29, v29\n\tv_rsq_f32_e32 v32, v32\n\tv_rsq_f32_e32 v33, v33\n\t"       \
    NB_SYM_GAP                                                                                                   \
    NB_SYM_PRIO_POST                                                                                             \
    S3_POST("v[22:23]", "v[68:69]", "v[70:71]", "v[72:73]", "v[30:31]", "v[34:35]", "v[38:39]", "v[28:29]")      \
    S3_POST("v[26:27]", "v[74:75]", "v[76:77]", "v[78:79]", "v[42:43]", "v[46:47]", "v[50:51]", "v[32:33]")      \
    NB_SYM_PRIO_PRE                                                                                              \
    NEXT_MASSES /* the next step's masses */                                                                     \
    S2_ROTATE
#define S11_GROUP_LOOP                                                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
    S2_READ("0", "32", "128", "160") S11_EPS("0", "32")                                                          \
    S3_MASSES("128", "160")                                                                                      \
    S2_ROTATE /* of zeros: primes the in-order LDS queue */                                                      \
    NB_SYM_PRIO_PRE                                                                                              \
    "s_mov_b32 %[cnt], 8\n"                                                                                      \
    "1:\n\t"                                                                                                     \
    S11_STEP(S2_READ("1", "33", "129", "161") S11_EPS("1", "33"), S3_MASSES("129", "161"))                       \
    S11_STEP(S2_READ("2", "34", "130", "162") S11_EPS("2", "34"), S3_MASSES("130", "162"))                       \
    S11_STEP(S2_READ("3", "35", "131", "163") S11_EPS("3", "35"), S3_MASSES("131", "163"))                       \
    S11_STEP(S11_ADVANCE S2_READ("0", "32", "128", "160") S11_EPS("0", "32"), S3_MASSES("128", "160"))           \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                                            \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                 \
    "s_cbranch_scc1 1b\n\t"                                                                                      \
    "s_setprio 0\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"

// ring distance of the tile (R, C): its slot in the partial-sum arrays
__device__ __forceinline__ int sym_distance(int R, int C, int S)
{
    const int d = C - R;
    return d < 0 ? d + S : d;
}
// P_col[R_local][d - 1][0 .. split_len)
__device__ __forceinline__ float3 *sym_col_slot(float3 *col_partials, int r_local, int d, int S, int L)
{
    return col_partials + ((size_t)r_local * (size_t)(S / 2) + (size_t)(d - 1)) * (size_t)L;
}

struct SymLds {
    float4 *stage;  // this wave's 64-body group
    float *sx, *sy, *sz;
};

template <int W>
__device__ __forceinline__ SymLds sym_lds(float *smem, int L)
{
    SymLds s;
    s.stage = reinterpret_cast<float4 *>(smem) + (threadIdx.x >> 6) * (kSymStageFloatsPerWave / 4);
    s.sx = smem + W * kSymStageFloatsPerWave;
    s.sy = s.sx + L;
    s.sz = s.sy + L;
    return s;
}

// The column group wave `wave` works on in round g: `spacing` groups after wave - 1, all walking the same way, so two
// waves can only meet on a group if one gets `spacing` rounds ahead -- a barrier every `spacing` rounds rules that out
// and fixes the order in which the waves' terms reach each LDS entry.
__device__ __forceinline__ int sym_group(int g, int wave, int spacing, int G)
{
    int cg = g + spacing * wave;
    while (cg >= G)
        cg -= G;
    return cg;
}

// ---- off-diagonal tiles (I < J) ------------------------------------------------------------------------------
typedef float nb_f16 __attribute__((ext_vector_type(16)));
// ROWS8 = 1: equal-mass tiles through the eight-rows-per-lane loop (<= 128 VGPRs: four waves per SIMD), the others through
// the four-row loops; 2: the eight-row loops for both kinds of tile (S9_GROUP_LOOP needs ~150 VGPRs: three waves per SIMD);
// 3: every tile through S10_GROUP_LOOP (per-particle softening, ~160 VGPRs: three waves per SIMD); 4: every tile through the
// four-row S11_GROUP_LOOP (per-particle softening where a split is not a whole multiple of 512 bodies; four waves per SIMD)
#ifndef NB_S8_WAVES
#define NB_S8_WAVES 4  /* waves per SIMD the eight-row equal-mass kernel is allocated for */
#endif
// MODE 0: every strip is one tile (strip_len 1: the launcher's promise) -- the kernel of rounds 1-3.
// MODE 1 (ROWS8 = 2, 3 only): the split is ONE pass of the eight-row loops (split_len = 512 rows x W, the launcher's promise), so
// the rows and their sums stay in registers across the strip's tiles -- one chain per row over all the strip's columns -- and
// the strip takes one kind of loop.
// MODE 2: the strip's tiles are served one after the other and each tile's row sums are added to the strip's in memory (the
// kernels that cannot keep the rows: eps = 0, the four-row loops; A/B arrangements).
template <int W, bool GUARD, int ROWS8 = 0, int MODE = 0>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(ROWS8 == 4 ? 4 : ROWS8 >= 2 ? 3 : ROWS8 == 1 ? NB_S8_WAVES : 5))) void force_sym_kernel(SymArgs a)  // ROWS8 = 0: <= 96 VGPRs
{
    constexpr bool STRIP = MODE == 1;
    static_assert(!STRIP || ((ROWS8 == 2 || ROWS8 == 3) && !GUARD), "rows stay in registers only in the eight-row kernels");
    constexpr int kSymThreads = 64 * W, kSymWaves = W, kSymRowsPerPass = W * 64 * kSymRows;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int L = a.split_len, G = L / 64;
    const SymLds lds = sym_lds<W>(smem, L);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int4 t = a.tiles[blockIdx.x];  // the strip {R, first C, count, slot}: count tiles (R, C), (R, C + 1), ... of one row split
    const int rowbase = t.x * L;
    const int spacing = G >= kSymWaves ? G / kSymWaves : 1;
    const int row_hi = min(a.row_lo + a.row_count, a.n_total);
    const int S = (a.n_total + L - 1) / L;
    // the one mass of the row split's / column split's bodies, NaN where they differ (split_mass_kernel): workgroup-uniform
    const float mass_rows = GUARD ? __builtin_nanf("") : a.split_mass[t.x];
    // the tile at hand (tile_at): its column split, its columns' one mass, and whether it takes the equal-mass loops --
    // only when every pass is full: the lanes of a partial last pass (split lengths that are not a multiple of the rows per pass,
    // e.g. 768 with two waves) carry dummy rows whose zero MASS is what keeps them out of the column sums
    struct Tile {
        int C, colbase;
        float mass_cols, row_scale, col_scale;
        bool uniform;
        bool accumulate;  // MODE 2: the tile's row sums are ADDED to the strip's (tiles after the first)
    };
    auto tile_at = [&](int c, bool accumulate) {
        Tile tl;
        tl.C = c;
        tl.colbase = c * L;
        tl.mass_cols = GUARD ? mass_rows : a.split_mass[c];
        tl.uniform = ROWS8 <= 3 && !GUARD && mass_rows == mass_rows && tl.mass_cols == tl.mass_cols && L % kSymRowsPerPass == 0;  // 4: masses in the loop
        tl.row_scale = tl.uniform ? tl.mass_cols : 1.f;
        tl.col_scale = tl.uniform ? mass_rows : 1.f;
        tl.accumulate = accumulate;
        return tl;
    };

    for (int c = tid; c < L; c += kSymThreads)
        lds.sx[c] = lds.sy[c] = lds.sz[c] = 0.f;
    __syncthreads();
    // the tile's column sums out of LDS -- P_col[R][d - 1][.] -- and LDS cleared for the next tile of the strip
    auto write_columns = [&](const Tile tl) {
        // (a DIAGONAL tile in the list -- strips: the plan puts them into the tile launch, as full squares -- keeps its row side only:
        // every ordered pair of the split has been met, the column side holds the same forces a second time)
        const bool keep = tl.C != t.x;
        float3 *out = sym_col_slot(a.col_partials, t.x - a.row_lo / L, keep ? sym_distance(t.x, tl.C, S) : 1, S, L);
        for (int c = tid; c < L; c += kSymThreads) {
            if (keep && tl.colbase + c < a.n_total)
                out[c] = make_float3(lds.sx[c] * tl.col_scale, lds.sy[c] * tl.col_scale, lds.sz[c] * tl.col_scale);
            if (MODE != 0)
                lds.sx[c] = lds.sy[c] = lds.sz[c] = 0.f;
        }
        if (MODE != 0)
            __syncthreads();
    };

    auto passes = [&](auto variant_tag, const Tile tl) {  // one copy of the loops per inner-loop variant: no merged live ranges
    // 0 general masses, 1 equal-mass tile (one column per step); packed, two columns per step: 2 equal-mass tile, 3 general masses,
    // 4 general masses + per-particle softening
    constexpr int VARIANT = decltype(variant_tag)::value;
    constexpr bool PACKED = VARIANT >= 2;
    constexpr bool PPS4 = VARIANT == 4;
    float *estage4 = lds.sz + L + wave * 128;  // PPS4: the group's eps_j^2, the 64 columns twice
    constexpr bool UNIFORM = VARIANT == 1;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);  // a local of the lambda: a captured one would live in scratch
    for (int pass0 = 0; pass0 < L; pass0 += kSymRowsPerPass) {
        nb_f4 row[kSymRows];
        float ax[kSymRows], ay[kSymRows], az[kSymRows];
        float er[kSymRows];  // PPS4: eps^2 + eps_i^2 of the rows
#pragma unroll
        for (int k = 0; k < kSymRows; ++k) {
            const int r = pass0 + (wave * kSymRows + k) * 64 + lane;
            float4 p = zero4;
            float e = 0.f;
            if (r < L && rowbase + r < row_hi) {
                p = a.pos[rowbase + r];
                if (PPS4)
                    e = a.eps_pp[rowbase + r];
            }
            row[k] = nb_f4{p.x, p.y, p.z, p.w};
            er[k] = __builtin_fmaf(e, e, a.eps2);
            ax[k] = ay[k] = az[k] = 0.f;
        }

        float4 cnext = zero4;  // the next group's column bodies, loaded a group ahead
        float enext = 0.f;     // PPS4: and their softening lengths
        {
            const int gc = tl.colbase + sym_group(0, wave, spacing, G) * 64 + lane;
            if (gc < a.n_total) {
                cnext = a.pos[gc];
                if (PPS4)
                    enext = a.eps_pp[gc];
            }
        }
        nb_f2 ra[kSymRows][3];  // VARIANT 2: row sums per column of the pair
#pragma unroll
        for (int k = 0; k < kSymRows; ++k)
            ra[k][0] = ra[k][1] = ra[k][2] = nb_f2{0.f, 0.f};
        for (int g = 0; g < G; ++g) {
            const int cg = sym_group(g, wave, spacing, G);
            if constexpr (PACKED) {  // the group as x[128], y[128], z[128] (, m[128]): the 64 columns twice
                float *st = reinterpret_cast<float *>(lds.stage);
                st[lane] = st[64 + lane] = cnext.x;
                st[128 + lane] = st[192 + lane] = cnext.y;
                st[256 + lane] = st[320 + lane] = cnext.z;
                if (VARIANT >= 3)
                    st[384 + lane] = st[448 + lane] = cnext.w;
                if (PPS4)
                    estage4[lane] = estage4[64 + lane] = enext * enext;
            } else {
                lds.stage[lane] = cnext;
            }
            if (g + 1 < G) {
                const int gc = tl.colbase + sym_group(g + 1, wave, spacing, G) * 64 + lane;
                cnext = zero4;
                enext = 0.f;
                if (gc < a.n_total) {
                    cnext = a.pos[gc];
                    if (PPS4)
                        enext = a.eps_pp[gc];
                }
            }
            if constexpr (VARIANT == 4) {
                nb_f2 cx = {0.f, 0.f}, cy = cx, cz = cx;  // sums of columns (lane + s) and (lane + s + 32) mod 64, travelling
                unsigned addr = (unsigned)(size_t)lds.stage + 4u * (unsigned)lane, addr_zm = addr + 1024u, cnt;
                unsigned addr_e = (unsigned)(size_t)estage4 + 4u * (unsigned)lane;
                const unsigned next_lane = 4u * ((lane + 1) & 63);
                const nb_f2 e01 = {er[0], er[1]}, e23 = {er[2], er[3]};
                asm volatile(S11_GROUP_LOOP
                             : "+{v[56:57]}"(ra[0][0]), "+{v[58:59]}"(ra[0][1]), "+{v[60:61]}"(ra[0][2]), "+{v[62:63]}"(ra[1][0]),
                               "+{v[64:65]}"(ra[1][1]), "+{v[66:67]}"(ra[1][2]), "+{v[68:69]}"(ra[2][0]), "+{v[70:71]}"(ra[2][1]),
                               "+{v[72:73]}"(ra[2][2]), "+{v[74:75]}"(ra[3][0]), "+{v[76:77]}"(ra[3][1]), "+{v[78:79]}"(ra[3][2]),
                               "+{v[36:37]}"(cx), "+{v[40:41]}"(cy), "+{v[44:45]}"(cz), "+{v1}"(addr_zm), "+{v0}"(addr),
                               "+{v86}"(addr_e), [cnt] "=&s"(cnt)
                             : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]),
                               "{v[84:85]}"(e01), "{v[88:89]}"(e23), "{v53}"(next_lane)
                             : "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v28", "v29", "v30", "v31", "v32", "v33", "v34",
                               "v35", "v38", "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v80", "v81", "v82",
                               "v83", "scc", "memory");
                const int ca = cg * 64 + ((lane + 32) & 63), cb = cg * 64 + lane;  // as in the equal-mass loop below
                lds.sx[ca] -= cx.x;
                lds.sy[ca] -= cy.x;
                lds.sz[ca] -= cz.x;
                lds.sx[cb] -= cx.y;
                lds.sy[cb] -= cy.y;
                lds.sz[cb] -= cz.y;
                if ((g + 1) % spacing == 0)
                    __syncthreads();
                continue;
            }
            if constexpr (VARIANT == 3) {
                nb_f2 cx = {0.f, 0.f}, cy = cx, cz = cx;  // sums of columns (lane + s) and (lane + s + 32) mod 64, travelling
                unsigned addr = (unsigned)(size_t)lds.stage + 4u * (unsigned)lane, addr_zm = addr + 1024u, cnt;
                const unsigned next_lane = 4u * ((lane + 1) & 63);
                const float eps2 = a.eps2;
                asm volatile(S3_GROUP_LOOP
                             : "+{v[56:57]}"(ra[0][0]), "+{v[58:59]}"(ra[0][1]), "+{v[60:61]}"(ra[0][2]), "+{v[62:63]}"(ra[1][0]),
                               "+{v[64:65]}"(ra[1][1]), "+{v[66:67]}"(ra[1][2]), "+{v[68:69]}"(ra[2][0]), "+{v[70:71]}"(ra[2][1]),
                               "+{v[72:73]}"(ra[2][2]), "+{v[74:75]}"(ra[3][0]), "+{v[76:77]}"(ra[3][1]), "+{v[78:79]}"(ra[3][2]),
                               "+{v[36:37]}"(cx), "+{v[40:41]}"(cy), "+{v[44:45]}"(cz), "+{v1}"(addr_zm), "+{v0}"(addr), [cnt] "=&s"(cnt)
                             : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]),
                               "{v8}"(eps2), "{v53}"(next_lane)
                             : "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v28", "v29", "v30", "v31", "v32", "v33", "v34",
                               "v35", "v38", "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v80", "v81",
                               "scc", "memory");
                const int ca = cg * 64 + ((lane + 32) & 63), cb = cg * 64 + lane;  // as in the equal-mass loop below
                lds.sx[ca] -= cx.x;
                lds.sy[ca] -= cy.x;
                lds.sz[ca] -= cz.x;
                lds.sx[cb] -= cx.y;
                lds.sy[cb] -= cy.y;
                lds.sz[cb] -= cz.y;
                if ((g + 1) % spacing == 0)
                    __syncthreads();
                continue;
            }
            if constexpr (VARIANT == 2) {
                nb_f2 cx = {0.f, 0.f}, cy = cx, cz = cx;  // sums of columns (lane + s) and (lane + s + 32) mod 64, travelling
                unsigned addr = (unsigned)(size_t)lds.stage + 4u * (unsigned)lane, addr_z = addr + 1024u, cnt;
                const unsigned next_lane = 4u * ((lane + 1) & 63);
                const nb_f2 epsv = {a.eps2, 0.f};
                asm volatile(S2_GROUP_LOOP
                             : "+{v[56:57]}"(ra[0][0]), "+{v[58:59]}"(ra[0][1]), "+{v[60:61]}"(ra[0][2]), "+{v[62:63]}"(ra[1][0]),
                               "+{v[64:65]}"(ra[1][1]), "+{v[66:67]}"(ra[1][2]), "+{v[68:69]}"(ra[2][0]), "+{v[70:71]}"(ra[2][1]),
                               "+{v[72:73]}"(ra[2][2]), "+{v[74:75]}"(ra[3][0]), "+{v[76:77]}"(ra[3][1]), "+{v[78:79]}"(ra[3][2]),
                               "+{v[36:37]}"(cx), "+{v[40:41]}"(cy), "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), [cnt] "=&s"(cnt)
                             : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]),
                               "{v[8:9]}"(epsv), "{v53}"(next_lane)
                             : "v2", "v3", "v4", "v5", "v6", "v7", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38",
                               "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "scc", "memory");
                // after 32 steps and rotations the lane holds the first-half sum of column lane + 32 and the second-half
                // sum of column lane: two LDS updates in program order (a column's two halves come from different lanes)
                const int ca = cg * 64 + ((lane + 32) & 63), cb = cg * 64 + lane;
                lds.sx[ca] -= cx.x;
                lds.sy[ca] -= cy.x;
                lds.sz[ca] -= cz.x;
                lds.sx[cb] -= cx.y;
                lds.sy[cb] -= cy.y;
                lds.sz[cb] -= cz.y;
                if ((g + 1) % spacing == 0)
                    __syncthreads();
                continue;
            }
            float cx = 0.f, cy = 0.f, cz = 0.f;  // accumulators of column (lane + s) mod 64, travelling
            {
                unsigned off = 16u * (unsigned)lane, addr = 0, cnt;
                const unsigned base = (unsigned)(size_t)lds.stage, mask = 1023u, next_lane = 4u * ((lane + 1) & 63);
                float eps2 = a.eps2;  // in a VGPR: an SGPR source operand costs an fp32 instruction two extra cycles
#define SY_OPERANDS(GUARD_INPUTS)                                                                                     \
                : "+{v53}"(ax[0]), "+{v54}"(ay[0]), "+{v52}"(az[0]), "+{v57}"(ax[1]), "+{v58}"(ay[1]), "+{v56}"(az[1]),      \
                  "+{v61}"(ax[2]), "+{v62}"(ay[2]), "+{v60}"(az[2]), "+{v65}"(ax[3]), "+{v66}"(ay[3]), "+{v64}"(az[3]),      \
                  "+{v45}"(cx), "+{v68}"(cy), "+{v49}"(cz), "+{v1}"(off), "+{v0}"(addr), [cnt] "=&s"(cnt)                  \
                : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]), "{v11}"(eps2),     \
                  GUARD_INPUTS "{v10}"(base), "{v55}"(mask), "{v59}"(next_lane)                                              \
                : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",      \
                  "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v46", "v47", "v48", "v50", "v51", "scc",    \
                  "vcc", "memory"
#define SY_NO_GUARD_INPUTS
                if constexpr (GUARD) {
                    const float tiny = kGuardMin, pinf = __builtin_inff();
#define SY_GUARD_INPUTS "{v69}"(tiny), "{v70}"(pinf),
                    asm volatile(SY_GROUP_LOOP(SY_GUARD, SY_POST) SY_OPERANDS(SY_GUARD_INPUTS));
                } else if constexpr (UNIFORM) {
                    asm volatile(SY_GROUP_LOOP(SY_NOGUARD, SY_POSTU) SY_OPERANDS(SY_NO_GUARD_INPUTS));
                } else {
                    asm volatile(SY_GROUP_LOOP(SY_NOGUARD, SY_POST) SY_OPERANDS(SY_NO_GUARD_INPUTS));
                }
#undef SY_OPERANDS
#undef SY_GUARD_INPUTS
#undef SY_NO_GUARD_INPUTS
            }
            // after 64 rotations lane l holds column l of the group; force on the column body is -m_row * d * inv3
            lds.sx[cg * 64 + lane] -= cx;
            lds.sy[cg * 64 + lane] -= cy;
            lds.sz[cg * 64 + lane] -= cz;
            if ((g + 1) % spacing == 0)
                __syncthreads();
        }
        __syncthreads();

        float3 *const row_out = a.row_partials + (size_t)t.w * a.row_count;  // the strip's row sums: P_row[slot][row]
#pragma unroll
        for (int k = 0; k < kSymRows; ++k) {
            const int r = pass0 + (wave * kSymRows + k) * 64 + lane;
            if (r < L && rowbase + r < row_hi) {
                float3 v = PACKED ? make_float3((ra[k][0].x + ra[k][0].y) * tl.row_scale, (ra[k][1].x + ra[k][1].y) * tl.row_scale,
                                                (ra[k][2].x + ra[k][2].y) * tl.row_scale)
                                  : make_float3(ax[k] * tl.row_scale, ay[k] * tl.row_scale, az[k] * tl.row_scale);
                if (MODE == 2 && tl.accumulate) {  // the same lane wrote the strip's sums so far
                    const float3 o = row_out[rowbase + r - a.row_lo];
                    v = make_float3(o.x + v.x, o.y + v.y, o.z + v.z);
                }
                row_out[rowbase + r - a.row_lo] = v;
            }
        }
    }
    };
    // eight rows per lane (S8_GROUP_LOOP: equal-mass tiles; S9_GROUP_LOOP: arbitrary masses): a wave owns 512 rows of a pass
    // STRIP: the strip's tiles inside (the rows and their sums stay in registers); otherwise the lambda serves the tile at hand.
    auto passes8 = [&](auto general_tag, const Tile tl) {
        constexpr bool own_strip = STRIP;
        Tile cur = tl;  // the tile at hand (STRIP: it advances along the strip; the loop and the scales are the strip's)
        // 0: equal-mass tile (S8), 1: arbitrary masses (S9), 2: arbitrary masses + per-particle softening (S10), 3: equal-mass tile +
        // per-particle softening (S12)
        constexpr int VARIANT8 = decltype(general_tag)::value;
        constexpr bool GENERAL = VARIANT8 == 1 || VARIANT8 == 2, PPS8 = VARIANT8 >= 2;
        float *estage8 = lds.sz + L + wave * 128;  // PPS8: the group's eps_j^2, the 64 columns twice
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        // Which of the 64 bodies of a row block / column group a lane holds: 4 (lane mod 16) + lane / 16, so that the four
        // lanes one ALU lane serves in consecutive cycles (l, l + 16, l + 32, l + 48) hold four CONSECUTIVE bodies -- with the
        // bodies stored along a space-filling curve the operands of consecutive cycles then differ in few bits (power, hence
        // clock).  Any assignment is correct; this one is part of the summation order.
        const int sl = 4 * (lane & 15) + (lane >> 4);
        for (int pass0 = 0; pass0 < L; pass0 += kSymThreads * 8) {
            float4 p[8];
            float er[8];  // PPS8: eps^2 + eps_i^2 of the rows
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = pass0 + (wave * 8 + k) * 64 + sl;
                p[k] = zero4;
                float e = 0.f;
                if (r < L && rowbase + r < row_hi) {
                    p[k] = a.pos[rowbase + r];
                    if (PPS8)
                        e = a.eps_pp[rowbase + r];
                }
                er[k] = __builtin_fmaf(e, e, a.eps2);
            }
            const nb_f2 e01 = {er[0], er[1]}, e23 = {er[2], er[3]}, e45 = {er[4], er[5]}, e67 = {er[6], er[7]};
            const nb_f16 rows = {p[0].x, p[0].y, p[0].z, p[1].z, p[1].x, p[1].y, p[2].z, p[3].z,
                                 p[2].x, p[2].y, p[4].z, p[5].z, p[3].x, p[3].y, p[6].z, p[7].z};
            const nb_f2 xy4 = {p[4].x, p[4].y}, xy5 = {p[5].x, p[5].y}, xy6 = {p[6].x, p[6].y}, xy7 = {p[7].x, p[7].y};
            const nb_f2 m01 = {p[0].w, p[1].w}, m23 = {p[2].w, p[3].w}, m45 = {p[4].w, p[5].w}, m67 = {p[6].w, p[7].w};  // GENERAL
            nb_f16 ra0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
            for (int ti = 0, n_tiles_here = own_strip ? t.z : 1; ti < n_tiles_here; ++ti) {
            if (own_strip) {  // the strip's next tile: same rows, same loop (the strip takes ONE kind of loop, see below)
                cur.C = t.y + ti;
                cur.colbase = cur.C * L;
            }
            float4 cnext = zero4;  // the next group's column bodies, loaded a group ahead
            float enext = 0.f;     // PPS8: and their softening lengths
            {
                const int gc = cur.colbase + sym_group(0, wave, spacing, G) * 64 + sl;
                if (gc < a.n_total) {
                    cnext = a.pos[gc];
                    if (PPS8)
                        enext = a.eps_pp[gc];
                }
            }
            for (int g = 0; g < G; ++g) {
                const int cg = sym_group(g, wave, spacing, G);
                float *st = reinterpret_cast<float *>(lds.stage);
                st[lane] = st[64 + lane] = cnext.x;
                st[128 + lane] = st[192 + lane] = cnext.y;
                st[256 + lane] = st[320 + lane] = cnext.z;
                if (GENERAL)
                    st[384 + lane] = st[448 + lane] = cnext.w;
                if (PPS8)
                    estage8[lane] = estage8[64 + lane] = enext * enext;
                if (g + 1 < G) {
                    const int gc = cur.colbase + sym_group(g + 1, wave, spacing, G) * 64 + sl;
                    cnext = zero4;
                    enext = 0.f;
                    if (gc < a.n_total) {
                        cnext = a.pos[gc];
                        if (PPS8)
                            enext = a.eps_pp[gc];
                    }
                }
                nb_f2 cx = {0.f, 0.f}, cy = cx, cz = cx;  // sums of columns (lane + s) and (lane + s + 32) mod 64, travelling
                unsigned addr = (unsigned)(size_t)lds.stage + 4u * (unsigned)lane, addr_z = addr + 1024u, cnt;
                const unsigned next_lane = 4u * ((lane + 1) & 63);
                const nb_f2 epsv = {a.eps2, 0.f};
                if constexpr (VARIANT8 == 3) {
                    unsigned addr_e = (unsigned)(size_t)estage8 + 4u * (unsigned)lane;
                    asm volatile(S12_GROUP_LOOP
                                 : "+{v[64:79]}"(ra0), "+{v[80:95]}"(ra1), "+{v[96:111]}"(ra2), "+{v[36:37]}"(cx), "+{v[40:41]}"(cy),
                                   "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), "+{v146}"(addr_e), [cnt] "=&s"(cnt)
                                 : "{v[12:27]}"(rows), "{v[48:49]}"(xy4), "{v[52:53]}"(xy5), "{v[56:57]}"(xy6), "{v[60:61]}"(xy7),
                                   "{v[8:9]}"(epsv), "{v10}"(next_lane), "{v[132:133]}"(e01), "{v[136:137]}"(e23),
                                   "{v[140:141]}"(e45), "{v[144:145]}"(e67)
                                 : "v2", "v3", "v4", "v5", "v6", "v7", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38",
                                   "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v58", "v59", "v134", "v135",
                                   "scc", "memory");
                } else if constexpr (PPS8) {
                    unsigned addr_e = (unsigned)(size_t)estage8 + 4u * (unsigned)lane;
                    asm volatile(S10_GROUP_LOOP
                                 : "+{v[64:79]}"(ra0), "+{v[80:95]}"(ra1), "+{v[96:111]}"(ra2), "+{v[36:37]}"(cx), "+{v[40:41]}"(cy),
                                   "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), "+{v146}"(addr_e), [cnt] "=&s"(cnt)
                                 : "{v[12:27]}"(rows), "{v[48:49]}"(xy4), "{v[52:53]}"(xy5), "{v[56:57]}"(xy6), "{v[60:61]}"(xy7),
                                   "{v[8:9]}"(epsv), "{v10}"(next_lane), "{v[118:119]}"(m01), "{v[122:123]}"(m23),
                                   "{v[126:127]}"(m45), "{v[130:131]}"(m67), "{v[132:133]}"(e01), "{v[136:137]}"(e23),
                                   "{v[140:141]}"(e45), "{v[144:145]}"(e67)
                                 : "v2", "v3", "v4", "v5", "v6", "v7", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38",
                                   "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v58", "v59", "v112", "v113",
                                   "v114", "v115", "v116", "v117", "v134", "v135", "scc", "memory");
                } else if constexpr (GENERAL) {
                    asm volatile(S9_GROUP_LOOP
                                 : "+{v[64:79]}"(ra0), "+{v[80:95]}"(ra1), "+{v[96:111]}"(ra2), "+{v[36:37]}"(cx), "+{v[40:41]}"(cy),
                                   "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), [cnt] "=&s"(cnt)
                                 : "{v[12:27]}"(rows), "{v[48:49]}"(xy4), "{v[52:53]}"(xy5), "{v[56:57]}"(xy6), "{v[60:61]}"(xy7),
                                   "{v[8:9]}"(epsv), "{v10}"(next_lane), "{v[118:119]}"(m01), "{v[122:123]}"(m23),
                                   "{v[126:127]}"(m45), "{v[130:131]}"(m67)
                                 : "v2", "v3", "v4", "v5", "v6", "v7", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38",
                                   "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v58", "v59", "v112", "v113",
                                   "v114", "v115", "v116", "v117", "scc", "memory");
                } else {
                    asm volatile(S8_GROUP_LOOP
                                 : "+{v[64:79]}"(ra0), "+{v[80:95]}"(ra1), "+{v[96:111]}"(ra2), "+{v[36:37]}"(cx), "+{v[40:41]}"(cy),
                                   "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), [cnt] "=&s"(cnt)
                                 : "{v[12:27]}"(rows), "{v[48:49]}"(xy4), "{v[52:53]}"(xy5), "{v[56:57]}"(xy6), "{v[60:61]}"(xy7),
                                   "{v[8:9]}"(epsv), "{v10}"(next_lane)
                                 : "v2", "v3", "v4", "v5", "v6", "v7", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38",
                                   "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v58", "v59", "scc", "memory");
                }
                // slot (lane + 32) mod 64 and slot lane of the stage: the bodies 4 (slot mod 16) + slot / 16 of the group
                const int ca = cg * 64 + 4 * (lane & 15) + (((lane >> 4) + 2) & 3), cb = cg * 64 + sl;
                lds.sx[ca] -= cx.x;
                lds.sy[ca] -= cy.x;
                lds.sz[ca] -= cz.x;
                lds.sx[cb] -= cx.y;
                lds.sy[cb] -= cy.y;
                lds.sz[cb] -= cz.y;
                if ((g + 1) % spacing == 0)
                    __syncthreads();
            }
            __syncthreads();
            if (own_strip)
                write_columns(cur);
            }
            float3 *const row_out = a.row_partials + (size_t)t.w * a.row_count;  // the strip's row sums: P_row[slot][row]
            float sum[48];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sum[i] = ra0[i];
                sum[16 + i] = ra1[i];
                sum[32 + i] = ra2[i];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = pass0 + (wave * 8 + k) * 64 + sl;
                if (r < L && rowbase + r < row_hi) {
                    float3 v = make_float3((sum[6 * k] + sum[6 * k + 1]) * cur.row_scale, (sum[6 * k + 2] + sum[6 * k + 3]) * cur.row_scale,
                                           (sum[6 * k + 4] + sum[6 * k + 5]) * cur.row_scale);
                    if (MODE == 2 && tl.accumulate) {
                        const float3 o = row_out[rowbase + r - a.row_lo];
                        v = make_float3(o.x + v.x, o.y + v.y, o.z + v.z);
                    }
                    row_out[rowbase + r - a.row_lo] = v;
                }
            }
        }
    };
    // The strip.  Where the split is one pass of the eight-row loops (L = 512 rows x waves: 2048-body splits with four waves, the
    // splits of N >= 2^20) the rows' sums stay in registers across the strip's tiles and the strip takes ONE loop: the equal-mass
    // one when the row split carries one mass and every column split of the strip carries one and the same mass (every strip of
    // the benchmark's sphere), else the loop with the masses in it.  Elsewhere the tiles are served one after the other and each
    // tile's row sums are added to the strip's (accumulate): a function of the data and the split boundaries either way.
    if constexpr (STRIP) {
        Tile tl = tile_at(t.y, false);
        bool strip_uniform = tl.uniform;
        for (int ti = 1; ti < t.z; ++ti)
            strip_uniform = strip_uniform && a.split_mass[t.y + ti] == tl.mass_cols;  // NaN (mixed masses) compares unequal
        tl.uniform = strip_uniform;
        tl.row_scale = strip_uniform ? tl.mass_cols : 1.f;
        tl.col_scale = strip_uniform ? mass_rows : 1.f;
        if constexpr (ROWS8 == 3) {
            if (tl.uniform)
                passes8(std::integral_constant<int, 3>{}, tl);
            else
                passes8(std::integral_constant<int, 2>{}, tl);
        } else if (tl.uniform) {
            passes8(std::integral_constant<int, 0>{}, tl);
        } else {
            passes8(std::integral_constant<int, 1>{}, tl);
        }
        return;
    }
    for (int ti = 0; ti < (MODE == 2 ? t.z : 1); ++ti) {
        const Tile tl = tile_at(t.y + ti, ti > 0);
        bool done = false;
        if constexpr (ROWS8 != 0 && !GUARD) {
            if (L % (kSymThreads * 8) == 0) {  // whole passes of 512 rows per wave (dummy rows would need their zero mass: see above)
                if constexpr (ROWS8 == 3) {
                    if (tl.uniform)
                        passes8(std::integral_constant<int, 3>{}, tl);
                    else
                        passes8(std::integral_constant<int, 2>{}, tl);
                    done = true;
                } else if (tl.uniform) {
                    passes8(std::integral_constant<int, 0>{}, tl);
                    done = true;
                } else if constexpr (ROWS8 == 2) {
                    passes8(std::integral_constant<int, 1>{}, tl);
                    done = true;
                }
            }
        }
        if constexpr (ROWS8 == 4 && !GUARD) {  // per-particle softening, four rows per lane: the only loop of this instantiation
            passes(std::integral_constant<int, 4>{}, tl);
        } else {
            if (done)
                ;
            else if (tl.uniform && a.packed)
                passes(std::integral_constant<int, 2>{}, tl);
            else if (tl.uniform)
                passes(std::integral_constant<int, 1>{}, tl);
            else if (!GUARD && a.packed)
                passes(std::integral_constant<int, 3>{}, tl);
            else
                passes(std::integral_constant<int, 0>{}, tl);
        }
        write_columns(tl);
    }
}

// ---- small systems (256-body splits): a tile by FOUR waves, one 64-column group each (round 4) ----------------------------
// With one wave per 256 x 256 tile (force_sym_kernel<1, ...>) a pass of the reference's own size -- 80 splits, 3160 tiles -- is
// 3.09 tiles of ~18 us per SIMD: the SIMDs with three wait for those with four or five, and the 80 diagonal tiles, one wave of
// the compiler-scheduled kernel each, take 62 us alone and 128 us beside the tiles (profiles/r04_pair_once_small_n.txt).  Here the
// unit is a quarter of that: every wave of the workgroup holds ALL 256 rows of the row split (four per lane, as in the loops
// above) and meets ONE 64-column group of the column split -- 32 steps of S2_GROUP_LOOP / S3_GROUP_LOOP.  A column's sum is complete
// inside its wave (no LDS array: the two half sums meet through one permute and go to P_col[R][d - 1][.] as before); a row's sum
// is the four waves' sums added in wave order through LDS -- one row-side partial sum per tile, the layout of the other kernels.
// The equal-mass decision is per wave and made here (rows: the 256 bodies of the split, missing ones counting as zero-mass;
// columns: the wave's 64), so no flag kernel runs in front; a function of the data and the split boundaries only, like
// everything else about the sums.  The DIAGONAL tiles go through the same loop as full squares -- every ordered pair inside the
// split, the self pair contributing exactly 0 (dx = 0 against a finite inv3; with eps = 0 through the guard) -- and keep the row side only: twice the
// pair evaluations of a triangle on 80 of 3240 tiles, at the hand-scheduled rate, in the same launch (workgroups >= n_tiles).
// LOOP 0: eps > 0 (S2_GROUP_LOOP where the wave's rows carry one mass and its columns one mass, else S3_GROUP_LOOP);
// 1: per-particle softening with eps > 0 (S11_GROUP_LOOP); 2: eps = 0 (the guarded one-column loop: a zero-distance pair -- the self
// pair of a diagonal tile among them -- contributes exactly 0).
// NH = 2: 512-body splits (65 536 <= N < 131 072) by EIGHT waves -- wave (h, q) holds the 256 rows of half h of the row split and
// meets quarter q of the column split, two 64-column groups one after the other; a row's sum is the sums of the four waves of
// its half in wave order, a column's sum the sum of half 0's wave and half 1's, in that order, through LDS.
template <int LOOP, int NH>
__global__ __launch_bounds__(256 * NH) __attribute__((amdgpu_waves_per_eu(NH == 1 ? 5 : 4))) void force_sym_quarter_kernel(SymArgs a)
{
    constexpr int L = 256 * NH, NW = 4 * NH, GPW = NH;  // split length, waves, 64-column groups per wave
    __shared__ __attribute__((aligned(1024))) float stage_all[NW * kSymStageFloatsPerWave];
    __shared__ float estage_all[LOOP == 1 ? NW * 128 : 1];
    __shared__ float xch[NW][3][256];                           // the waves' row sums
    __shared__ float colx[NH == 2 ? 4 : 1][GPW][3][64];         // NH = 2: half 1's column sums on their way to half 0's wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = NH == 1 ? 0 : wave >> 2, q = wave & 3;
    const bool diag = (int)blockIdx.x >= a.n_tiles;
    int R, C, slot;
    if (diag) {
        R = C = a.diag_tiles[blockIdx.x - a.n_tiles].x;
        slot = 0;
    } else {
        const int4 t = a.tiles[blockIdx.x];
        R = t.x, C = t.y, slot = t.w;
    }
    const int rowbase = R * L + h * 256, colbase = C * L;
    const int row_hi = min(a.row_lo + a.row_count, a.n_total);
    const int S = (a.n_total + L - 1) / L;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    nb_f4 row[kSymRows];
    float er[kSymRows];  // LOOP 1: eps^2 + eps_i^2 of the rows
    bool same = true;
    // the one mass of the wave's rows and of its columns, if there is one: missing bodies count as zero-mass ones (split_mass_kernel)
    const unsigned mref = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, rowbase < row_hi ? a.pos[rowbase].w : 0.f));
#pragma unroll
    for (int k = 0; k < kSymRows; ++k) {
        const int r = rowbase + k * 64 + lane;
        float4 p = zero4;
        float e = 0.f;
        if (r < row_hi) {
            p = a.pos[r];
            if (LOOP == 1)
                e = a.eps_pp[r];
        }
        row[k] = nb_f4{p.x, p.y, p.z, p.w};
        er[k] = __builtin_fmaf(e, e, a.eps2);
        same &= __builtin_bit_cast(unsigned, p.w) == mref;
    }
    float4 c[GPW];
    float ec[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int gc = colbase + (q * GPW + i) * 64 + lane;
        c[i] = zero4;
        ec[i] = 0.f;
        if (gc < a.n_total) {
            c[i] = a.pos[gc];
            if (LOOP == 1)
                ec[i] = a.eps_pp[gc];
        }
    }
    const unsigned cref = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, c[0].w));
#pragma unroll
    for (int i = 0; i < GPW; ++i)
        same &= __builtin_bit_cast(unsigned, c[i].w) == cref;
    const float mass_rows = __builtin_bit_cast(float, mref), mass_cols = __builtin_bit_cast(float, cref);
    const bool uniform = LOOP == 0 && a.equal_mass_path && __all(same) && fabsf(mass_rows) <= 3.4e38f && fabsf(mass_cols) <= 3.4e38f;
    const float row_scale = uniform ? mass_cols : 1.f, col_scale = uniform ? mass_rows : 1.f;

    float *st = stage_all + wave * kSymStageFloatsPerWave;
    float *estage = estage_all + (LOOP == 1 ? wave * 128 : 0);
    float3 csum[GPW];  // column `lane` of the wave's groups over its 256 rows
    // the rows' sums over the wave's columns: per column of the pair in the packed loops (added when the groups are through)
    nb_f2 ra[kSymRows][3];
    float ax[kSymRows], ay[kSymRows], az[kSymRows];
#pragma unroll
    for (int k = 0; k < kSymRows; ++k) {
        ra[k][0] = ra[k][1] = ra[k][2] = nb_f2{0.f, 0.f};
        ax[k] = ay[k] = az[k] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        if constexpr (LOOP == 2) {
            reinterpret_cast<float4 *>(st)[lane] = c[i];
            float cx = 0.f, cy = 0.f, cz = 0.f;  // accumulators of column (lane + s) mod 64, travelling
            unsigned off = 16u * (unsigned)lane, addr = 0, cnt;
            const unsigned base = (unsigned)(size_t)st, mask = 1023u, next_lane = 4u * ((lane + 1) & 63);
            float eps2 = a.eps2;
            const float tiny = kGuardMin, pinf = __builtin_inff();
            asm volatile(SY_GROUP_LOOP(SY_GUARD, SY_POST)
                         : "+{v53}"(ax[0]), "+{v54}"(ay[0]), "+{v52}"(az[0]), "+{v57}"(ax[1]), "+{v58}"(ay[1]), "+{v56}"(az[1]),
                           "+{v61}"(ax[2]), "+{v62}"(ay[2]), "+{v60}"(az[2]), "+{v65}"(ax[3]), "+{v66}"(ay[3]), "+{v64}"(az[3]),
                           "+{v45}"(cx), "+{v68}"(cy), "+{v49}"(cz), "+{v1}"(off), "+{v0}"(addr), [cnt] "=&s"(cnt)
                         : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]), "{v11}"(eps2),
                           "{v69}"(tiny), "{v70}"(pinf), "{v10}"(base), "{v55}"(mask), "{v59}"(next_lane)
                         : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                           "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v46", "v47", "v48", "v50", "v51", "scc",
                           "vcc", "memory");
            // after 64 rotations lane l holds column l of the group; force on the column body is -m_row * d * inv3
            csum[i] = make_float3(0.f - cx, 0.f - cy, 0.f - cz);
        } else {
            // the group as x[128], y[128], z[128], m[128]: the 64 columns twice
            st[lane] = st[64 + lane] = c[i].x;
            st[128 + lane] = st[192 + lane] = c[i].y;
            st[256 + lane] = st[320 + lane] = c[i].z;
            st[384 + lane] = st[448 + lane] = c[i].w;
            if (LOOP == 1)
                estage[lane] = estage[64 + lane] = ec[i] * ec[i];
            nb_f2 cx = {0.f, 0.f}, cy = cx, cz = cx;  // sums of columns (lane + s) and (lane + s + 32) mod 64, travelling
            unsigned addr = (unsigned)(size_t)st + 4u * (unsigned)lane, addr_z = addr + 1024u, cnt;
            const unsigned next_lane = 4u * ((lane + 1) & 63);
            if constexpr (LOOP == 1) {
                unsigned addr_e = (unsigned)(size_t)estage + 4u * (unsigned)lane;
                const nb_f2 e01 = {er[0], er[1]}, e23 = {er[2], er[3]};
                asm volatile(S11_GROUP_LOOP
                             : "+{v[56:57]}"(ra[0][0]), "+{v[58:59]}"(ra[0][1]), "+{v[60:61]}"(ra[0][2]), "+{v[62:63]}"(ra[1][0]),
                               "+{v[64:65]}"(ra[1][1]), "+{v[66:67]}"(ra[1][2]), "+{v[68:69]}"(ra[2][0]), "+{v[70:71]}"(ra[2][1]),
                               "+{v[72:73]}"(ra[2][2]), "+{v[74:75]}"(ra[3][0]), "+{v[76:77]}"(ra[3][1]), "+{v[78:79]}"(ra[3][2]),
                               "+{v[36:37]}"(cx), "+{v[40:41]}"(cy), "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr),
                               "+{v86}"(addr_e), [cnt] "=&s"(cnt)
                             : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]),
                               "{v[84:85]}"(e01), "{v[88:89]}"(e23), "{v53}"(next_lane)
                             : "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v28", "v29", "v30", "v31", "v32", "v33", "v34",
                               "v35", "v38", "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v80", "v81", "v82",
                               "v83", "scc", "memory");
            } else if (uniform) {
                const nb_f2 epsv = {a.eps2, 0.f};
                asm volatile(S2_GROUP_LOOP
                             : "+{v[56:57]}"(ra[0][0]), "+{v[58:59]}"(ra[0][1]), "+{v[60:61]}"(ra[0][2]), "+{v[62:63]}"(ra[1][0]),
                               "+{v[64:65]}"(ra[1][1]), "+{v[66:67]}"(ra[1][2]), "+{v[68:69]}"(ra[2][0]), "+{v[70:71]}"(ra[2][1]),
                               "+{v[72:73]}"(ra[2][2]), "+{v[74:75]}"(ra[3][0]), "+{v[76:77]}"(ra[3][1]), "+{v[78:79]}"(ra[3][2]),
                               "+{v[36:37]}"(cx), "+{v[40:41]}"(cy), "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), [cnt] "=&s"(cnt)
                             : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]),
                               "{v[8:9]}"(epsv), "{v53}"(next_lane)
                             : "v2", "v3", "v4", "v5", "v6", "v7", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38",
                               "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "scc", "memory");
            } else {
                const float eps2 = a.eps2;
                asm volatile(S3_GROUP_LOOP
                             : "+{v[56:57]}"(ra[0][0]), "+{v[58:59]}"(ra[0][1]), "+{v[60:61]}"(ra[0][2]), "+{v[62:63]}"(ra[1][0]),
                               "+{v[64:65]}"(ra[1][1]), "+{v[66:67]}"(ra[1][2]), "+{v[68:69]}"(ra[2][0]), "+{v[70:71]}"(ra[2][1]),
                               "+{v[72:73]}"(ra[2][2]), "+{v[74:75]}"(ra[3][0]), "+{v[76:77]}"(ra[3][1]), "+{v[78:79]}"(ra[3][2]),
                               "+{v[36:37]}"(cx), "+{v[40:41]}"(cy), "+{v[44:45]}"(cz), "+{v1}"(addr_z), "+{v0}"(addr), [cnt] "=&s"(cnt)
                             : "{v[12:15]}"(row[0]), "{v[16:19]}"(row[1]), "{v[20:23]}"(row[2]), "{v[24:27]}"(row[3]),
                               "{v8}"(eps2), "{v53}"(next_lane)
                             : "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v28", "v29", "v30", "v31", "v32", "v33", "v34",
                               "v35", "v38", "v39", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55", "v80", "v81",
                               "scc", "memory");
            }
            // after 32 steps and rotations the lane holds the first-half sum of column lane + 32 and the second-half sum of
            // column lane: the two halves of column `lane` meet through one permute
            const int from = 4 * ((lane + 32) & 63);
            const float hx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, cx.x)));
            const float hy = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, cy.x)));
            const float hz = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, cz.x)));
            csum[i] = make_float3((0.f - hx) - cx.y, (0.f - hy) - cy.y, (0.f - hz) - cz.y);
        }
        csum[i] = make_float3(csum[i].x * col_scale, csum[i].y * col_scale, csum[i].z * col_scale);
        if (NH == 2 && h == 1 && !diag) {
            colx[q][i][0][lane] = csum[i].x;
            colx[q][i][1][lane] = csum[i].y;
            colx[q][i][2][lane] = csum[i].z;
        }
    }
#pragma unroll
    for (int k = 0; k < kSymRows; ++k) {
        const float3 v = LOOP == 2 ? make_float3(ax[k], ay[k], az[k])
                                   : make_float3(ra[k][0].x + ra[k][0].y, ra[k][1].x + ra[k][1].y, ra[k][2].x + ra[k][2].y);
        xch[wave][0][k * 64 + lane] = v.x * row_scale;
        xch[wave][1][k * 64 + lane] = v.y * row_scale;
        xch[wave][2][k * 64 + lane] = v.z * row_scale;
    }
    if (NH == 2)
        __syncthreads();
    if (!diag && h == 0) {
        float3 *out = sym_col_slot(a.col_partials, R - a.row_lo / L, sym_distance(R, C, S), S, L);
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int cc = (q * GPW + i) * 64 + lane;
            if (colbase + cc < a.n_total) {
                float3 v = csum[i];
                if (NH == 2)
                    v = make_float3(v.x + colx[q][i][0][lane], v.y + colx[q][i][1][lane], v.z + colx[q][i][2][lane]);
                out[cc] = v;
            }
        }
    }
    if (NH == 1)
        __syncthreads();
    // thread t adds the four waves' sums of row t of its half, in wave order
    const int hh = NH == 1 ? 0 : tid >> 8, rr = tid & 255;
    const int r = R * L + hh * 256 + rr;
    if (r < row_hi) {
        float3 *const row_out = a.row_partials + (size_t)slot * a.row_count;  // P_row[slot][row]
        const int w0 = hh * 4;
        row_out[r - a.row_lo] = make_float3(((xch[w0][0][rr] + xch[w0 + 1][0][rr]) + xch[w0 + 2][0][rr]) + xch[w0 + 3][0][rr],
                                            ((xch[w0][1][rr] + xch[w0 + 1][1][rr]) + xch[w0 + 2][1][rr]) + xch[w0 + 3][1][rr],
                                            ((xch[w0][2][rr] + xch[w0 + 1][2][rr]) + xch[w0 + 2][2][rr]) + xch[w0 + 3][2][rr]);
    }
}

// split_mass[s] = the mass every body of split s has, or NaN when they differ (a ragged last split counts its missing
// bodies as zero-mass ones).  O(N), launched in front of the tiles: the masses live in the caller's buffer and may
// change between steps.  Speed only -- a tile takes the same path whatever the sharding, the flag being a function of
// the data and of the split boundaries.
__global__ __launch_bounds__(kTile) void split_mass_kernel(const float4 *pos, float *split_mass, int n_total, int split_len,
                                                           int enabled)
{
    __shared__ int differs;
    if (!enabled) {  // nbody_set_equal_mass_path(ctx, 0): every tile takes the general path
        if (threadIdx.x == 0)
            split_mass[blockIdx.x] = __builtin_nanf("");
        return;
    }
    const int base = blockIdx.x * split_len;
    const float m0 = pos[base].w;  // base < n_total: the grid has ceil(n_total / split_len) workgroups
    if (threadIdx.x == 0)
        differs = 0;
    __syncthreads();
    bool bad = false;
    for (int c = threadIdx.x; c < split_len; c += kTile) {
        const float m = base + c < n_total ? pos[base + c].w : 0.f;
        bad |= __builtin_bit_cast(unsigned, m) != __builtin_bit_cast(unsigned, m0);
    }
    if (bad)
        differs = 1;
    __syncthreads();
    if (threadIdx.x == 0)
        split_mass[blockIdx.x] = differs || !(fabsf(m0) <= 3.4e38f) ? __builtin_nanf("") : m0;
}

hipError_t launch_split_mass(const float4 *pos, float *split_mass, int n_total, int split_len, bool enabled, hipStream_t stream)
{
    if (n_total <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(split_mass_kernel, dim3((n_total + split_len - 1) / split_len), dim3(kTile), 0, stream, pos, split_mass,
                       n_total, split_len, enabled ? 1 : 0);
    return hipGetLastError();
}

// ---- the compiler-scheduled tile kernel: diagonal tiles, and every tile under per-particle softening ---------------
// DIAG: rows and columns are the same bodies; every (row, column) combination is visited and the pairs with row index
// < column index are kept (drops the self pair too); both sides end in P_row[B].  PPS: eps_ij^2 = eps^2 + eps_i^2 +
// eps_j^2 (symmetric in the pair, so it fits the pair-once scheme): one extra add per pair and a 64-float LDS stage
// of the group's eps_j^2 per wave.  Same data flow and summation order as the hand-scheduled kernel above.
template <int W, bool DIAG, bool GUARD, bool PPS>
__global__ __launch_bounds__(64 * W) void force_sym_general_kernel(SymArgs a)
{
    constexpr int kSymThreads = 64 * W, kSymWaves = W, kSymRowsPerPass = W * 64 * kSymRows;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int L = a.split_len, G = L / 64;
    const SymLds lds = sym_lds<W>(smem, L);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *estage = lds.sz + L + wave * 128;  // 128 floats per wave are reserved (the eight-row loop stages the group twice)
    // DIAG: one diagonal tile (B, B); else a strip {R, first C, count, slot}, its tiles served one after the other and the row
    // sums of the tiles after the first ADDED to the strip's (the same lanes wrote them)
    const int4 t = DIAG ? make_int4(a.diag_tiles[blockIdx.x].x, a.diag_tiles[blockIdx.x].y, 1, 0) : a.tiles[blockIdx.x];
    const int rowbase = t.x * L;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int spacing = G >= kSymWaves ? G / kSymWaves : 1;
    const int row_hi = min(a.row_lo + a.row_count, a.n_total);
    const int S = (a.n_total + L - 1) / L;
    float eps2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(eps2) : "s"(a.eps2));

    for (int c = tid; c < L; c += kSymThreads)
        lds.sx[c] = lds.sy[c] = lds.sz[c] = 0.f;
    __syncthreads();

    for (int ti = 0; ti < t.z; ++ti) {
    const int C = t.y + ti, colbase = C * L;
    for (int pass0 = 0; pass0 < L; pass0 += kSymRowsPerPass) {
        float x[kSymRows], y[kSymRows], z[kSymRows], m[kSymRows], ax[kSymRows], ay[kSymRows], az[kSymRows], er[kSymRows];
        int rl[kSymRows];  // row index inside the split, or -1
#pragma unroll
        for (int k = 0; k < kSymRows; ++k) {
            const int r = pass0 + (wave * kSymRows + k) * 64 + lane;
            float4 p = zero4;
            float e = 0.f;
            rl[k] = -1;
            if (r < L && rowbase + r < row_hi) {
                p = a.pos[rowbase + r];
                if (PPS)
                    e = a.eps_pp[rowbase + r];
                rl[k] = r;
            }
            x[k] = p.x; y[k] = p.y; z[k] = p.z; m[k] = p.w;
            er[k] = __builtin_fmaf(e, e, eps2);  // eps^2 + eps_i^2
            ax[k] = ay[k] = az[k] = 0.f;
        }
        for (int g = 0; g < G; ++g) {
            const int cg = sym_group(g, wave, spacing, G);
            const int gc = colbase + cg * 64 + lane;
            float4 c = zero4;
            float ec = 0.f;
            if (gc < a.n_total) {
                c = a.pos[gc];
                if (PPS)
                    ec = a.eps_pp[gc];
            }
            lds.stage[lane] = c;
            if (PPS)
                estage[lane] = ec * ec;
            float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll 4
            for (int s = 0; s < 64; ++s) {
                const int cl = (lane + s) & 63;
                const float4 pj = lds.stage[cl];
                const float ej = PPS ? estage[cl] : 0.f;
                const int col = cg * 64 + cl;
#pragma unroll
                for (int k = 0; k < kSymRows; ++k) {
                    const float dx = pj.x - x[k], dy = pj.y - y[k], dz = pj.z - z[k];
                    float r2 = __builtin_fmaf(dx, dx, PPS ? er[k] + ej : eps2);
                    r2 = __builtin_fmaf(dy, dy, r2);
                    r2 = __builtin_fmaf(dz, dz, r2);
                    if (GUARD)
                        r2 = guard_r2(r2);
                    const float inv = __builtin_amdgcn_rsqf(r2);
                    float inv3 = inv * (inv * inv);
                    if (DIAG)
                        inv3 = (rl[k] >= 0 && rl[k] < col) ? inv3 : 0.f;
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
            lds.sx[cg * 64 + lane] -= cx;
            lds.sy[cg * 64 + lane] -= cy;
            lds.sz[cg * 64 + lane] -= cz;
            if ((g + 1) % spacing == 0)
                __syncthreads();
        }
        __syncthreads();
        if (DIAG) {  // row sums of this pass join the column sums of the same bodies
#pragma unroll
            for (int k = 0; k < kSymRows; ++k)
                if (rl[k] >= 0) {
                    lds.sx[rl[k]] += ax[k];
                    lds.sy[rl[k]] += ay[k];
                    lds.sz[rl[k]] += az[k];
                }
            __syncthreads();
        } else {
            float3 *out = a.row_partials + (size_t)t.w * a.row_count;  // P_row[slot][row]
#pragma unroll
            for (int k = 0; k < kSymRows; ++k)
                if (rl[k] >= 0) {
                    float3 v = make_float3(ax[k], ay[k], az[k]);
                    if (ti > 0) {
                        const float3 o = out[rowbase + rl[k] - a.row_lo];
                        v = make_float3(o.x + v.x, o.y + v.y, o.z + v.z);
                    }
                    out[rowbase + rl[k] - a.row_lo] = v;
                }
        }
    }

    // DIAG: both sides of the split, P_row[0][b]; else the tile's column sums, P_col[R][d-1][.], and LDS cleared for the next tile
    const bool keep = DIAG || C != t.x;  // a diagonal tile in the TILE list (strips): a full square, the row side only
    float3 *out = DIAG ? a.row_partials + (rowbase - a.row_lo)
                       : sym_col_slot(a.col_partials, t.x - a.row_lo / L, keep ? sym_distance(t.x, C, S) : 1, S, L);
    const int col_hi = DIAG ? row_hi : a.n_total;  // a diagonal tile's columns are the context's own rows
    for (int c = tid; c < L; c += kSymThreads) {
        if (keep && colbase + c < col_hi)
            out[c] = make_float3(lds.sx[c], lds.sy[c], lds.sz[c]);
        lds.sx[c] = lds.sy[c] = lds.sz[c] = 0.f;
    }
    __syncthreads();
    }
}



// waves per tile workgroup: W x 256 rows per pass must not exceed the split
static int sym_waves(int split_len) { return split_len >= 1024 ? 4 : split_len >= 512 ? 2 : 1; }

// the column-group stage, three column-sum arrays, and the per-wave eps_j^2 stage of the per-particle-softening variant
size_t symmetric_lds_bytes(int split_len)
{
    const size_t w = (size_t)sym_waves(split_len);
    return w * kSymStageFloatsPerWave * sizeof(float) + (size_t)split_len * 12 + w * 128 * sizeof(float);
}

template <typename K>
static hipError_t sym_launch(K kernel, int blocks, int waves, size_t lds, const SymArgs &a, hipStream_t stream)
{
    if (blocks <= 0)
        return hipSuccess;
    // the attribute is per function and device: set it once (and again only if a longer split needs more)
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    {
        std::lock_guard<std::mutex> lock(mu);
        size_t &have = granted[{reinterpret_cast<const void *>(kernel), dev}];
        if (have < lds) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds);
            if (e != hipSuccess)
                return e;
            have = lds;
        }
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * waves), lds, stream, a);
    return hipGetLastError();
}

static size_t sym_lds_bytes_for(int waves, int split_len)
{
    return (size_t)waves * kSymStageFloatsPerWave * sizeof(float) + (size_t)split_len * 12 + (size_t)waves * 128 * sizeof(float);
}

template <int W>
static hipError_t sym_launch_tiles(const SymArgs &a, size_t lds, hipStream_t stream)
{
    if constexpr (W == 4) {
        if (a.strip_len > 1) {  // the kernels that cannot keep a strip's rows in registers add its tiles' row sums in memory
            if (a.eps_pp && a.eps2 > 0.f && a.packed >= 2)
                return sym_launch(&force_sym_kernel<4, false, 4, 2>, a.n_tiles, 4, lds, a, stream);
            if (a.eps_pp)
                return a.eps2 > 0.f ? sym_launch(&force_sym_general_kernel<4, false, false, true>, a.n_tiles, 4, lds, a, stream)
                                    : sym_launch(&force_sym_general_kernel<4, false, true, true>, a.n_tiles, 4, lds, a, stream);
            return a.eps2 > 0.f ? sym_launch(&force_sym_kernel<4, false, 0, 2>, a.n_tiles, 4, lds, a, stream)
                                : sym_launch(&force_sym_kernel<4, true, 0, 2>, a.n_tiles, 4, lds, a, stream);
        }
    }
    // per-particle softening: the four-row loop S11 with eps > 0; with eps = 0 a particle may have eps_i = 0 too and the guarded,
    // compiler-scheduled kernel runs (NBODY_SYM_PACKED=0 / rows_per_lane 4: that kernel always -- A/B, tests)
    if (a.eps_pp && a.eps2 > 0.f && a.packed >= 2)
        return sym_launch(&force_sym_kernel<W, false, 4>, a.n_tiles, W, lds, a, stream);
    if (a.eps_pp)
        return a.eps2 > 0.f ? sym_launch(&force_sym_general_kernel<W, false, false, true>, a.n_tiles, W, lds, a, stream)
                            : sym_launch(&force_sym_general_kernel<W, false, true, true>, a.n_tiles, W, lds, a, stream);
    return a.eps2 > 0.f ? sym_launch(&force_sym_kernel<W, false>, a.n_tiles, W, lds, a, stream)
                        : sym_launch(&force_sym_kernel<W, true>, a.n_tiles, W, lds, a, stream);
}

template <int W>
static hipError_t sym_launch_diag(const SymArgs &a, size_t lds, hipStream_t stream)
{
    if (a.eps_pp)
        return a.eps2 > 0.f ? sym_launch(&force_sym_general_kernel<W, true, false, true>, a.n_diag, W, lds, a, stream)
                            : sym_launch(&force_sym_general_kernel<W, true, true, true>, a.n_diag, W, lds, a, stream);
    return a.eps2 > 0.f ? sym_launch(&force_sym_general_kernel<W, true, false, false>, a.n_diag, W, lds, a, stream)
                        : sym_launch(&force_sym_general_kernel<W, true, true, false>, a.n_diag, W, lds, a, stream);
}

hipError_t launch_forces_symmetric(const SymArgs &a, hipStream_t stream)
{
    if (sym_quarter_tiles(a.split_len, a.eps2, a.eps_pp, a.packed)) {  // small systems: tiles and diagonal tiles in one launch
        if (a.n_tiles + a.n_diag > 0) {
            const dim3 grid(a.n_tiles + a.n_diag);
            const int loop = a.eps_pp ? 1 : a.eps2 > 0.f ? 0 : 2;
            if (a.split_len == 256) {
                if (loop == 1)
                    hipLaunchKernelGGL((force_sym_quarter_kernel<1, 1>), grid, dim3(256), 0, stream, a);
                else if (loop == 0)
                    hipLaunchKernelGGL((force_sym_quarter_kernel<0, 1>), grid, dim3(256), 0, stream, a);
                else
                    hipLaunchKernelGGL((force_sym_quarter_kernel<2, 1>), grid, dim3(256), 0, stream, a);
            } else {  // (sym_quarter_tiles: no per-particle softening here)
                if (loop == 0)
                    hipLaunchKernelGGL((force_sym_quarter_kernel<0, 2>), grid, dim3(512), 0, stream, a);
                else
                    hipLaunchKernelGGL((force_sym_quarter_kernel<2, 2>), grid, dim3(512), 0, stream, a);
            }
        }
        return hipGetLastError();
    }
    // eight rows per lane for the equal-mass tiles (packed == 2): half the waves per split, 512 rows each
    // (512-body splits, one wave per workgroup, measured 0.8 % slower than the four-row loop at N = 131072: multiples of
    // 1024 only)
    if (a.packed >= 2 && a.eps_pp && a.eps2 > 0.f && a.split_len % 512 == 0) {  // per-particle softening: the eight-row loop S10
        const int w8 = a.split_len % 2048 == 0 ? 4 : a.split_len % 1024 == 0 ? 2 : 1;  // whole passes of 512 rows per wave
        const size_t lds8 = sym_lds_bytes_for(w8, a.split_len);
        if (a.split_len == 512 * w8)  // one pass: the rows stay in registers across a strip
            return w8 == 4   ? sym_launch(&force_sym_kernel<4, false, 3, 1>, a.n_tiles, 4, lds8, a, stream)
                   : w8 == 2 ? sym_launch(&force_sym_kernel<2, false, 3, 1>, a.n_tiles, 2, lds8, a, stream)
                             : sym_launch(&force_sym_kernel<1, false, 3, 1>, a.n_tiles, 1, lds8, a, stream);
        return w8 == 4   ? sym_launch(&force_sym_kernel<4, false, 3>, a.n_tiles, 4, lds8, a, stream)
               : w8 == 2 ? sym_launch(&force_sym_kernel<2, false, 3>, a.n_tiles, 2, lds8, a, stream)
                         : sym_launch(&force_sym_kernel<1, false, 3>, a.n_tiles, 1, lds8, a, stream);
    }
    if ((a.packed == 2 || a.packed == 3) && !a.eps_pp && a.eps2 > 0.f && a.split_len % 1024 == 0) {
        const int w8 = a.split_len % 2048 == 0 ? 4 : 2;  // whole passes of 512 rows per wave
        const size_t lds8 = sym_lds_bytes_for(w8, a.split_len);
        if (a.packed == 3 && a.split_len == 512 * w8)  // both eight-row loops, one pass: the rows stay in registers across a strip
            return w8 == 4 ? sym_launch(&force_sym_kernel<4, false, 2, 1>, a.n_tiles, 4, lds8, a, stream)
                           : sym_launch(&force_sym_kernel<2, false, 2, 1>, a.n_tiles, 2, lds8, a, stream);
        if (a.packed == 3)  // the eight-row loop for arbitrary masses too (three waves per SIMD)
            return w8 == 4 ? sym_launch(&force_sym_kernel<4, false, 2>, a.n_tiles, 4, lds8, a, stream)
                           : sym_launch(&force_sym_kernel<2, false, 2>, a.n_tiles, 2, lds8, a, stream);
        if (a.strip_len > 1)  // strips exist with 2048-body splits only: four waves
            return sym_launch(&force_sym_kernel<4, false, 1, 2>, a.n_tiles, 4, lds8, a, stream);
        return w8 == 4 ? sym_launch(&force_sym_kernel<4, false, 1>, a.n_tiles, 4, lds8, a, stream)
                       : sym_launch(&force_sym_kernel<2, false, 1>, a.n_tiles, 2, lds8, a, stream);
    }
    const size_t lds = symmetric_lds_bytes(a.split_len);
    switch (sym_waves(a.split_len)) {
    case 4: return sym_launch_tiles<4>(a, lds, stream);
    case 2: return sym_launch_tiles<2>(a, lds, stream);
    default: return sym_launch_tiles<1>(a, lds, stream);
    }
}

hipError_t launch_forces_symmetric_diag(const SymArgs &a, hipStream_t stream)
{
    if (sym_quarter_tiles(a.split_len, a.eps2, a.eps_pp, a.packed))
        return hipSuccess;  // served by the tile launch
    // A diagonal workgroup must fit where a tile workgroup leaves: beside the two-wave tile kernels of 1024-body splits
    // (three waves of ~165 registers per SIMD) a four-wave diagonal workgroup found room only in the launch's tail -- at N = 131 072
    // the diagonal launch ended 85 us after the tiles and was the step's critical path (profiles/r04_diagonal_tiles.txt).
    if (a.split_len == 1024)
        return sym_launch_diag<2>(a, sym_lds_bytes_for(2, a.split_len), stream);
    const size_t lds = symmetric_lds_bytes(a.split_len);
    switch (sym_waves(a.split_len)) {
    case 4: return sym_launch_diag<4>(a, lds, stream);
    case 2: return sym_launch_diag<2>(a, lds, stream);
    default: return sym_launch_diag<1>(a, lds, stream);
    }
}

// ---- the canonical summation of the pair-once partial sums (HBM-bound, O(N n_splits)) ------------------------------
// The order is the same for 1, 2, 4 or 8 ranks: kSymGroups groups of `group_splits` consecutive splits; a rank owns
// whole groups, sums the column-side terms of each of its groups for EVERY body (sym_colparts_kernel; that is what
// the ranks exchange), and the owner of a body adds, group by group, its row-side terms of the group and the group's
// column-side sum (sym_finalize_kernel).

__global__ __launch_bounds__(kTile) void sym_colparts_kernel(const float3 *col_partials, float4 *colparts, int n_total,
                                                             int split_len, int n_splits, int split_lo, int group_splits,
                                                             int group_lo)
{
    const int c = blockIdx.x * kTile + threadIdx.x;
    const int g = group_lo + blockIdx.y;
    if (c >= n_total)
        return;
    const int C = c / split_len, off = c - C * split_len;  // C is uniform in the workgroup (split_len % kTile == 0)
    const int r0 = g * group_splits, r1 = min(r0 + group_splits, n_splits);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    // the loads of eight row splits are issued together, the adds follow in order: with one load in flight per lane the walk ran
    // at the memory latency (3.8 TB/s with every wave slot taken), not at HBM speed
    constexpr int kBatch = 8;
    for (int R0 = r0; R0 < r1; R0 += kBatch) {
        float3 v[kBatch];
        bool on[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int R = R0 + i;
            on[i] = R < r1 && sym_rows_side(R, C, n_splits);
            if (on[i])
                v[i] = sym_col_slot(const_cast<float3 *>(col_partials), R - split_lo, sym_distance(R, C, n_splits), n_splits,
                                    split_len)[off];
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i)
            if (on[i]) {
                sx += v[i].x;
                sy += v[i].y;
                sz += v[i].z;
            }
    }
    colparts[(size_t)g * n_total + c] = make_float4(sx, sy, sz, 0.f);
}

// rowsum[g][b] = sum over the column splits C of group g (ascending, where the tile (B(b), C) exists, and the diagonal
// C == B(b)) of P_row[d(B, C)][b], for the rows [row_lo, row_lo + row_count) whose [n_splits/2+1][row_count] array
// row_partials is: the row-side half of the canonical summation.  It needs only the tiles whose row side is b's split, so
// the rows of the groups a launch has finished can be summed while later tiles still run.  rowsum (already offset to the
// first of these rows) has out_stride entries per group.
__global__ __launch_bounds__(kTile) void sym_rowsum_kernel(const float3 *row_partials, float4 *rowsum, int row_lo, int row_count,
                                                           int split_len, int n_splits, int group_splits, int out_stride, int strip_len)
{
    // grid.y = the column groups: one (row, group) sum per lane.  A part of one row group at N = 2^20 is 131 072 rows: with
    // one lane per row walking all 257 entries the pass was bound by load latency (0.27 ms for 0.4 GB), not by HBM.
    const int b = blockIdx.x * kTile + threadIdx.x;
    const int g = blockIdx.y;
    if (b >= row_count)
        return;
    const int B = (row_lo + b) / split_len;
    const int c0 = g * group_splits, c1 = min(c0 + group_splits, n_splits);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    // the group's column splits in ascending order, strip by strip (strip_len 1: tile by tile); the diagonal tile at its place
    // Strip by strip (strip_len 1: tile by tile): the strip's sum where its first tile stands, the diagonal tile's where the own
    // split stands -- before or behind the strip of its block.  The loads of eight strips are issued together, the adds follow in
    // that order (one load in flight per lane ran at the memory latency: 0.8 TB/s for this walk at N = 2^20).
    constexpr int kBatch = 8;
    const int K = strip_len;  // divides group_splits (nbody_set_strip_len's condition), so a block never straddles a group
    for (int J0 = c0 / K; J0 * K < c1; J0 += kBatch) {
        float3 v[kBatch], d[kBatch];
        int slot[kBatch], diag_at[kBatch];  // diag_at: 0 no diagonal tile in the block, 1 before the strip's sum, 2 behind it
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int lo = (J0 + i) * K;
            slot[i] = -1;
            diag_at[i] = 0;
            if (lo < c1) {
                int first = -1;
                for (int C = lo; C < min(lo + K, c1); ++C)
                    if (first < 0 && C != B && sym_rows_side(B, C, n_splits))
                        first = C;
                if (first >= 0) {
                    slot[i] = sym_row_slot(B, first, n_splits, K);
                    v[i] = row_partials[(size_t)slot[i] * row_count + b];
                }
                if (B >= lo && B < min(lo + K, c1)) {
                    diag_at[i] = first >= 0 && first < B ? 2 : 1;
                    d[i] = row_partials[b];  // slot 0
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            if (diag_at[i] == 1) {
                sx += d[i].x;
                sy += d[i].y;
                sz += d[i].z;
            }
            if (slot[i] >= 0) {
                sx += v[i].x;
                sy += v[i].y;
                sz += v[i].z;
            }
            if (diag_at[i] == 2) {
                sx += d[i].x;
                sy += d[i].y;
                sz += d[i].z;
            }
        }
    }
    rowsum[(size_t)g * out_stride + b] = make_float4(sx, sy, sz, 0.f);
}

// acc[b] = sum over the groups g (ascending) of ( rowsum[g][b] + colparts[g][b] ): the same association for any number
// of ranks and whether or not part of the sums was formed early.
__global__ __launch_bounds__(kTile) void sym_combine_kernel(const float4 *rowsum, const float4 *colparts, float4 *acc, int row_lo,
                                                            int row_count, int n_total, int n_groups)
{
    const int b = blockIdx.x * kTile + threadIdx.x;
    if (b >= row_count)
        return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int g = 0; g < n_groups; ++g) {
        const float4 rs = rowsum[(size_t)g * row_count + b];
        const float4 cp = colparts[(size_t)g * n_total + row_lo + b];
        ax += rs.x + cp.x;
        ay += rs.y + cp.y;
        az += rs.z + cp.z;
    }
    acc[b] = make_float4(ax, ay, az, 0.f);
}

// ---- one finishing kernel for small and mid-size systems (round 3) ----------------------------------------------------------
// A context that owns every row and ran its tiles in ONE part: sym_colparts_kernel's sum, sym_rowsum_kernel's sum, their
// combination and the kick-drift of update_kernel in one launch instead of three -- one lane per (body, group), the eight
// group values of a body combined in ascending order through LDS: the same sums in the same association, not a bit changes.
// At N = 32 768 the tile kernel takes 185 us of a 234 us step and the rest is small launches (profiles/r03_mid_range_kernel_stats.txt).
// MODE 0: kick-drift (update_kernel); 1: the closing half kick of kick-drift-kick, the acceleration kept (kdk_kick_kernel<1>);
// 2: the acceleration only (kdk_kick_kernel<0>) -- pos_all is then the acceleration array.
template <int MODE>
__global__ __launch_bounds__(256) void sym_finish_update_kernel(const float3 *row_partials, const float3 *col_partials,
                                                                float4 *pos_all, float4 *vel_rows, int n_total, int split_len,
                                                                int n_splits, int group_splits, int n_groups, float dt)
{
    __shared__ float part[kSymGroups][32][3];
    const int bl = threadIdx.x & 31, g = threadIdx.x >> 5;  // 32 consecutive bodies per group: 384 contiguous bytes per load
    const int b = blockIdx.x * 32 + bl;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (b < n_total && g < n_groups) {
        const int B = b / split_len, off = b - B * split_len;
        const int s0 = g * group_splits, s1 = min(s0 + group_splits, n_splits);
        float cx = 0.f, cy = 0.f, cz = 0.f, rx = 0.f, ry = 0.f, rz = 0.f;
        // the loads of eight splits are issued together, the adds follow in order (one load per add was a chain of ~20 memory
        // latencies: 11 us at 20 225 bodies, profiles/r04_pair_once_small_n.txt)
        constexpr int kBatch = 8;
        for (int R0 = s0; R0 < s1; R0 += kBatch) {  // column side: body b as a column of the tiles (R, B), R in group g
            float3 v[kBatch];
            bool on[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int R = R0 + i;
                on[i] = R < s1 && sym_rows_side(R, B, n_splits);
                if (on[i])
                    v[i] = sym_col_slot(const_cast<float3 *>(col_partials), R, sym_distance(R, B, n_splits), n_splits, split_len)[off];
            }
#pragma unroll
            for (int i = 0; i < kBatch; ++i)
                if (on[i]) {
                    cx += v[i].x;
                    cy += v[i].y;
                    cz += v[i].z;
                }
        }
        for (int C0 = s0; C0 < s1; C0 += kBatch) {  // row side: body b as a row of the tiles (B, C), C in group g, and of the diagonal tile
            float3 v[kBatch];
            bool on[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int C = C0 + i;
                on[i] = C < s1 && (C == B || sym_rows_side(B, C, n_splits));
                if (on[i])
                    v[i] = row_partials[(size_t)sym_distance(B, C, n_splits) * n_total + b];
            }
#pragma unroll
            for (int i = 0; i < kBatch; ++i)
                if (on[i]) {
                    rx += v[i].x;
                    ry += v[i].y;
                    rz += v[i].z;
                }
        }
        vx = rx + cx;  // sym_combine_kernel: rs + cp, then into the running sum
        vy = ry + cy;
        vz = rz + cz;
    }
    part[g][bl][0] = vx;
    part[g][bl][1] = vy;
    part[g][bl][2] = vz;
    __syncthreads();
    if (g != 0 || b >= n_total)
        return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int k = 0; k < n_groups; ++k) {
        ax += part[k][bl][0];
        ay += part[k][bl][1];
        az += part[k][bl][2];
    }
    if (MODE != 0) {
        pos_all[b] = make_float4(ax, ay, az, 0.f);  // sym_combine_kernel's entry, which kdk_kick_kernel copies
        if (MODE == 1) {
            float4 v = vel_rows[b];
            const double hh = 0.5 * (double)dt;
            v.x = (float)__builtin_fma((double)ax, hh, (double)v.x);
            v.y = (float)__builtin_fma((double)ay, hh, (double)v.y);
            v.z = (float)__builtin_fma((double)az, hh, (double)v.z);
            vel_rows[b] = v;
        }
        return;
    }
    float4 v = vel_rows[b];
    float4 x = pos_all[b];
    const double h = (double)dt;
    v.x = (float)__builtin_fma((double)ax, h, (double)v.x);
    v.y = (float)__builtin_fma((double)ay, h, (double)v.y);
    v.z = (float)__builtin_fma((double)az, h, (double)v.z);
    x.x = (float)__builtin_fma((double)v.x, h, (double)x.x);
    x.y = (float)__builtin_fma((double)v.y, h, (double)x.y);
    x.z = (float)__builtin_fma((double)v.z, h, (double)x.z);
    vel_rows[b] = v;
    pos_all[b] = x;
}

hipError_t launch_sym_finish_update(const float3 *row_partials, const float3 *col_partials, float4 *pos_all, float4 *vel_rows,
                                    int n_total, int split_len, int n_splits, int group_splits, float dt, hipStream_t stream)
{
    if (n_total <= 0)
        return hipSuccess;
    const int n_groups = (n_splits + group_splits - 1) / group_splits;
    hipLaunchKernelGGL(sym_finish_update_kernel<0>, dim3((n_total + 31) / 32), dim3(256), 0, stream, row_partials, col_partials, pos_all,
                       vel_rows, n_total, split_len, n_splits, group_splits, n_groups, dt);
    return hipGetLastError();
}

hipError_t launch_sym_finish_kick(const float3 *row_partials, const float3 *col_partials, float4 *acc, float4 *vel_rows, int n_total,
                                  int split_len, int n_splits, int group_splits, float dt, bool kick, hipStream_t stream)
{
    if (n_total <= 0)
        return hipSuccess;
    const int n_groups = (n_splits + group_splits - 1) / group_splits;
    if (kick)
        hipLaunchKernelGGL(sym_finish_update_kernel<1>, dim3((n_total + 31) / 32), dim3(256), 0, stream, row_partials, col_partials,
                           acc, vel_rows, n_total, split_len, n_splits, group_splits, n_groups, dt);
    else
        hipLaunchKernelGGL(sym_finish_update_kernel<2>, dim3((n_total + 31) / 32), dim3(256), 0, stream, row_partials, col_partials,
                           acc, nullptr, n_total, split_len, n_splits, group_splits, n_groups, 0.f);
    return hipGetLastError();
}

hipError_t launch_sym_colparts(const float3 *col_partials, float4 *colparts, int n_total, int split_len, int n_splits,
                               int split_lo, int group_splits, int group_lo, int group_count, hipStream_t stream)
{
    if (n_total <= 0 || group_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(sym_colparts_kernel, dim3((n_total + kTile - 1) / kTile, group_count), dim3(kTile), 0, stream,
                       col_partials, colparts, n_total, split_len, n_splits, split_lo, group_splits, group_lo);
    return hipGetLastError();
}

hipError_t launch_sym_rowsum(const float3 *row_partials, float4 *rowsum, int row_lo, int row_count, int split_len, int n_splits,
                             int group_splits, int out_stride, int strip_len, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    const int n_groups = (n_splits + group_splits - 1) / group_splits;
    hipLaunchKernelGGL(sym_rowsum_kernel, dim3((row_count + kTile - 1) / kTile, n_groups), dim3(kTile), 0, stream, row_partials,
                       rowsum, row_lo, row_count, split_len, n_splits, group_splits, out_stride, strip_len);
    return hipGetLastError();
}

hipError_t launch_sym_combine(const float4 *rowsum, const float4 *colparts, float4 *acc, int row_lo, int row_count, int n_total,
                              int n_groups, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(sym_combine_kernel, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, rowsum, colparts, acc,
                       row_lo, row_count, n_total, n_groups);
    return hipGetLastError();
}

}  // namespace nbody
