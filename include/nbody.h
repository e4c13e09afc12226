/*
 * nbody.h -- C ABI of the MI355X-native all-pairs N-body step (libnbody_amd.so).
 *
 * Drop-in boundary for ONE path of ctbfl/N_body_problem: the per-frame "step" bracket of
 * main_project/kernel.cu:1225-1242 (map the position VBO, run the force + update kernels,
 * synchronise, unmap) together with the buffer set-up that feeds it (kernel.cu:130-188).
 * File:line citations below are relative to the reference's main_project/.
 *
 * Buffer layout (the reference's OpenGL-interop layout, kernel.cu:139-158, 1168-1171):
 *   positions  : n x float4 {x, y, z, mass}, 16-byte aligned, updated IN PLACE by a step
 *                (a renderer may read it between steps, as GL does at kernel.cu:1259-1261);
 *   velocities : n x float4 {vx, vy, vz, w}; .w (the per-particle eps the reference loads at
 *                kernel.cu:223 and never reads) is preserved untouched.
 * All device pointers are plain HIP device addresses on the context's device.  No GL objects:
 * cudaGraphicsResourceGetMappedPointer (kernel.cu:1226) becomes "the caller passes a pointer".
 *
 * Every entry point returns NBODY_OK (0) or a negative nbody_status; nothing throws, nothing
 * prints.  The message of the last failure is kept per context (nbody_last_error).
 * A context is not thread-safe (the reference drives everything from one host thread).
 *
 * Numerics: fp32 storage and pair arithmetic; r^2 + eps^2 by an FMA chain; v_rsq_f32; each
 * row's sum over columns runs in ascending column order inside fixed-length column "splits",
 * and the per-split partial sums are added in ascending split order by the update kernel, so
 * results are bit-identical for any sharding of rows or columns that respects split_len.
 * Update: v <- (float)fma((double)a,(double)dt,(double)v); x <- (float)fma((double)v,(double)dt,(double)x)
 * (kernel.cu:777-801 with TIME_TICK -> dt).
 */
#ifndef NBODY_AMD_H
#define NBODY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_ABI_VERSION 5
#define NBODY_MIN_SOFTENING 1.0e-9f

typedef struct nbody_ctx nbody_ctx;

typedef enum nbody_status {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = -1,   /* bad argument (NULL, negative size, misaligned range) */
    NBODY_ERR_ALLOC = -2,     /* device or host allocation failed (kernel.cu:1151-1160 returns -1) */
    NBODY_ERR_DEVICE = -3,    /* a HIP call or kernel failed (kernel.cu:1238-1241 prints and continues) */
    NBODY_ERR_NO_DEVICE = -4, /* no usable gfx950 device: there is no CPU fallback */
    NBODY_ERR_STATE = -5      /* call needs context-owned buffers that were never uploaded */
} nbody_status;

int nbody_abi_version(void);
const char *nbody_status_string(int status);

/* ---- context: replaces initialize(numBodies), kernel.cu:130-161, and the scratch set-up at :1148-1160 ----
 * nbody_create      : all n_total bodies are rows and columns of this context (one GPU).
 * nbody_create_shard: the context integrates rows [row_lo, row_lo+row_count) against all n_total
 *                     columns (one rank of a multi-GPU run).  split_len = columns per partial sum,
 *                     a multiple of 64 (of 256 in the pair-once mode), 0 = nbody_default_split_len(n_total) (n_total/128
 *                     rounded up to a multiple of 256, at most 8192; for systems of ~12 000 to 32 767 bodies the
 *                     length that makes rows x splits a whole number of waves per SIMD, e.g. 320 at the reference's
 *                     20 225 bodies); row_lo must be a
 *                     multiple of split_len so that shard boundaries never cut a split.
 * The context owns the acceleration partials (the reference's gravity_sum_array), a stream and, on
 * demand, position/velocity buffers.  n_total need not be padded; the reference's roundup(n,256)+1
 * zero-mass padding (kernel.cu:260-278) is accepted and leaves real bodies' results unchanged. */
int nbody_create(nbody_ctx **out, int device, int64_t n_total);
int nbody_create_shard(nbody_ctx **out, int device, int64_t n_total, int64_t row_lo, int64_t row_count,
                       int64_t split_len);
/* nbody_create_auto: nbody_create with the force mode that is faster at this body count already selected, and the split
 * length that mode wants (what nbody_set_force_mode(ctx, NBODY_FORCE_AUTO) does to an existing context): the pair-once
 * kernels from NBODY_PAIR_ONCE_MIN_BODIES bodies on (since round 4: at every size), the one-sided ones below.  The one call
 * a caller of the reference's bracket (kernel.cu:1225-1242) needs to land on the fast kernels at every N. */
int nbody_create_auto(nbody_ctx **out, int device, int64_t n_total);
int nbody_destroy(nbody_ctx *ctx);
const char *nbody_last_error(const nbody_ctx *ctx); /* ctx == NULL: last error of a failed create */
int64_t nbody_default_split_len(int64_t n_total);
int64_t nbody_split_len(const nbody_ctx *ctx);
int64_t nbody_n_total(const nbody_ctx *ctx);

/* ---- body order (host helper, no device work) ----
 * The force kernels run at the clock the power limit leaves them, and the power follows the operands: when neighbours
 * in the arrays are neighbours in space, consecutive pair evaluations toggle fewer bits -- the same N = 2^20 Plummer
 * sphere stored along a Morton curve instead of in random order runs its force pass 3.7 % faster, same instructions
 * (profiles/r02_body_order_force_pass.txt).  The reference draws whatever order its loader produced (kernel.cu:190-556);
 * order is not part of the physics.  nbody_morton_order fills perm[k] = the index, in the caller's arrays, of the body to
 * store at slot k: along a Morton curve (20 bits per axis over the bounding cube of the finite positions; ties by index),
 * and, when the set has at most NBODY_ORDER_MAX_SPECIES distinct masses, the bodies of one mass together in order of mass
 * (so that the splits of a few-species set each keep one mass -- the equal-mass inner loops -- wherever the species were
 * stored).  Deterministic; a pure function of the n float4 {x, y, z, m}.  The host layers apply it at upload and undo it
 * at download (n_body_problem_amd.NBodySystem(body_order="morton"), nbody_run --morton); with caller-owned device
 * buffers the caller stores its bodies that way itself. */
#define NBODY_ORDER_MAX_SPECIES 16
enum { NBODY_ORDER_GIVEN = 0, NBODY_ORDER_MORTON = 1 };
int nbody_morton_order(const float *host_xyzm, int64_t n, int64_t *perm);

/* ---- the same order computed ON THE DEVICE (csrc/nbody_order.hip): a layout refresh without a host copy of the state ----
 * The layout decays as the bodies move (N = 2^20, dt = 1e-3: half of the gain is gone after ~300 steps); through the host a
 * refresh costs 0.13-0.3 s at N = 2^20, here well under a millisecond of device time (bounding cube and mass species by
 * atomics, keys in double exactly as the host computes them, a stable 64-bit radix sort, gathers) -- the SAME permutation
 * as nbody_morton_order, bit for bit, so every rank of a multi-GPU run gets it from its own replica.  All calls are
 * asynchronous on the context's stream.
 * nbody_reorder: a context that owns every row: the first n bodies (n <= n_total; a zero-mass padding tail stays where it
 *   is) of d_positions_xyzm, d_velocities_xyzw and, when not NULL, d_eps (n floats, e.g. the array given to
 *   nbody_set_particle_softening) are permuted in place into nbody_morton_order of the CURRENT positions; the context's own
 *   softening copy (nbody_upload_particle_softening) follows; d_order (n int64 on the device, or NULL) is composed:
 *   d_order[k] <- d_order[perm[k]], so that an array started with nbody_order_identity keeps saying which of the caller's
 *   bodies sits in slot k -- and bodies with equal keys are placed in the order of THOSE indices, so the layout is
 *   nbody_morton_order of the bodies in the caller's order, a pure function of the body set, however often it has been
 *   refreshed.  Cached accelerations (kick-drift-kick) are forgotten.
 * nbody_morton_order_device: only the permutation, n int64 on the device.
 * The pieces, for a sharded host (nbody_multi_reorder is built from them): nbody_order_compute keeps the permutation of
 *   the first n bodies in the context (the identity beyond n; d_order as above, or NULL); nbody_order_gather applies it,
 *   d_dst[j] = d_src[perm[first + j]] for j < count, rows of 1, 2 (one int64) or 4 floats, d_src indexed by body -- in
 *   place (d_dst == d_src) only from first = 0; nbody_order_read widens it to int64; nbody_order_permute_softening
 *   applies it to the context's own softening copy, if it has one in use. */
int nbody_reorder(nbody_ctx *ctx, float *d_positions_xyzm, float *d_velocities_xyzw, float *d_eps, int64_t *d_order, int64_t n);
int nbody_morton_order_device(nbody_ctx *ctx, const float *d_xyzm, int64_t n, int64_t *d_perm);
int nbody_order_compute(nbody_ctx *ctx, const float *d_xyzm, int64_t n, const int64_t *d_order);
int nbody_order_gather(nbody_ctx *ctx, void *d_dst, const void *d_src, int64_t first, int64_t count, int floats_per_row);
int nbody_order_read(nbody_ctx *ctx, int64_t *d_perm);
int nbody_order_identity(nbody_ctx *ctx, int64_t *d_order, int64_t n);
/* The context's permutation from an index array on the device: perm[k] = d_order[k], or, inverse != 0, perm[d_order[k]] = k
 * (d_order a permutation of 0..n-1): gathering with the inverse of an order array puts the bodies back in the caller's order. */
int nbody_order_set(nbody_ctx *ctx, const int64_t *d_order, int64_t n, int inverse);
int nbody_order_permute_softening(nbody_ctx *ctx);

/* ---- context-owned buffers: setParticlesPosition / setParticlesVelocity, kernel.cu:163-188 ----
 * Host arrays of n_total float4 (positions) and row_count float4 (velocities of this context's rows).
 * Allocates the device buffers on first use.  nbody_download copies back; either pointer may be NULL. */
int nbody_set_positions(nbody_ctx *ctx, const float *host_xyzm);
int nbody_set_velocities(nbody_ctx *ctx, const float *host_xyzw);
int nbody_download(nbody_ctx *ctx, float *host_xyzm, float *host_xyzw);
/* Device addresses of the owned buffers (NULL before the first set_*): the mapped-pointer analogue
 * of cudaGraphicsResourceGetMappedPointer, kernel.cu:1226. */
float *nbody_positions_device(nbody_ctx *ctx);
float *nbody_velocities_device(nbody_ctx *ctx);

/* ---- the step: the bracket kernel.cu:1225-1242 = step(positions, velocities, masses, dt, softening) ----
 * d_positions_xyzm : n_total float4, device, updated in place (rows of this context only).
 * d_velocities_xyzw: row_count float4, device: velocities of rows row_lo.. (the whole array for nbody_create).
 * d_masses         : NULL => mass is positions[4i+3] (the reference's only mode); else n_total floats that are
 *                    first copied into positions[4i+3].
 * dt               : TIME_TICK (kernel.cu:63; the reference uses 0.008).
 * softening        : eps, a length; eps^2 replaces EPSILON (kernel.cu:66).  The reference's VERSION 3
 *                    corresponds to 1e-2 and VERSIONs 1/2 to 1e-3.  0 is allowed: pairs
 *                    at zero distance (closer than 2.3e-13: the self pair, coincident bodies) then contribute
 *                    nothing, whatever their masses.  0 < softening < NBODY_MIN_SOFTENING is rejected (NBODY_ERR_INVALID): eps^-3 x mass
 *                    of the self pair would overflow fp32 and poison every sum with 0 x inf.
 * nbody_step returns after the device work is complete (the reference synchronises at kernel.cu:1232,1236);
 * nbody_step_async only enqueues on the context's stream; nbody_sync waits and reports kernel errors. */
int nbody_step(nbody_ctx *ctx, float *d_positions_xyzm, float *d_velocities_xyzw, const float *d_masses, float dt,
               float softening);
int nbody_step_async(nbody_ctx *ctx, float *d_positions_xyzm, float *d_velocities_xyzw, const float *d_masses,
                     float dt, float softening);
int nbody_step_n(nbody_ctx *ctx, int k, float dt, float softening); /* k steps on the owned buffers, one sync */
/* The same on caller-owned device buffers.  Where the loop is launch-bound -- measured: a pair-once step of several launches on
 * two streams (per-particle softening with softening = 0, the A/B arrangements of nbody_set_rows_per_lane) up to 32 768
 * bodies; the launches of a one-sided step, and since round 4 the two kernels of a pair-once step with 256- or 512-body splits,
 * are faster enqueued eagerly at every size -- ONE step is captured from the stream into a HIP graph after an eager first step and replayed k - 1 times: the
 * same kernels, arguments and order, hence the same bits.  nbody_set_graph_replay: -1 automatic (that rule), 0 never,
 * 1 always. */
int nbody_step_n_on(nbody_ctx *ctx, float *d_positions_xyzm, float *d_velocities_xyzw, int k, float dt, float softening);
int nbody_set_graph_replay(nbody_ctx *ctx, int mode);
int nbody_sync(nbody_ctx *ctx);

/* ---- the two halves of a step, for callers that interleave an exchange (multi-GPU) ----
 * nbody_forces: partial accelerations of this context's rows from columns [col_lo, col_lo+col_count)
 *   (cal_acc_advanced's job, kernel.cu:703-774, for a column range); col_lo must be a multiple of split_len
 *   and the range must end on a split boundary or at n_total.  Asynchronous.
 * nbody_update: adds the partials of ALL splits in ascending order and applies the kick-drift of
 *   use_acc_update_position (kernel.cu:777-801) to this context's rows.  Asynchronous.
 * Every split must have been produced by nbody_forces since the previous nbody_update. */
int nbody_forces(nbody_ctx *ctx, const float *d_positions_xyzm, int64_t col_lo, int64_t col_count, float softening);
/* The same for every column EXCEPT [col_lo, col_lo+col_count), in one launch: what a rank runs once the
 * exchange has delivered the other ranks' rows, its own column chunk having been done beside the exchange. */
int nbody_forces_complement(nbody_ctx *ctx, const float *d_positions_xyzm, int64_t col_lo, int64_t col_count,
                            float softening);
int nbody_update(nbody_ctx *ctx, float *d_positions_xyzm, float *d_velocities_xyzw, float dt);

/* ---- integrator ----
 * NBODY_INTEGRATOR_KICK_DRIFT (default): the reference's final scheme, v += a(x) dt ; x += v dt (kernel.cu:777-801).
 * NBODY_INTEGRATOR_KDK: velocity Verlet, v += a dt/2 ; x += v dt ; v += a(x_new) dt/2 -- the scheme of the reference's
 *   historical update_speed_half / update_position_complete (unused_files/backup.cu:859-887, driven at :1848-1866
 *   with TWO force evaluations per step); here the accelerations are cached, one evaluation per step.  nbody_step
 *   / nbody_step_n follow the selected integrator; positions changed behind the library's back (a new buffer, an
 *   edit) need nbody_invalidate_forces.  A sharded host drives the pieces itself, because the drifted rows must be
 *   exchanged before the forces: forces + nbody_kdk_prepare once, then per step nbody_kdk_kick_drift -> exchange ->
 *   nbody_forces (+ _complement) -> nbody_kdk_kick. */
enum { NBODY_INTEGRATOR_KICK_DRIFT = 0, NBODY_INTEGRATOR_KDK = 1 };
int nbody_set_integrator(nbody_ctx *ctx, int integrator);
/* Forgets the cached accelerations and every partial sum nbody_forces has produced since the last update. */
int nbody_invalidate_forces(nbody_ctx *ctx);
int nbody_kdk_prepare(nbody_ctx *ctx);
int nbody_kdk_kick_drift(nbody_ctx *ctx, float *d_positions_xyzm, float *d_velocities_xyzw, float dt);
int nbody_kdk_kick(nbody_ctx *ctx, float *d_velocities_xyzw, float dt);

/* nbody_set_stream: enqueue on the caller's HIP stream (a hipStream_t passed as void*, used verbatim: NULL is
 * the HIP default stream).  nbody_reset_stream: back to the context's own non-blocking stream (the default). */
int nbody_set_stream(nbody_ctx *ctx, void *hip_stream);
int nbody_reset_stream(nbody_ctx *ctx);

/* ---- diagnostics (the reference has none) ----
 * nbody_energy: out = {kinetic, potential, total} of this context's rows against all columns
 *   (potential = -1/2 sum_i m_i sum_{j!=i} m_j / sqrt(r^2+eps^2), fp32 pair terms, fp64 sums); for a
 *   sharded run the caller adds the ranks' triples.  Synchronous.
 * nbody_momentum: out = {px, py, pz, mass} of this context's rows.  Synchronous. */
int nbody_energy(nbody_ctx *ctx, const float *d_positions_xyzm, const float *d_velocities_xyzw, float softening,
                 double *out3);
int nbody_momentum(nbody_ctx *ctx, const float *d_positions_xyzm, const float *d_velocities_xyzw, double *out4);

/* ---- measurement: HIP-event timing of the kernels on the stream they run on ----
 * With timing on, every force / update launch is bracketed by events.  nbody_timing_read synchronises,
 * returns the accumulated milliseconds and launch counts since the last read, and resets them. */
int nbody_timing_enable(nbody_ctx *ctx, int on);
/* out6 = {force_ms, force launches, update_ms, update launches, auxiliary ms, auxiliary launches}: the totals are SUMS of
 * per-launch durations (launches that overlap on two streams each count in full), not wall time.  force = the dominant
 * force kernel; update = what runs behind the force pass (summation, update, kicks); auxiliary = what the pair-once mode
 * runs on its second stream BESIDE the tile launches (the diagonal tiles, the summation of the finished row
 * groups) -- stretched by the sharing, hidden in wall time.  Waits for every recorded launch. */
int nbody_timing_read_ex(nbody_ctx *ctx, double *out6);
int nbody_timing_read(nbody_ctx *ctx, double *force_ms, int64_t *force_launches, double *update_ms,
                      int64_t *update_launches);

/* Force algorithm.  NBODY_FORCE_ONE_SIDED (default): every ordered interaction is evaluated, rows are independent
 * (the shape of simple_update_all, kernel.cu:828-884).  NBODY_FORCE_SYMMETRIC ("pair-once"): each unordered pair once,
 * applied to both bodies -- the idea of cal_acc_advanced, kernel.cu:703-774, without its float atomics.  Needs
 * 256 <= split_len <= 4096 (create the context with nbody_pair_once_split_len(n_total): small splits keep 5
 * workgroups on a CU).  Results agree with the default to rounding (not bit for bit) and are themselves bit-reproducible, and
 * identical for 1, 2, 4 or 8 contexts sharing the rows: the partial sums are added in NBODY_SYM_GROUPS groups of
 * ceil(n_splits / 8) splits, a sharded context must own whole groups, and per step
 *     nbody_forces / nbody_forces_complement   (as in the default mode: any split-aligned column ranges)
 *     nbody_sym_reduce                          (column-side sums of the context's groups, for every body)
 *     [exchange the contexts' slices of the colparts buffer -- the caller's job, e.g. RCCL -- and, beside it, nbody_sym_rowsum]
 *     nbody_update / nbody_kdk_*                (adds row-side and column-side sums in the fixed order)
 * A context that owns all rows may skip nbody_sym_reduce (nbody_update / nbody_step run it). */
/* NBODY_FORCE_AUTO (nbody_set_force_mode only; a context that owns every row, nothing pending): the pair-once mode from
 * NBODY_PAIR_ONCE_MIN_BODIES bodies on -- where it delivers more interactions per second -- the one-sided mode below, and the
 * context's split length is changed to the one that mode wants (nbody_pair_once_split_len / nbody_default_split_len).
 * Rounds 1-3: 32 768 (below it one wave per 256 x 256 tile left the grid too coarse: 0.157 against 0.122 ms per step at the
 * reference's 20 225 bodies).  Round 4: 0 -- with 256-body splits a tile is served by four waves, a 64-column group each, the
 * diagonal tiles in the same launch, and the pair-once step is the faster one at every size measured, 256 bodies to 2^22
 * (profiles/r04_pair_once_small_n.txt: 20 225 bodies 0.087 against 0.109 ms, 4096: 0.019 against 0.029, 32 768: 0.195 against
 * 0.258, 2^20: 150 against 229).  nbody_force_mode reads the mode in use. */
#define NBODY_PAIR_ONCE_MIN_BODIES 0
enum { NBODY_FORCE_ONE_SIDED = 0, NBODY_FORCE_SYMMETRIC = 1, NBODY_FORCE_AUTO = 2 };
int nbody_force_mode(const nbody_ctx *ctx);
/* The split length to create a pair-once context with.  A function of n_total ONLY (split boundaries define the
 * summation order, so they must not depend on the sharding): 1024 from 131 072 bodies up -- whole passes of the eight-row
 * loops, the finest grid that keeps every wave busy -- 512 from 65 536 and 256 below that (small systems need more, smaller
 * tiles to fill the chip: measured per size, profiles/r03_split_len_mid_range.txt), 2048 from N = 2^20 (half the partial sums: 2.2 % faster at equal memory)
 * and 4096 from 2^23, so that the partial sums of one pass (n_total^2 / split_len entries of 12 bytes over all contexts:
 * 6.4 GB at N = 2^20, 103 GB at N = 2^22) would still fit one GPU even in one summation part
 * (nbody_set_summation_parts; by default a single context holds 4.8 GB and 26 GB of them). */
int64_t nbody_pair_once_split_len(int64_t n_total);
#define NBODY_SYM_GROUPS 8
#define NBODY_PARTIAL_SUM_BUDGET_BYTES (5ll << 30) /* nbody_set_summation_parts(ctx, 0): the fewest parts that stay below */
int nbody_set_force_mode(nbody_ctx *ctx, int mode);
/* The exchange buffer of the pair-once mode: NBODY_SYM_GROUPS x n_total x float4 on the device, group-major.
 * d_buf is borrowed (NULL: a buffer the context owns -- enough for a single context).  nbody_sym_reduce writes the
 * slices [group_lo, group_lo + group_count) x n_total (nbody_sym_groups); the other slices must hold the other
 * contexts' sums before nbody_update. */
int nbody_sym_set_colparts(nbody_ctx *ctx, float *d_buf);
int nbody_sym_groups(const nbody_ctx *ctx, int64_t *group_lo, int64_t *group_count, int64_t *group_splits);
int nbody_sym_reduce(nbody_ctx *ctx);
/* Optional, after nbody_sym_reduce: the row-side sums of the context's rows now instead of inside nbody_update / nbody_kdk_*.
 * They need the context's own partial sums only, so a sharded host runs them while the column-side sums are on the wire
 * (nbody_multi_step does).  Not a bit changes. */
int nbody_sym_rowsum(nbody_ctx *ctx);

/* Per-particle softening: the reference loads a per-particle eps into velocities[4i+3]
 * (kernel.cu:223, 237) and no kernel ever reads it.  With d_eps (n_total floats on the device, borrowed until replaced;
 * NULL switches it off) every pair is softened by eps_ij^2 = softening^2 + eps_i^2 + eps_j^2 in the forces and in
 * nbody_energy.  One extra add per interaction, inside the hand-scheduled loops: the packed one-sided loops (+8 % per step at
 * N = 2^20, nothing at the reference's 20 225 bodies; same bits as the compiler-allocated kernel); in the pair-once mode the
 * eight-row loop (softening > 0, splits of whole 512 bodies, i.e. from 65 536 bodies on: 179.8 against 214.3 ms per
 * N = 2^20 step, 0.82 against 1.01 ms at N = 65 536, profiles/r03_pps_modes.txt) or the four-row loop (other split
 * lengths: 0.48 against 0.58 ms at N = 49 152); with softening = 0 the compiler-scheduled, guarded kernel. */
int nbody_set_particle_softening(nbody_ctx *ctx, const float *d_eps);
/* The same from n_total HOST floats, copied into a buffer the context owns (NULL switches it off). */
int nbody_upload_particle_softening(nbody_ctx *ctx, const float *h_eps);

/* Kernel selection for experiments and A/B measurement: rows per lane (1, 2, 4 or 8; 0 = default: 4, the kernel with the
 * hand-allocated, packed-fp32 inner loop, in 1024-row workgroups where those fill the chip and in one-wave workgroups of 256
 * rows below that; 41 forces the one-wave form, 40 = the loop with one row per instruction, -4 = four rows with the
 * compiler-allocated loop).  In the one-sided mode it never changes a result bit.  In the pair-once mode: 0 or 8 = the
 * eight-row loops on every tile of splits of whole 1024 bodies (the default), 4 = the packed four-row loops everywhere, 2 =
 * round 2's arrangement (eight rows on equal-mass tiles only, in a four-waves-per-SIMD kernel), 1 = the one-column loops
 * (what the environment switches of rounds 1-3 selected; the association of the sums follows the rows per wave). */
int nbody_set_rows_per_lane(nbody_ctx *ctx, int rows_per_lane);

/* Equal-mass splits (on by default).  Before every force launch an O(N) pass notes, per split, whether all its bodies
 * carry one mass (every split of an equal-mass system such as the benchmark's Plummer sphere; most splits of a
 * few-species file like galaxy_20K.bin).  Such a split's inner loop leaves the mass out -- the sums collect d x inv^3 and
 * are multiplied by the mass once when they are written -- which saves one of 12 fp32 instructions per interaction in the
 * one-sided kernels and two of 16 per pair in the pair-once tiles (both sides of the tile must qualify).  The choice is
 * a function of the data and of the split boundaries only, so every invariance stays bit-exact (register blocking, row
 * sharding, column ranges, GPU count); against the general path the results differ by rounding (m x sum instead of
 * sum of m x term).  0 switches it off: every split takes the general path (A/B measurement, tests). */
int nbody_set_equal_mass_path(nbody_ctx *ctx, int on);

/* Summation parts (pair-once mode, one context that owns every row, all columns in one force call).  The partial sums are
 * added in row groups (NBODY_SYM_GROUPS), so the tiles can be launched in parts of whole groups, and while one part's
 * tiles run an auxiliary stream already forms the sums of the part before it: only the last part's share of the
 * summation and the combination stay behind the force pass.  The result does not change by a bit (the association is by
 * groups either way).  parts = 1: one launch, then the whole summation.  2: every group but the last, then the last -- the
 * last part's tiles on a stream of the lowest priority that does not wait for the first part's, so the dispatcher fills the
 * first launch's tail with the second's workgroups (measured 0.2-0.3 % faster than one launch at N = 2^20; the two launches
 * overlap, so per-launch timings stop adding up to the pass); the partial-sum arrays hold the whole pass.
 * 4, 8: parts whose arrays live in two slots used in turn, for one launch tail (0.2-0.5 ms) per extra part (measured at
 * N = 2^20 with 1024-body splits: 8 parts cost ~1 % of the step against 1).  8: equal parts, a quarter of the pass held
 * (26 GB at N = 2^22).  4: 3 + 3 + 1 + 1 of the 8 groups -- what stays behind the force pass is the LAST part's summation,
 * so the last part is one group (update_ms 0.3 instead of 0.5 ms at N = 2^20) and the two slots hold three groups each:
 * 3/4 of the pass (4.8 GB at N = 2^20).  0, the default: automatic -- one launch while the whole pass fits
 * NBODY_PARTIAL_SUM_BUDGET_BYTES (5 GiB: up to N = 650 000 with 1024-body splits), else 4 parts if 3/4 of it does, else 8
 * (N = 2^20: 4 parts, 4.8 GB; N = 2^22: 8 parts, 26 GB).  Systems too small
 * for several launches, shards and column-range calls always take one part.  nbody_set_early_summation(on) is
 * nbody_set_summation_parts(on ? 0 : 1). */
int nbody_set_summation_parts(nbody_ctx *ctx, int parts);
/* Strips (pair-once mode, round 4).  With 2048-body splits (N >= 2^20) and a number of splits that is a multiple of 8 x K, a
 * tile workgroup takes K consecutive column splits of its row split and keeps the rows' sums in registers across them: 1 / K
 * of the row-side partial sums, and the whole pass of N = 2^20 held in ONE launch (the automatic summation parts then take one
 * part).  Automatic: K = 4 from 1024 splits on (N >= 2^21: a quarter of the row side), K = 2 below (N = 2^20: 4.8 instead of
 * 6.4 GB of partial sums per pass) -- a strip is one workgroup, and with strips of four a rank of eight's share of an N = 2^20
 * pass is 5.3 rounds of 3.5 ms workgroups: 21.6 ms against 20.0 with strips of two and 18.7 with single tiles, while one GPU
 * runs both strip lengths equally fast.  The blocks of K are absolute, so a strip never straddles a summation group or a
 * rank's column chunk: the state stays bit-identical for 1, 2, 4 or 8 contexts sharing the rows; against single tiles the
 * results differ by rounding (one chain per row over a strip's columns instead of a sum of K).  Column ranges given to
 * nbody_forces must then start and end on multiples of the strip length.  len: 0 = automatic (that rule), 1 = single tiles
 * everywhere, 2, 4, 8 = that many splits where the split count allows it (A/B measurement, tests). */
int nbody_set_strip_len(nbody_ctx *ctx, int len);
int nbody_set_early_summation(nbody_ctx *ctx, int on);
/* Bytes of partial-sum arrays the context holds at the moment (they are allocated by the first force call and only grow). */
int64_t nbody_partial_sum_bytes(const nbody_ctx *ctx);

/* Device facts for the roofline: out = {compute units, max clock MHz, wavefront size, LDS bytes per CU}. */
int nbody_device_info(nbody_ctx *ctx, int64_t *out4, char *name, int name_len);

/* ======== multi-GPU: rows sharded over the GPUs of one node, the exchange owned by the library ========
 * The reference is single-GPU (kernel.cu:1225-1242: one device, the default stream), so nothing here replaces a
 * reference interface: it is the same bracket -- step(positions, velocities, masses, dt, softening) -- for a body set
 * whose rows are dealt to P GPUs (SURVEY.md 8b "ownership"/"threading", 8e).  A nbody_multi owns per local rank one shard
 * context, a full replica of the positions, the rank's velocity rows, two compute streams, a communication stream and
 * one RCCL communicator; nbody_multi_step is the whole step (csrc/nbody_multi.hip):
 *     own-chunk force launch  ||  all-gather of the previous step's updated rows (in place, RCCL over xGMI)
 *     complement force launch behind the all-gather, on the second stream
 *     [pair-once mode: column-side sums of the rank's groups, then every rank hands every other rank the segments of its
 *      rows -- ncclSend / ncclRecv to each peer inside one group call]   update; the next all-gather is issued behind it.
 * The body count need not divide: the system is padded with zero-mass bodies at the origin (the reference's own padding
 * device, kernel.cu:265-277) to n_padded = P x rows_per_rank, rows_per_rank whole splits (whole split groups in the
 * pair-once mode, which therefore shards over 1, 2, 4 or 8 ranks).  The state is bit-identical to ONE context on the
 * same padded system for any P, exchange and transport.
 * Process models: nbody_multi_create -- every rank in this process, driven from the calling host thread (RCCL calls of
 * the local ranks fused with ncclGroupStart/End); nbody_multi_create_rank -- one rank per process (the launch model of
 * torchrun / mpirun): rank 0 calls nbody_multi_unique_id and the caller hands the 128 bytes to every rank by any
 * channel it has.  Failure detection: the communicators are non-blocking (ncclCommInitRankConfig, blocking = 0) and made by a
 * helper thread the caller waits for under the timeout, so their creation -- the bootstrap, where a job with a missing rank
 * hangs first -- is bounded whether or not the RCCL at hand honours blocking = 0 there; RCCL calls that answer ncclInProgress
 * are polled under the timeout;
 * RCCL's asynchronous error state is polled after every step and inside every wait; nbody_multi_step_n keeps the host at most
 * four steps ahead of the device, so no wait covers more than four steps; a wait longer than the timeout (default 600 s; the
 * environment variable NBODY_EXCHANGE_TIMEOUT_S -- the only one the library reads --, nbody_multi_config.create_timeout_s or
 * nbody_multi_set_timeout) aborts the communicators and returns NBODY_ERR_DEVICE instead of hanging on a dead peer. */
typedef struct nbody_multi nbody_multi;
#define NBODY_UNIQUE_ID_BYTES 128
enum { NBODY_EXCHANGE_ALLGATHER = 0, /* one ncclAllGather per step */
       NBODY_EXCHANGE_RING = 1 };     /* P-1 ncclSend/ncclRecv hops, the force launch of chunk rank-h starts as hop h lands */
enum { NBODY_TRANSPORT_RCCL = 0,
       NBODY_TRANSPORT_PEER_COPY = 1 }; /* hipMemcpyPeerAsync between the replicas: single process only; also the way two
                                           shards on ONE device are run (RCCL refuses duplicate devices) */
typedef struct nbody_multi_config {
    int64_t n_bodies;  /* real bodies */
    int64_t split_len; /* 0 = nbody_default_split_len / nbody_pair_once_split_len of n_bodies */
    int force_mode;    /* NBODY_FORCE_ONE_SIDED | NBODY_FORCE_SYMMETRIC | NBODY_FORCE_AUTO (pair-once from
                          NBODY_PAIR_ONCE_MIN_BODIES bodies on when the rank count divides NBODY_SYM_GROUPS) */
    int integrator;    /* NBODY_INTEGRATOR_KICK_DRIFT | NBODY_INTEGRATOR_KDK */
    int exchange;      /* NBODY_EXCHANGE_* */
    int transport;     /* NBODY_TRANSPORT_* */
    int body_order;    /* NBODY_ORDER_GIVEN | NBODY_ORDER_MORTON: nbody_multi_set_state stores the bodies in nbody_morton_order
                          of the positions it is given (the same on every rank), nbody_multi_download and
                          nbody_multi_set_particle_softening speak the caller's order; nbody_multi_order reads the permutation */
    int create_timeout_s; /* seconds the creation of the communicators and every later wait may take; 0 = the default
                             (600 s, or NBODY_EXCHANGE_TIMEOUT_S); nbody_multi_set_timeout changes it for the waits that follow */
} nbody_multi_config;

/* Pure host functions (no device needed): the padded size and rows per rank, and hop `hop` (1..P-1) of the ring. */
int nbody_multi_geometry(int64_t n_bodies, int world_size, int force_mode, int64_t split_len, int64_t *n_padded,
                         int64_t *rows_per_rank, int64_t *split_len_out);
int nbody_multi_ring_schedule(int rank, int world_size, int hop, int *send_chunk, int *recv_chunk);

int nbody_multi_unique_id(void *id128); /* ncclGetUniqueId into NBODY_UNIQUE_ID_BYTES bytes */
int nbody_multi_create(nbody_multi **out, const nbody_multi_config *cfg, const int *devices, int n_devices);
int nbody_multi_create_rank(nbody_multi **out, const nbody_multi_config *cfg, int device, int rank, int world_size,
                            const void *unique_id128);
int nbody_multi_destroy(nbody_multi *m);
const char *nbody_multi_last_error(const nbody_multi *m); /* m == NULL: last error of a failed create */
int nbody_multi_set_timeout(nbody_multi *m, double seconds);

/* Host arrays of n_bodies float4 each (setParticlesPosition / setParticlesVelocity, kernel.cu:163-188): every local
 * rank's replica and velocity rows are filled.  download: all n_bodies rows on every process (either pointer may be
 * NULL).  set_particle_softening: n_bodies host floats or NULL. */
int nbody_multi_set_state(nbody_multi *m, const float *host_xyzm, const float *host_xyzw);
/* The reference's two setters are independent copies (kernel.cu:163-188); so are these: new positions keep every body's
 * velocity and softening length (with NBODY_ORDER_MORTON the new curve is laid through the new positions and the velocities
 * follow their bodies), new velocities leave the positions and the layout alone. */
int nbody_multi_set_positions(nbody_multi *m, const float *host_xyzm);
int nbody_multi_set_velocities(nbody_multi *m, const float *host_xyzw);
int nbody_multi_set_particle_softening(nbody_multi *m, const float *host_eps);
int nbody_multi_download(nbody_multi *m, float *host_xyzm, float *host_xyzw);
/* perm[k] = the caller's index of the body stored in slot k of the replicas (the identity with NBODY_ORDER_GIVEN); n_bodies
 * entries, valid after nbody_multi_set_state.  A renderer that reads nbody_multi_positions_device sees this order. */
int nbody_multi_order(nbody_multi *m, int64_t *perm);
/* NBODY_ORDER_MORTON only (no-ops otherwise): the layout decays as the bodies move -- at N = 2^20 the first 100 steps of
 * dt = 1e-3 run 4.2 % faster than in the generator's order, steps 900-1000 1.5 % (profiles/r02_longrun_morton_decay_*).
 * nbody_multi_reorder lays a new curve through the current positions ON THE DEVICE: every rank sorts its own replica
 * (nbody_order_compute: the same permutation everywhere), the velocity rows are re-dealt through one gather of all rows
 * (RCCL all-gather or peer copies), softening lengths and the order array follow; no host copy of the state (round 2 went
 * through the host: 0.13-0.3 s at N = 2^20).  The kick-drift-kick mode recomputes its cached accelerations.  With a period
 * > 0 the first step that is due does it by itself.  Results stay deterministic and the same for every rank count; they
 * depend on the period (the order of the sums does).  Collective in the one-rank-per-process model. */
int nbody_multi_reorder(nbody_multi *m);
int nbody_multi_set_reorder_period(nbody_multi *m, int64_t steps);

/* The step.  nbody_multi_step / _step_n return with every replica current and all device work complete;
 * nbody_multi_step_async only enqueues (the exchange of the updated rows stays in flight under the next step). */
int nbody_multi_step(nbody_multi *m, float dt, float softening);
int nbody_multi_step_n(nbody_multi *m, int k, float dt, float softening);
int nbody_multi_step_async(nbody_multi *m, float dt, float softening);
int nbody_multi_sync(nbody_multi *m);

/* In the one-rank-per-process model every call from nbody_multi_set_state to nbody_multi_replica_checksums is
 * COLLECTIVE: all ranks make the same calls in the same order (the steps exchange rows, download gathers the
 * velocities, the diagnostics reduce over the ranks).
 * Diagnostics of the WHOLE system, the same values on every process.  replica_checksums: out2 = {smallest, largest}
 * checksum of the position replicas over all ranks -- equal when every rank holds the same bits. */
int nbody_multi_energy(nbody_multi *m, float softening, double *out3);
int nbody_multi_momentum(nbody_multi *m, double *out4);
int nbody_multi_replica_checksums(nbody_multi *m, uint64_t *out2);
/* out8 = {n_bodies, n_padded, rows_per_rank, split_len, world_size, local ranks, ranks of the RCCL communicator, exchange} */
int nbody_multi_info(const nbody_multi *m, int64_t *out8);
/* Measurement: where a rank's step goes besides its kernels.  With timing on, the exchanges are bracketed by HIP events on
 * the streams they run on (and the shard contexts time their kernels, nbody_timing_enable).  nbody_multi_timing_read waits
 * for everything recorded, returns the totals of local rank local_index since the last read and resets them:
 *   out16 = { steps, host milliseconds spent enqueuing them,
 *             force kernels ms, launches,   behind-the-force-pass kernels (summation, update, kicks) ms, launches,
 *             auxiliary-stream kernels ms, launches,
 *             position exchange on the communication stream ms, count,
 *             position exchange as the waiting force launch saw it -- from the end of the previous update (the moment the
 *               launch could have started) to the arrival of the rows -- ms, count (ring: per hop),
 *             pair-once column-sum exchange ms, count (on the compute stream: not hidden, by design),
 *             layout refreshes ms, count }
 * sums of event-pair durations, not wall time. */
int nbody_multi_timing_enable(nbody_multi *m, int on);
int nbody_multi_timing_read(nbody_multi *m, int local_index, double *out16);
/* Local rank i's shard context (timing, device info, kernel selection) and device buffers; borrowed. */
nbody_ctx *nbody_multi_shard(nbody_multi *m, int local_index);
float *nbody_multi_positions_device(nbody_multi *m, int local_index);
float *nbody_multi_velocities_device(nbody_multi *m, int local_index);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_H */
