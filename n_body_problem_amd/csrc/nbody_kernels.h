// nbody_kernels.h -- launch interface between the C ABI (nbody_capi.hip) and the gfx950 kernels
// (nbody_kernels.hip).  Internal; the public surface is include/nbody.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbody {

constexpr int kTile = 256;  // bodies per LDS tile == threads per workgroup (reference BLOCK_SIZE, kernel.cu:65)

// Zero-distance guard of the eps == 0 kernel variants: a pair whose r^2 is below 2^-84 (closer than 2.3e-13: the self
// pair, coincident bodies, and bodies that only rounding has moved apart) contributes exactly 0 for ANY finite mass --
// r^2 is replaced by +inf, so v_rsq_f32 returns 0 and every later product is 0.  A clamp of r^2 instead makes m * inv^3
// overflow for heavy bodies and 0 * inf = NaN; a threshold at the smallest normal float would still let inv^3 = r^-3
// overflow (inv up to 1e19) and a ZERO-mass column turn that into 0 * inf = NaN -- which the library's own padding
// bodies met: zero-mass bodies at one point stay coincident under the one-sided kernels (every row sums its columns in
// the same order) but drift apart by an ulp under the pair-once tiles (a lane's column order depends on the lane).
// With r^2 >= 2^-84, inv <= 2^42 and inv^3 <= 2^126 is finite.  One v_cmp + one v_cndmask.
constexpr float kGuardMin = 0x1p-84f;
__device__ __forceinline__ float guard_r2(float r2) { return r2 >= kGuardMin ? r2 : __builtin_inff(); }

struct ForceArgs {
    const float4 *pos;   // all n_total bodies {x,y,z,m}
    float4 *partials;    // [n_splits][row_count] partial accelerations {ax,ay,az,unused}
    int row_lo;          // first row (global body index) of this context
    int row_count;       // rows of this context
    int n_total;         // columns
    int split_len;       // columns per split (multiple of kTile)
    int split_first;     // first split computed by this launch
    int split_count;     // splits computed by this launch (grid.y)
    int skip_first;      // splits [skip_first, skip_first+skip_count) are stepped over (a launch that covers
    int skip_count;      //   "every split except a range": split = split_first + y, += skip_count once >= skip_first
    float eps2;          // softening length squared
    const float *eps_pp; // optional per-particle softening lengths (n_total floats): eps_ij^2 = eps2 + eps_i^2 + eps_j^2
    const float *split_mass;  // [n_splits]: the one mass of a split's bodies, or NaN (launch_split_mass)
    int own_split_mass;       // force_kernel_r4pk_w1, one-tile splits: the kernel forms that flag itself (nothing was launched)
};

// Pair-once kernel (nbody_symmetric.hip): one workgroup per ordered pair of splits (R, C), R's bodies as rows (one
// context's own rows), C's bodies as columns; each unordered pair {R, C} is computed once, by the owner of the side
// sym_rows_side() names.
constexpr int kSymGroups = 8;  // the canonical summation: 8 groups of ceil(n_splits / 8) splits, see sym_finalize

// True when the tile of the unordered split pair {R, C} (R != C) is computed with R's bodies as rows: the "forward
// half" of the ring of S splits, so every split is the row side of (S - 1) / 2 tiles -- equal work for every rank that
// owns equally many splits.  A pure function of (R, C, S): the summation kernels use it to know which partial sums exist.
__host__ __device__ inline bool sym_rows_side(int R, int C, int S)
{
    int d = C - R;
    if (d < 0)
        d += S;
    if (d == 0)
        return false;
    if (2 * d != S)
        return 2 * d < S;
    const int lo = R < C ? R : C;  // S even, opposite splits: alternate
    return ((lo & 1) == 0) == (R == lo);
}

// Strips (round 4).  A workgroup takes `strip_len` = K consecutive column splits of one row split -- the splits C with equal
// C / K that form a tile with R (K = 1: every tile alone, rounds 1-3) -- and keeps the rows' sums in registers across them: one
// row-side partial sum per (row, strip) instead of per (row, tile).  The strip that holds column split C is slot
// sym_row_slot(R, C) of the row split's array: the blocks of K splits are counted along the ring from the block of R + 1 (K = 1:
// the ring distance of the tile, the layout of rounds 1-3); slot 0 is the diagonal tile.  Blocks are absolute (C / K), the
// number of splits is a multiple of 8 K, so a strip never straddles a summation group or a rank's column chunk: which sums
// exist and in which order they are added is a function of (n_total, split_len) only, as before.
__host__ __device__ inline int sym_row_slot(int R, int C, int S, int K)
{
    int j = C / K - ((R + 1) % S) / K;
    if (j < 0)
        j += S / K;
    return j + 1;
}
__host__ __device__ inline int sym_row_slots(int S, int K) { return K == 1 ? S / 2 + 1 : S / (2 * K) + 3; }

struct SymArgs {
    const float4 *pos;     // all n_total bodies
    // Both arrays are indexed by the ring distance d = (C - R) mod S of the tile, 0 <= d <= S/2 (0: the diagonal), so
    // they hold exactly the partial sums that exist:
    // (12-byte entries {x, y, z}: a quarter fewer bytes than float4 for the 2 x n_total^2 / split_len entries a step writes
    // and the summation kernels read back)
    float3 *row_partials;  // [sym_row_slots][row_count]: P_row[slot][b] = force on own body b (split R) from the bodies of the strip
                           // in that slot (strip_len 1: slot = d, the bodies of split R + d)
    float3 *col_partials;  // [own splits][S/2][split_len]: P_col[R][d - 1][i] = force on body i of split R + d from own split R
    const int4 *tiles;     // n_tiles strips {R, first C, count, slot}: R an own split, C ... C + count - 1 its column splits
    int n_tiles;
    int strip_len;         // K
    const int2 *diag_tiles;  // n_diag pairs (B, B), own splits: every pair inside the split once, both sides (the quarter-tile kernel
                             // of small systems: every ordered pair, row side only -- the same sums, P_row[0][b])
    int n_diag;
    int n_total;
    int split_len;         // 256 <= split_len <= 4096, multiple of 256
    int row_lo, row_count; // the context's own rows (whole splits)
    float eps2;
    const float *eps_pp;   // optional per-particle softening lengths (n_total floats): eps_ij^2 = eps2 + eps_i^2 + eps_j^2
    const float *split_mass;  // [n_splits]: the one mass of a split's bodies, or NaN (launch_split_mass)
    int equal_mass_path;      // 0: no tile takes the equal-mass loops (nbody_set_equal_mass_path; the quarter-tile kernel decides itself)
    int packed;               // 0: one-column loops; 1: packed two-columns-per-step loops, four rows per lane; 2: and eight rows
                              // per lane on equal-mass tiles of splits of whole 1024 bodies; 3 (default): eight rows per lane
                              // on every tile of such splits (one kernel, allocated for three waves per SIMD)
};
// Small systems (256- and 512-body splits, the packed loops; not per-particle softening with eps = 0, where a pair may meet at r^2 = 0
// unguarded, and not per-particle softening at all with 512-body splits): the tiles AND the diagonal tiles are served by force_sym_quarter_kernel in ONE launch (launch_forces_symmetric),
// which needs no split_mass flags.
inline bool sym_quarter_tiles(int split_len, float eps2, const float *eps_pp, int packed)
{
    if (split_len == 512)  // eight waves per tile; per-particle softening keeps the eight-row loops (S10 / S12) of force_sym_kernel there
        return packed >= 2 && !eps_pp;
    return split_len == 256 && packed >= 2 && !(eps_pp && !(eps2 > 0.f));
}
// split_mass[s] for every split of the body set, from the masses now in pos (O(N); see split_mass_kernel)
hipError_t launch_split_mass(const float4 *pos, float *split_mass, int n_total, int split_len, bool enabled, hipStream_t stream);
hipError_t launch_forces_symmetric(const SymArgs &a, hipStream_t stream);       // the tiles (R != C); sym_quarter_tiles(): and the diagonal tiles
hipError_t launch_forces_symmetric_diag(const SymArgs &a, hipStream_t stream);  // the diagonal tiles (sym_quarter_tiles(): nothing left to do)
size_t symmetric_lds_bytes(int split_len);

// colparts[g][c] = sum over the own splits R of group g (ascending, where the tile (R, C(c)) exists) of P_col[R][c],
// for the own groups [group_lo, group_lo + group_count) and every body c.  colparts is [kSymGroups][n_total].
hipError_t launch_sym_colparts(const float3 *col_partials, float4 *colparts, int n_total, int split_len, int n_splits,
                               int split_lo, int group_splits, int group_lo, int group_count, hipStream_t stream);
// rowsum[g][b] = sum over C in group g (ascending, where the tile (B(b), C) exists, and the diagonal C == B(b)) of
// P_row[d(B, C)][b] for the rows [row_lo, row_lo + row_count) that row_partials ([n_splits/2+1][row_count]) holds; rowsum
// points at the first of them in a [groups][out_stride] float4 array.
hipError_t launch_sym_rowsum(const float3 *row_partials, float4 *rowsum, int row_lo, int row_count, int split_len, int n_splits,
                             int group_splits, int out_stride, int strip_len, hipStream_t stream);
// acc[b] = sum over the groups g (ascending) of ( rowsum[g][b] + colparts[g][b] ): the same association for any number of ranks.
hipError_t launch_sym_combine(const float4 *rowsum, const float4 *colparts, float4 *acc, int row_lo, int row_count, int n_total,
                              int n_groups, hipStream_t stream);

// A context that owns every row, tiles in one part: column sums + row sums + combination + kick-drift in one launch (the same
// association as the three kernels above).
hipError_t launch_sym_finish_update(const float3 *row_partials, const float3 *col_partials, float4 *pos_all, float4 *vel_rows,
                                    int n_total, int split_len, int n_splits, int group_splits, float dt, hipStream_t stream);
// The same sums ending in kick-drift-kick's closing half kick (kick) or in the accelerations alone: acc[b] as launch_sym_combine
// leaves it and kdk_kick_kernel copies it.
hipError_t launch_sym_finish_kick(const float3 *row_partials, const float3 *col_partials, float4 *acc, float4 *vel_rows, int n_total,
                                  int split_len, int n_splits, int group_splits, float dt, bool kick, hipStream_t stream);

// Partial accelerations of rows [row_lo,row_lo+row_count) from splits [split_first, split_first+split_count).
// rows_per_lane in {1,2,4,8}.  eps2 == 0 selects the zero-distance-guarded variant.
hipError_t launch_forces(const ForceArgs &a, int rows_per_lane, hipStream_t stream);

// Sum the partials of n_splits splits in ascending order and kick-drift rows of this context.
hipError_t launch_update(float4 *pos_all, float4 *vel_rows, const float4 *partials, int row_lo, int row_count,
                         int n_splits, float dt, hipStream_t stream);

// Pair-once mode: the combination of the group sums (launch_sym_combine's association) and the kick-drift in one launch.
hipError_t launch_update_sym(float4 *pos_all, float4 *vel_rows, const float4 *rowsum, const float4 *colparts, int row_lo,
                             int row_count, int n_total, int n_groups, float dt, hipStream_t stream);

// Velocity-Verlet (kick-drift-kick) pieces with cached accelerations (the reference's historical
// variant, unused_files/backup.cu:859-887 driven at :1848-1866, spends two force evaluations per step).
//   reduce    : acc[r] = sum of the splits' partials, ascending                              (first step only)
//   kick_drift: v += acc*dt/2 ; x += v*dt                                                   (fp64 FMA, as the update)
//   kick      : acc[r] = sum of partials ; v += acc*dt/2
hipError_t launch_kdk_reduce(float4 *acc, const float4 *partials, int row_count, int n_splits, hipStream_t stream);
hipError_t launch_kdk_kick_drift(float4 *pos_all, float4 *vel_rows, const float4 *acc, int row_lo, int row_count, float dt,
                                 hipStream_t stream);
hipError_t launch_kdk_kick(float4 *vel_rows, float4 *acc, const float4 *partials, int row_count, int n_splits, float dt,
                           hipStream_t stream);

// positions[4i+3] = masses[i]
hipError_t launch_scatter_mass(float4 *pos_all, const float *masses, int n_total, hipStream_t stream);

// Per-block {kinetic, potential} doubles into block_out[2*gridDim.x]; returns the grid size used.
hipError_t launch_energy(const float4 *pos_all, const float4 *vel_rows, double *block_out, int row_lo,
                         int row_count, int n_total, float eps2, const float *eps_pp, hipStream_t stream);
int energy_blocks(int row_count);         // blocks of launch_momentum (one row per lane)
int energy_kernel_blocks(int row_count);  // blocks of launch_energy (four rows per lane)

// Per-block {px,py,pz,m} doubles into block_out[4*gridDim.x].
hipError_t launch_momentum(const float4 *pos_all, const float4 *vel_rows, double *block_out, int row_lo,
                           int row_count, hipStream_t stream);

// ---- body order on the device (nbody_order.hip) ----------------------------------------------------------------------
// perm[k] (k < n) = the slot whose body belongs at slot k (nbody_morton_order), as 32-bit indices; scratch holds
// order_scratch_bytes(n).
size_t order_scratch_bytes(int n);
// order (n int64 on the device, or NULL): the caller's index of the body in each slot -- ties between equal keys follow it
hipError_t launch_morton_order(const float4 *pos, int n, unsigned *perm, void *scratch, hipStream_t stream,
                               const int64_t *order = nullptr);
hipError_t launch_identity_perm(unsigned *perm, int first, int count, hipStream_t stream);  // perm[first + j] = first + j
// dst[j] = src[perm[first + j]], j < count
hipError_t launch_gather_float4(float4 *dst, const float4 *src, const unsigned *perm, int first, int count, hipStream_t stream);
hipError_t launch_gather_float(float *dst, const float *src, const unsigned *perm, int first, int count, hipStream_t stream);
hipError_t launch_gather_int64(int64_t *dst, const int64_t *src, const unsigned *perm, int first, int count, hipStream_t stream);
hipError_t launch_widen_perm(int64_t *dst, const unsigned *src, int n, hipStream_t stream);
hipError_t launch_iota64(int64_t *dst, int n, hipStream_t stream);
// perm[k] = order[k] (k < n), or the inverse: perm[order[k]] = k (order must then be a permutation of 0..n-1)
hipError_t launch_set_perm(unsigned *perm, const int64_t *order, int n, bool inverse, hipStream_t stream);

}  // namespace nbody
