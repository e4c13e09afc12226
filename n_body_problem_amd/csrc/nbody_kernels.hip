// nbody_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the all-pairs step.
//
// What the reference computes on this path (main_project/kernel.cu):
//   cal_acc_advanced          :703-774  per-body acceleration from all other bodies
//   use_acc_update_position   :777-801  v += a*dt; x += v*dt (fp64 FMA, rounded to fp32)
//   simple_update_all         :828-884  the GPU-Gems shape: 1 body per thread, 256-body shared tile
// Written from scratch for CDNA4: 64-lane wavefronts, column tiles staged in LDS and read back as
// wave-uniform (broadcast) ds_read_b128, several rows register-blocked per lane so each LDS read
// feeds rows_per_lane interactions, v_rsq_f32 for the inverse square root, no atomics (the
// reference's shared/global float atomics, kernel.cu:758-773, are what it names as its bottleneck).
// The kernel is VALU-bound; HBM traffic is the O(N) state plus the partials.
#include "nbody_kernels.h"

namespace nbody {

// One body-body interaction: 3 sub, 3 fma (r^2+eps^2), 1 rsq, 3 mul, 3 fma = 13 VALU instructions,
// counted as 20 flop.  The softening is folded into the r^2 FMA chain
// (kernel.cu:679 adds EPSILON separately, in double).
template <bool GUARD>
__device__ __forceinline__ void interact(float xi, float yi, float zi, const float4 pj, float eps2, float &ax,
                                         float &ay, float &az)
{
    const float dx = pj.x - xi;
    const float dy = pj.y - yi;
    const float dz = pj.z - zi;
    float r2 = __builtin_fmaf(dx, dx, eps2);
    r2 = __builtin_fmaf(dy, dy, r2);
    r2 = __builtin_fmaf(dz, dz, r2);
    if (GUARD)  // eps == 0: a pair at zero distance (the self pair) must contribute 0, not NaN
        r2 = __builtin_fmaxf(r2, 1.0e-24f);
    const float inv = __builtin_amdgcn_rsqf(r2);
    const float inv2 = inv * inv;
    const float s = (pj.w * inv) * inv2;  // m_j / (r^2+eps^2)^(3/2)
    ax = __builtin_fmaf(dx, s, ax);
    ay = __builtin_fmaf(dy, s, ay);
    az = __builtin_fmaf(dz, s, az);
}

// grid.x = row tiles of kTile*RPL rows, grid.y = splits of this launch.  Each lane owns RPL rows
// (row = tile_base + k*kTile + lane-in-block, so row loads/stores are coalesced float4) and walks the
// split's columns in ascending order, one fp32 FMA chain per row: the order bit-exactly defines the
// partial sum whatever the grid, RPL or sharding.
template <int RPL, bool GUARD>
__global__ __launch_bounds__(kTile) void force_kernel(ForceArgs a)
{
    __shared__ float4 tile[2][kTile];

    const int tid = threadIdx.x;
    int split = a.split_first + blockIdx.y;
    if (split >= a.skip_first)
        split += a.skip_count;
    const int j0 = split * a.split_len;
    const int j1 = min(j0 + a.split_len, a.n_total);
    const int ntiles = (j1 - j0 + kTile - 1) / kTile;
    const int row_base = blockIdx.x * (kTile * RPL) + tid;

    float xi[RPL], yi[RPL], zi[RPL], ax[RPL], ay[RPL], az[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = row_base + k * kTile;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < a.row_count)
            p = a.pos[a.row_lo + r];
        xi[k] = p.x;
        yi[k] = p.y;
        zi[k] = p.z;
        ax[k] = ay[k] = az[k] = 0.f;
    }

    // an out-of-range column is staged as a zero-mass body at the origin: it adds exactly 0
    float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 + tid < j1)
        stage = a.pos[j0 + tid];
    tile[0][tid] = stage;
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int jn = j0 + (t + 1) * kTile + tid;
        if (t + 1 < ntiles) {  // in flight under the tile's arithmetic
            stage = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jn < j1)
                stage = a.pos[jn];
        }

        const float4 *cur = tile[t & 1];
#pragma unroll 8
        for (int jj = 0; jj < kTile; ++jj) {
            const float4 pj = cur[jj];  // wave-uniform address: broadcast ds_read_b128
#pragma unroll
            for (int k = 0; k < RPL; ++k)
                interact<GUARD>(xi[k], yi[k], zi[k], pj, a.eps2, ax[k], ay[k], az[k]);
        }

        if (t + 1 < ntiles)
            tile[(t + 1) & 1][tid] = stage;
        __syncthreads();
    }

    float4 *out = a.partials + (size_t)split * a.row_count;
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = row_base + k * kTile;
        if (r < a.row_count)
            out[r] = make_float4(ax[k], ay[k], az[k], 0.f);
    }
}

template <int RPL>
static hipError_t launch_forces_rpl(const ForceArgs &a, hipStream_t stream)
{
    const int rows_per_block = kTile * RPL;
    dim3 grid((a.row_count + rows_per_block - 1) / rows_per_block, a.split_count, 1);
    if (a.eps2 > 0.f)
        hipLaunchKernelGGL((force_kernel<RPL, false>), grid, dim3(kTile), 0, stream, a);
    else
        hipLaunchKernelGGL((force_kernel<RPL, true>), grid, dim3(kTile), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_forces(const ForceArgs &a, int rows_per_lane, hipStream_t stream)
{
    if (a.row_count <= 0 || a.split_count <= 0)
        return hipSuccess;
    switch (rows_per_lane) {
    case 1: return launch_forces_rpl<1>(a, stream);
    case 2: return launch_forces_rpl<2>(a, stream);
    case 4: return launch_forces_rpl<4>(a, stream);
    case 8: return launch_forces_rpl<8>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

// use_acc_update_position, kernel.cu:777-801, with the reference's fp64 FMA (TIME_TICK is a double
// literal there) and the partial sums of the splits added first, in ascending split order.
__global__ __launch_bounds__(kTile) void update_kernel(float4 *pos_all, float4 *vel_rows, const float4 *partials,
                                                       int row_lo, int row_count, int n_splits, float dt)
{
    const int r = blockIdx.x * kTile + threadIdx.x;
    if (r >= row_count)
        return;
    float4 acc = partials[r];
    for (int s = 1; s < n_splits; ++s) {
        const float4 p = partials[(size_t)s * row_count + r];
        acc.x += p.x;
        acc.y += p.y;
        acc.z += p.z;
    }
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
    hipLaunchKernelGGL(update_kernel, dim3((row_count + kTile - 1) / kTile), dim3(kTile), 0, stream, pos_all,
                       vel_rows, partials, row_lo, row_count, n_splits, dt);
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

// Potential of each row against ALL columns (self pair excluded by index), fp32 pair terms, fp32 sum
// inside a tile, fp64 across tiles and across rows; kinetic energy of the rows.
__global__ __launch_bounds__(kTile) void energy_kernel(const float4 *pos_all, const float4 *vel_rows,
                                                       double *block_out, int row_lo, int row_count, int n_total,
                                                       float eps2)
{
    __shared__ float4 tile[kTile];
    const int tid = threadIdx.x;
    const int r = blockIdx.x * kTile + tid;
    const bool live = r < row_count;
    const int gi = row_lo + r;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live)
        pi = pos_all[gi];
    double phi = 0.0;
    for (int j0 = 0; j0 < n_total; j0 += kTile) {
        __syncthreads();
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j0 + tid < n_total)
            pt = pos_all[j0 + tid];
        tile[tid] = pt;
        __syncthreads();
        float s = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < kTile; ++jj) {
            const float4 pj = tile[jj];
            const float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
            const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, eps2)));
            const float inv = (j0 + jj != gi && r2 > 0.f) ? __builtin_amdgcn_rsqf(r2) : 0.f;
            s = __builtin_fmaf(pj.w, inv, s);
        }
        phi += (double)s;
    }
    double v[2] = {0.0, 0.0};
    if (live) {
        const float4 w = vel_rows[r];
        v[0] = 0.5 * (double)pi.w * ((double)w.x * w.x + (double)w.y * w.y + (double)w.z * w.z);
        v[1] = -0.5 * (double)pi.w * phi;
    }
    block_reduce_store<2>(v, block_out);
}

int energy_blocks(int row_count) { return (row_count + kTile - 1) / kTile; }

hipError_t launch_energy(const float4 *pos_all, const float4 *vel_rows, double *block_out, int row_lo,
                         int row_count, int n_total, float eps2, hipStream_t stream)
{
    if (row_count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(energy_kernel, dim3(energy_blocks(row_count)), dim3(kTile), 0, stream, pos_all, vel_rows,
                       block_out, row_lo, row_count, n_total, eps2);
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
