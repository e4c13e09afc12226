// nbody_capi.hip -- the C ABI of include/nbody.h: context, buffers, step sequencing, timing.
// Host side of the reference's buffer/interop boundary (main_project/kernel.cu:130-188, 1148-1160)
// and of its per-frame step bracket (kernel.cu:1225-1242), for a headless MI355X.
#include "../../include/nbody.h"
#include "nbody_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

using namespace nbody;

struct nbody_ctx {
    int device = 0;
    int64_t n_total = 0, row_lo = 0, row_count = 0, split_len = 0;
    int n_splits = 0;
    int rows_per_lane = 0;  // 0 = pick per launch
    bool equal_mass_path = true;  // splits whose bodies share one mass take the inner loop without mass multiplies
    int sum_parts = 0;  // pair-once mode, one context: launches the row groups are cut into (nbody_set_summation_parts)
    int force_mode = NBODY_FORCE_ONE_SIDED;
    int strip_setting = 0;  // nbody_set_strip_len: 0 = automatic
    int strip_len = 1;      // pair-once mode: column splits a tile workgroup takes with the rows' sums kept in registers (SymArgs)
    int integrator = NBODY_INTEGRATOR_KICK_DRIFT;
    float4 *acc = nullptr;      // kick-drift-kick mode: accelerations of the own rows at the current positions
    bool acc_valid = false;
    const float *eps_pp = nullptr;  // per-particle softening lengths in use, n_total floats (borrowed or eps_own), or NULL
    float *eps_own = nullptr;       // the copy nbody_upload_particle_softening made
    // pair-once mode (nbody_symmetric.hip)
    // One force call = one or more parts: a part is a run of whole row groups whose tiles go in one launch and whose partial
    // sums are added up (on the auxiliary stream, beside the next part's tiles) as soon as that launch is over.  With more than
    // two parts the partial sums live in two buffer slots used in turn: the arrays hold two parts, not the whole pass.
    struct SymPart {
        int g0 = 0, g1 = 0;              // row groups [g0, g1)
        int split_lo = 0, split_hi = 0;  // = row splits [split_lo, split_hi)
        int64_t b0 = 0, rows = 0;        // = rows [b0, b0 + rows) of this context
        int4 *tiles = nullptr;  // strips {R, first C, count, slot}
        int2 *diag = nullptr;
        int n_tiles = 0, n_diag = 0;
        size_t row_off = 0, col_off = 0;  // where the part's [n_splits/2+1][rows] and [splits][n_splits/2][split_len] arrays
                                          // start in partials / col_partials, in 12-byte entries
    };
    struct SymPlan { std::vector<SymPart> parts; size_t row_entries = 0, col_entries = 0; };
    std::map<std::tuple<int, int, bool>, SymPlan> sym_plans;  // per column range asked for
    const SymPart *pending = nullptr;  // the part of the last force call whose sums are still to be formed (the last one)
    std::vector<hipEvent_t> ev_tiles, ev_red;  // per part: tile launch over / its sums formed (the slot is free again)
    float3 *col_partials = nullptr;  // column-side partial sums, see SymArgs; sized by the plan (ensure_sym_buffers)
    size_t col_entries = 0;
    float4 *colparts = nullptr;      // [kSymGroups][n_total] in use: the caller's (nbody_sym_set_colparts) or colparts_own
    float4 *colparts_own = nullptr;
    float4 *sym_acc = nullptr;       // [row_count]: the summed accelerations the update kernels read as one split
    float4 *rowsum = nullptr;        // [kSymGroups][row_count]: row-side sums per column group (launch_sym_rowsum)
    float *split_mass = nullptr;     // [n_splits]: the one mass of each split's bodies or NaN (pair-once tiles' fast path)
    bool sym_reduced = false;        // nbody_sym_reduce has run since the last forces
    bool sym_rows_summed = false;    // ... and nbody_sym_rowsum (the row-side sums of the last part)
    int group_splits = 1, group_lo = 0, group_count = 0;
    int cu_count = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream work is enqueued on (own_stream or the caller's)
    // nbody_step_n on small systems: one step captured into a HIP graph and replayed (the inner loop is launch-bound)
    hipGraphExec_t step_graph = nullptr;
    struct GraphKey {
        const void *pos = nullptr, *vel = nullptr, *eps_pp = nullptr, *stream = nullptr;
        float dt = 0.f, softening = 0.f;
        int force_mode = -1, integrator = -1, rows_per_lane = 0;
        bool equal_mass = false;
        bool operator==(const GraphKey &o) const
        {
            return pos == o.pos && vel == o.vel && eps_pp == o.eps_pp && stream == o.stream && dt == o.dt &&
                   softening == o.softening && force_mode == o.force_mode && integrator == o.integrator &&
                   rows_per_lane == o.rows_per_lane && equal_mass == o.equal_mass;
        }
    } graph_key;
    int graph_replay = -1;  // -1: automatic (systems of at most kGraphAutoBodies bodies), 0: off, 1: on
    hipStream_t aux_stream = nullptr;  // pair-once mode: the diagonal-tile launch runs here, beside the tile launch
    hipStream_t tail_stream = nullptr; // pair-once mode, two summation parts: the LAST part's tiles, lowest priority, beside the first's
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_graph_in = nullptr, ev_graph_out = nullptr;
    hipEvent_t ev_flags = nullptr;  // the step's equal-mass flags have been written (a later force call may run on another stream)
    bool flags_valid = false;       // split_mass holds the flags of the positions of this step (launch_split_mass has run since the
                                    // partial sums were last consumed or forgotten)
    float4 *partials = nullptr;    // [n_splits][row_count]  (the reference's gravity_sum_array, kernel.cu:1148);
                                   // pair-once mode: [n_splits / 2 + 1][row_count].  Allocated at the first force call.
    size_t partials_entries = 0;
    float4 *pos = nullptr;         // owned position buffer (n_total), optional
    float4 *vel = nullptr;         // owned velocity buffer (row_count), optional
    double *reduce_dev = nullptr;  // per-block partials of the diagnostics kernels
    std::vector<double> reduce_host;
    std::vector<unsigned char> split_done;  // which splits nbody_forces has produced since the last update
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_force, ev_update, ev_aux;  // launches not yet added to the totals
    std::vector<hipEvent_t> ev_pool;
    double force_ms = 0, update_ms = 0, aux_ms = 0;  // aux: what the pair-once mode runs on the auxiliary stream BESIDE the
                                                      // tile launches (diagonal tiles, early summation)
    int64_t force_launches = 0, update_launches = 0, aux_launches = 0;
    // body order on the device (nbody_order.hip): the last permutation nbody_order_compute produced, the sort's scratch and
    // the buffer in-place gathers go through.  Allocated by the first use.
    unsigned *order_perm = nullptr;  // [n_total]: slot k <- body order_perm[k]; the identity beyond the n of the last compute
    int64_t order_n = -1;            // that n, or -1: no permutation yet
    void *order_scratch = nullptr, *order_tmp = nullptr;
    size_t order_scratch_bytes = 0, order_tmp_bytes = 0;
    std::string err;
};

static thread_local std::string g_create_error;

static int fail(nbody_ctx *c, int status, const std::string &msg)
{
    if (c)
        c->err = msg;
    else
        g_create_error = msg;
    return status;
}

static void free_sym_tiles(nbody_ctx *c);

// Small and mid-size pair-once systems (every row, the tiles in one part, nothing summed yet, up to 320 splits of single tiles):
// ONE finishing kernel forms the column sums, the row sums and their combination and ends in the update / the half kick.
static bool sym_fused_finish(const nbody_ctx *c)
{
    constexpr int fused_max_splits = 320;
    return c->force_mode == NBODY_FORCE_SYMMETRIC && c->row_lo == 0 && c->row_count == c->n_total && c->pending &&
           !c->sym_reduced && !c->sym_rows_summed && c->pending->g1 - c->pending->g0 == c->group_count &&
           c->pending->row_off == 0 && c->pending->col_off == 0 && c->n_splits <= fused_max_splits && c->strip_len == 1;
}

// SymArgs::packed of the context's rows-per-lane setting (nbody_set_rows_per_lane)
static int sym_packed(const nbody_ctx *c)
{
    int packed = c->rows_per_lane == 4 ? 1 : c->rows_per_lane == 2 ? 2 : c->rows_per_lane == 1 ? 0 : 3;
    if (packed >= 2 && !c->equal_mass_path)
        packed = 3;  // no tile can take the equal-mass loop
    return packed;
}

// Partial sums consumed or forgotten: the next force call starts a new step (and recomputes the equal-mass flags).
static void clear_split_done(nbody_ctx *c)
{
    std::fill(c->split_done.begin(), c->split_done.end(), 0);
    c->flags_valid = false;
}

// The captured step bakes in every device pointer the launches take (partial sums, tile lists, exchange buffer): whatever
// frees or replaces one of them drops the graph, and the next nbody_step_n captures a new one.
static void drop_step_graph(nbody_ctx *c)
{
    if (c->step_graph) {
        (void)hipGraphExecDestroy(c->step_graph);
        c->step_graph = nullptr;
    }
}

#define HIP_TRY(c, call)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail((c), e_ == hipErrorOutOfMemory ? NBODY_ERR_ALLOC : NBODY_ERR_DEVICE,               \
                        std::string(#call) + ": " + hipGetErrorString(e_));                                \
    } while (0)

extern "C" {

int nbody_abi_version(void) { return NBODY_ABI_VERSION; }

const char *nbody_status_string(int s)
{
    switch (s) {
    case NBODY_OK: return "ok";
    case NBODY_ERR_INVALID: return "invalid argument";
    case NBODY_ERR_ALLOC: return "allocation failed";
    case NBODY_ERR_DEVICE: return "HIP error";
    case NBODY_ERR_NO_DEVICE: return "no usable gfx950 device";
    case NBODY_ERR_STATE: return "context buffers not initialised";
    default: return "unknown status";
    }
}

const char *nbody_last_error(const nbody_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int64_t nbody_default_split_len(int64_t n_total)
{
    // Columns per partial sum.  A function of n_total ONLY: split boundaries define the summation order, so
    // they must not depend on the sharding.  n_total/128 rounded up to whole 256-body tiles, at most 8192
    // (32 LDS tiles): 128 splits up to 2^20 bodies (79 at the reference's 20 000), n_total/8192 beyond.
    // Many short splits keep the grid fine-grained -- at N = 65 536 a 256-column split measured 20 % faster
    // than a 4096-column one, and 8 ranks sharing N = 2^20 still get 128 row tiles x 128 splits each -- at
    // the price of 16 B x n_splits per row of partial sums (2 GiB at N = 2^20, ~0.4 % of the step time).
    if (n_total <= 0)
        return kTile;
    // Small systems (one-wave workgroups of 256 rows x one split, section "small systems" of DESIGN.md): the pass takes
    // ceil(waves / 1024) rounds on the chip's 1024 SIMDs, so 256-column splits leave the reference's own size -- 20 225 bodies:
    // 80 x 80 = 6400 waves, 6.25 per SIMD -- waiting for the SIMDs that got seven (profiles/r03_pmc_small_n_one_sided.txt: 83 % of
    // the large kernel's VALU share).  Where that decomposition needs three rounds or more the split length (a multiple of 64)
    // is the one (a multiple of 64 from 256 to 512: the one-wave kernel stages such a split whole) that minimises rounds x
    // length, rounds = ceil(rows-of-256 x splits / 1024), among the lengths that leave at least four rounds (four waves per
    // SIMD to hide each other's latencies): 20 225 and 20 000 bodies get 64 splits of 320 columns -- 5120 waves, five per SIMD,
    // 5 x 320 = 1600 columns per SIMD instead of 7 x 256 = 1792; measured 100.4 against 108.3 us per force pass, 0.108 against
    // 0.116 ms per step (profiles/r04_small_n_split.txt).  256 stays wherever nothing is strictly better (shorter splits were
    // tried: the partial sums they add cost the update more than the pass gains).  Still a function of n_total only.
    constexpr int64_t kOneWaveKernelBodies = 32768;  // below it a one-sided workgroup is one wave (force_kernel_r4pk_w1)
    const int64_t rb = (n_total + kTile - 1) / kTile;
    if (n_total < kOneWaveKernelBodies && rb * rb > 2048) {
        auto rounds = [&](int64_t L) { return ((n_total + L - 1) / L * rb + 1023) / 1024; };
        int64_t best = kTile;
        for (int64_t L = kTile + 64; L <= 2 * kTile; L += 64)
            if (rounds(L) >= 4 && rounds(L) * L < rounds(best) * best)
                best = L;
        return best;
    }
    int64_t len = (n_total + 127) / 128;
    len = (len + kTile - 1) / kTile * kTile;
    return len > 8192 ? 8192 : len;
}

int64_t nbody_pair_once_split_len(int64_t n_total)
{
    // 1024 = the eight-row kernel's rows per pass with two waves (four-row kernel: four waves): shorter splits idle waves,
    // longer ones coarsen the grid (N = 131072 on one GPU 3.6 / 4.6 / 6.2 ms with 1024 / 2048 / 512).  One pass writes
    // n_total^2 / split_len partial sums of 12 bytes into the two arrays: from N = 2^20 on the splits are 2048 bodies -- half
    // the partial sums (6.4 GB per pass at N = 2^20, held in 4 summation parts of which two exist at a time: 3.2 GB) and
    // 2.2 % less time per step than 1024-body splits in 8 parts at the same memory (156.0 against 159.5 ms, one GPU; one
    // rank of 8: 19.7 against 20.0 ms; profiles/r02_split_len_auto_parts_sustained.txt, r02_shard_rate_eight_rows_split_len.txt);
    // at N = 524288 one rank of 8 would lose 5 % to 2048 (a quarter of the tiles per rank), so 1024 stays below 2^20.
    // The length doubles again where 16-byte entries (round 1's layout: the bound is kept) would pass 150 GB: 4096 from
    // N = 2^23 (N = 2^22: 2048, 103 GB per pass, 26 GB held).  A function of n_total only: split boundaries define the
    // summation order.
    // Below that the splits shrink with the system (more, smaller tiles to fill the chip; one or two waves per workgroup):
    // 256 bodies up to 65 535, 512 up to 131 071 -- measured per size with the round-3 kernels, one GPU
    // (profiles/r03_split_len_mid_range.txt: N = 49 152: 0.436 / 0.479 / 0.544 ms per step with 256 / 512 / 1024; 65 536:
    // 0.722 / 0.715 / 0.840; 98 304: 1.582 / 1.524 / 1.591; 131 072: 2.79 / 2.66 / 2.62).  768 is no length for the tile
    // kernels (two waves cover 512 rows per pass: the second pass would run half empty and the equal-mass loops are off
    // where a pass is partial -- N = 196 608 ran at 8.88 ms per step with it, 5.79 with 1024).
    const double pairs16 = 16.0 * (double)n_total * (double)n_total;
    if (n_total < 65536)
        return kTile;
    if (n_total < 131072)
        return 2 * kTile;
    int64_t len;
    len = n_total >= ((int64_t)1 << 20) ? 2048 : 1024;
    while (len < 4096 && pairs16 / (double)len > 150e9)
        len *= 2;
    return len;
}

static int morton_order_impl(const float *xyzm, int64_t n, int64_t *perm);

int nbody_morton_order(const float *xyzm, int64_t n, int64_t *perm)
{
    if (n < 0 || (n > 0 && (!xyzm || !perm)))
        return NBODY_ERR_INVALID;
    try {
        return morton_order_impl(xyzm, n, perm);
    } catch (...) {  // host allocation
        return NBODY_ERR_ALLOC;
    }
}

static int morton_order_impl(const float *xyzm, int64_t n, int64_t *perm)
{
    // the bounding cube of the finite positions
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    bool any = false;
    for (int64_t i = 0; i < n; ++i) {
        const float *p = xyzm + 4 * i;
        if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])))
            continue;
        for (int a = 0; a < 3; ++a) {
            lo[a] = any ? std::min(lo[a], p[a]) : p[a];
            hi[a] = any ? std::max(hi[a], p[a]) : p[a];
        }
        any = true;
    }
    double extent = 0.0;
    for (int a = 0; a < 3; ++a)
        extent = std::max(extent, (double)hi[a] - (double)lo[a]);
    const double scale = extent > 0.0 ? 1048575.0 / extent : 0.0;  // 20 bits per axis
    auto spread = [](uint64_t v) {  // bit k of v to bit 3k
        v &= 0xfffffull;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    // few species: the mass first (rank of the mass among the distinct values), then the curve
    std::vector<uint32_t> species;
    bool few = true;
    uint32_t last_bits = 0;
    bool have_last = false;
    for (int64_t i = 0; i < n && few; ++i) {
        uint32_t bits;
        std::memcpy(&bits, xyzm + 4 * i + 3, 4);
        if (have_last && bits == last_bits)
            continue;
        last_bits = bits;
        have_last = true;
        if (std::find(species.begin(), species.end(), bits) == species.end()) {
            species.push_back(bits);
            few = species.size() <= NBODY_ORDER_MAX_SPECIES;
        }
    }
    if (few)
        std::sort(species.begin(), species.end(), [](uint32_t a, uint32_t b) {
            float fa, fb;
            std::memcpy(&fa, &a, 4);
            std::memcpy(&fb, &b, 4);
            return fa < fb || (!(fb < fa) && a < b);  // by value; NaNs and signed zeros by bit pattern
        });
    // key = species rank (4 bits) | curve (60 bits); bodies without a finite position last within their species.
    // Host threads over contiguous chunks (a refresh of the layout costs a run as much as this function takes).
    const int threads = (int)std::max<int64_t>(1, std::min<int64_t>({8, (int64_t)std::thread::hardware_concurrency(), n / 65536}));
    auto chunk = [&](int t) { return std::make_pair(n * t / threads, n * (t + 1) / threads); };
    auto in_parallel = [&](auto &&body) {  // chunks that get no thread run on this one: nothing is thrown across the C ABI
        std::vector<std::thread> pool;
        int started = 1;
        try {
            pool.reserve((size_t)threads);
            for (; started < threads; ++started)
                pool.emplace_back(body, started);
        } catch (...) {
        }
        body(0);
        for (int t = started; t < threads; ++t)
            body(t);
        for (auto &th : pool)
            th.join();
    };
    std::vector<uint64_t> keys((size_t)n), keys2((size_t)n);
    std::vector<int64_t> idx2((size_t)n);
    in_parallel([&](int t) {
        for (int64_t i = chunk(t).first; i < chunk(t).second; ++i) {
            const float *p = xyzm + 4 * i;
            const bool finite = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
            uint64_t curve = (1ull << 60) - 1;
            if (finite) {
                uint64_t q[3];
                for (int a = 0; a < 3; ++a)
                    q[a] = (uint64_t)std::min(1048575.0, std::max(0.0, ((double)p[a] - (double)lo[a]) * scale));
                curve = spread(q[0]) | spread(q[1]) << 1 | spread(q[2]) << 2;
            }
            uint64_t rank = 0;
            if (few) {
                uint32_t bits;
                std::memcpy(&bits, p + 3, 4);
                rank = (uint64_t)(std::find(species.begin(), species.end(), bits) - species.begin());
            }
            keys[(size_t)i] = rank << 60 | curve;
            perm[i] = i;
        }
    });
    // stable LSD radix sort, 16 bits a pass (ties keep the caller's order): per-thread histograms, then every thread
    // scatters its chunk behind the chunks before it
    std::vector<std::vector<int64_t>> count((size_t)threads, std::vector<int64_t>(65536));
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 16 * pass;
        in_parallel([&](int t) {
            std::fill(count[(size_t)t].begin(), count[(size_t)t].end(), 0);
            for (int64_t i = chunk(t).first; i < chunk(t).second; ++i)
                ++count[(size_t)t][(size_t)((keys[(size_t)i] >> shift) & 0xffff)];
        });
        int64_t at = 0;
        for (size_t d = 0; d < 65536; ++d)
            for (int t = 0; t < threads; ++t) {
                const int64_t c = count[(size_t)t][d];
                count[(size_t)t][d] = at;
                at += c;
            }
        in_parallel([&](int t) {
            for (int64_t i = chunk(t).first; i < chunk(t).second; ++i) {
                const int64_t to = count[(size_t)t][(size_t)((keys[(size_t)i] >> shift) & 0xffff)]++;
                keys2[(size_t)to] = keys[(size_t)i];
                idx2[(size_t)to] = perm[i];
            }
        });
        keys.swap(keys2);
        std::memcpy(perm, idx2.data(), sizeof(int64_t) * (size_t)n);
    }
    return NBODY_OK;
}

int64_t nbody_split_len(const nbody_ctx *ctx) { return ctx ? ctx->split_len : 0; }
int64_t nbody_n_total(const nbody_ctx *ctx) { return ctx ? ctx->n_total : 0; }

int nbody_create_shard(nbody_ctx **out, int device, int64_t n_total, int64_t row_lo, int64_t row_count,
                       int64_t split_len)
{
    if (!out)
        return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: out is NULL");
    *out = nullptr;
    if (n_total < 0 || n_total > (int64_t)1 << 30)
        return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: n_total out of range [0, 2^30]");
    if (row_lo < 0 || row_count < 0 || row_lo + row_count > n_total)
        return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: rows [row_lo,row_lo+row_count) not inside [0,n_total)");
    if (split_len == 0)
        split_len = nbody_default_split_len(n_total);
    if (split_len < 0 || split_len % 64 != 0)
        return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: split_len must be a positive multiple of 64 (of 256 in the pair-once mode)");
    if (row_lo % split_len != 0)
        return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: row_lo must be a multiple of split_len");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NBODY_ERR_NO_DEVICE,
                    std::string("nbody_create: no HIP device (") + hipGetErrorString(e) +
                        "); this library has no CPU path");
    if (device < 0 || device >= ndev)
        return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: device index out of range");
    HIP_TRY(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, NBODY_ERR_NO_DEVICE,
                    std::string("nbody_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

    nbody_ctx *c = new (std::nothrow) nbody_ctx;
    if (!c)
        return fail(nullptr, NBODY_ERR_ALLOC, "nbody_create: host allocation failed");
    c->device = device;
    c->n_total = n_total;
    c->row_lo = row_lo;
    c->row_count = row_count;
    c->split_len = split_len;
    c->n_splits = (int)((n_total + split_len - 1) / split_len);
    c->cu_count = prop.multiProcessorCount;
    c->split_done.assign((size_t)c->n_splits, 0);

    int rc = NBODY_OK;
    auto guard = [&](hipError_t he, const char *what) {
        if (he != hipSuccess && rc == NBODY_OK)
            rc = fail(nullptr, he == hipErrorOutOfMemory ? NBODY_ERR_ALLOC : NBODY_ERR_DEVICE,
                      std::string(what) + ": " + hipGetErrorString(he));
    };
    guard(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate");
    guard(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking), "hipStreamCreate");
    guard(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), "hipEventCreate");
    guard(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming), "hipEventCreate");
    guard(hipEventCreateWithFlags(&c->ev_graph_in, hipEventDisableTiming), "hipEventCreate");
    guard(hipEventCreateWithFlags(&c->ev_graph_out, hipEventDisableTiming), "hipEventCreate");
    guard(hipEventCreateWithFlags(&c->ev_flags, hipEventDisableTiming), "hipEventCreate");
    c->stream = c->own_stream;
    size_t red = (size_t)std::max(1, energy_blocks((int)row_count)) * 4;
    if (rc == NBODY_OK)
        guard(hipMalloc((void **)&c->reduce_dev, red * sizeof(double)), "hipMalloc(reduce)");
    if (rc == NBODY_OK)
        guard(hipMalloc((void **)&c->split_mass, sizeof(float) * (size_t)std::max(1, c->n_splits)), "hipMalloc(split_mass)");
    c->reduce_host.resize(red);
    if (rc != NBODY_OK) {
        nbody_destroy(c);
        return rc;
    }
    *out = c;
    return NBODY_OK;
}

int nbody_create(nbody_ctx **out, int device, int64_t n_total)
{
    return nbody_create_shard(out, device, n_total, 0, n_total, 0);
}

int nbody_create_auto(nbody_ctx **out, int device, int64_t n_total)
{
    const bool pair_once = n_total >= NBODY_PAIR_ONCE_MIN_BODIES;
    int rc = nbody_create_shard(out, device, n_total, 0, n_total, pair_once ? nbody_pair_once_split_len(n_total) : 0);
    if (rc != NBODY_OK)
        return rc;
    rc = nbody_set_force_mode(*out, pair_once ? NBODY_FORCE_SYMMETRIC : NBODY_FORCE_ONE_SIDED);
    if (rc != NBODY_OK) {
        g_create_error = (*out)->err;
        nbody_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

int nbody_force_mode(const nbody_ctx *c) { return c ? c->force_mode : NBODY_ERR_INVALID; }

int nbody_destroy(nbody_ctx *c)
{
    if (!c)
        return NBODY_OK;
    (void)hipSetDevice(c->device);
    if (c->own_stream)
        (void)hipStreamSynchronize(c->own_stream);
    if (c->aux_stream)
        (void)hipStreamSynchronize(c->aux_stream);
    for (auto &p : c->ev_force) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : c->ev_update) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : c->ev_aux) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->partials) (void)hipFree(c->partials);
    if (c->pos) (void)hipFree(c->pos);
    if (c->vel) (void)hipFree(c->vel);
    if (c->eps_own) (void)hipFree(c->eps_own);
    if (c->reduce_dev) (void)hipFree(c->reduce_dev);
    free_sym_tiles(c);
    if (c->col_partials) (void)hipFree(c->col_partials);
    if (c->colparts_own) (void)hipFree(c->colparts_own);
    if (c->sym_acc) (void)hipFree(c->sym_acc);
    if (c->rowsum) (void)hipFree(c->rowsum);
    for (auto &e : c->ev_tiles) (void)hipEventDestroy(e);
    for (auto &e : c->ev_red) (void)hipEventDestroy(e);
    if (c->split_mass) (void)hipFree(c->split_mass);
    if (c->acc) (void)hipFree(c->acc);
    if (c->order_perm) (void)hipFree(c->order_perm);
    if (c->order_scratch) (void)hipFree(c->order_scratch);
    if (c->order_tmp) (void)hipFree(c->order_tmp);
    if (c->step_graph) (void)hipGraphExecDestroy(c->step_graph);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_graph_in) (void)hipEventDestroy(c->ev_graph_in);
    if (c->ev_graph_out) (void)hipEventDestroy(c->ev_graph_out);
    if (c->ev_flags) (void)hipEventDestroy(c->ev_flags);
    if (c->tail_stream) (void)hipStreamDestroy(c->tail_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return NBODY_OK;
}

int nbody_set_stream(nbody_ctx *c, void *hip_stream)
{
    if (!c)
        return NBODY_ERR_INVALID;
    c->stream = (hipStream_t)hip_stream;  // verbatim: NULL is the HIP default stream
    return NBODY_OK;
}

int nbody_reset_stream(nbody_ctx *c)
{
    if (!c)
        return NBODY_ERR_INVALID;
    c->stream = c->own_stream;
    return NBODY_OK;
}

int nbody_sync(nbody_ctx *c)
{
    if (!c)
        return NBODY_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return NBODY_OK;
}

// ---- owned buffers --------------------------------------------------------------------------

int nbody_set_positions(nbody_ctx *c, const float *host)
{
    if (!c || !host)
        return fail(c, NBODY_ERR_INVALID, "nbody_set_positions: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->pos && c->n_total)
        HIP_TRY(c, hipMalloc((void **)&c->pos, sizeof(float4) * (size_t)c->n_total));
    if (c->n_total) {
        HIP_TRY(c, hipMemcpyAsync(c->pos, host, sizeof(float4) * (size_t)c->n_total, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->acc_valid = false;
    return NBODY_OK;
}

int nbody_set_velocities(nbody_ctx *c, const float *host)
{
    if (!c || !host)
        return fail(c, NBODY_ERR_INVALID, "nbody_set_velocities: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->vel && c->row_count)
        HIP_TRY(c, hipMalloc((void **)&c->vel, sizeof(float4) * (size_t)c->row_count));
    if (c->row_count) {
        HIP_TRY(c, hipMemcpyAsync(c->vel, host, sizeof(float4) * (size_t)c->row_count, hipMemcpyHostToDevice,
                                  c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return NBODY_OK;
}

int nbody_download(nbody_ctx *c, float *host_pos, float *host_vel)
{
    if (!c)
        return NBODY_ERR_INVALID;
    if ((host_pos && !c->pos && c->n_total) || (host_vel && !c->vel && c->row_count))
        return fail(c, NBODY_ERR_STATE, "nbody_download: buffers were never set");
    HIP_TRY(c, hipSetDevice(c->device));
    if (host_pos && c->n_total)
        HIP_TRY(c, hipMemcpyAsync(host_pos, c->pos, sizeof(float4) * (size_t)c->n_total, hipMemcpyDeviceToHost,
                                  c->stream));
    if (host_vel && c->row_count)
        HIP_TRY(c, hipMemcpyAsync(host_vel, c->vel, sizeof(float4) * (size_t)c->row_count, hipMemcpyDeviceToHost,
                                  c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NBODY_OK;
}

float *nbody_positions_device(nbody_ctx *c) { return c ? reinterpret_cast<float *>(c->pos) : nullptr; }
float *nbody_velocities_device(nbody_ctx *c) { return c ? reinterpret_cast<float *>(c->vel) : nullptr; }

// ---- timing ---------------------------------------------------------------------------------

static hipError_t get_event(nbody_ctx *c, hipEvent_t *e)
{
    if (!c->ev_pool.empty()) {
        *e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return hipSuccess;
    }
    return hipEventCreate(e);
}

typedef std::vector<std::pair<hipEvent_t, hipEvent_t>> EventPairs;

// Adds the finished launches at the head of the list to the totals and returns their events to the pool (all of them,
// waiting for each stop event, when `wait`): a long run with timing on holds a bounded number of live events.
static int drain(nbody_ctx *c, EventPairs &l, double &ms, int64_t &n, bool wait)
{
    size_t done = 0;
    for (; done < l.size(); ++done) {
        auto &p = l[done];
        if (wait)
            HIP_TRY(c, hipEventSynchronize(p.second));  // whichever stream the launch ran on
        else if (hipEventQuery(p.second) != hipSuccess)
            break;
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, p.first, p.second));
        ms += t;
        ++n;
        c->ev_pool.push_back(p.first);
        c->ev_pool.push_back(p.second);
    }
    l.erase(l.begin(), l.begin() + (ptrdiff_t)done);
    return NBODY_OK;
}

struct TimedLaunch {
    nbody_ctx *c;
    EventPairs *list;
    double *ms;
    int64_t *count;
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
    TimedLaunch(nbody_ctx *ctx, EventPairs *l, double *total_ms, int64_t *launches, hipStream_t on = nullptr, bool other = false)
        : c(ctx), list(l), ms(total_ms), count(launches), stream(other ? on : ctx->stream)
    {
        if (!c->timing)
            return;
        if (list->size() >= 64)
            (void)drain(c, *list, *ms, *count, false);
        if (get_event(c, &a) != hipSuccess) {
            a = nullptr;
            return;
        }
        if (get_event(c, &b) != hipSuccess) {
            c->ev_pool.push_back(a);
            a = b = nullptr;
            return;
        }
        (void)hipEventRecord(a, stream);
    }
    ~TimedLaunch()
    {
        if (a && b) {
            (void)hipEventRecord(b, stream);
            list->emplace_back(a, b);
        }
    }
};

int nbody_timing_enable(nbody_ctx *c, int on)
{
    if (!c)
        return NBODY_ERR_INVALID;
    c->timing = on != 0;
    return NBODY_OK;
}

int nbody_timing_read_ex(nbody_ctx *c, double *out6)
{
    if (!c || !out6)
        return NBODY_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = drain(c, c->ev_force, c->force_ms, c->force_launches, true);
    if (rc == NBODY_OK)
        rc = drain(c, c->ev_update, c->update_ms, c->update_launches, true);
    if (rc == NBODY_OK)
        rc = drain(c, c->ev_aux, c->aux_ms, c->aux_launches, true);
    if (rc != NBODY_OK)
        return rc;
    out6[0] = c->force_ms;
    out6[1] = (double)c->force_launches;
    out6[2] = c->update_ms;
    out6[3] = (double)c->update_launches;
    out6[4] = c->aux_ms;
    out6[5] = (double)c->aux_launches;
    c->force_ms = c->update_ms = c->aux_ms = 0;
    c->force_launches = c->update_launches = c->aux_launches = 0;
    return NBODY_OK;
}

int nbody_timing_read(nbody_ctx *c, double *force_ms, int64_t *force_launches, double *update_ms,
                      int64_t *update_launches)
{
    double v[6];
    int rc = nbody_timing_read_ex(c, v);
    if (rc != NBODY_OK)
        return rc;
    if (force_ms) *force_ms = v[0];
    if (force_launches) *force_launches = (int64_t)v[1];
    if (update_ms) *update_ms = v[2];
    if (update_launches) *update_launches = (int64_t)v[3];
    return NBODY_OK;
}

// ---- the step -------------------------------------------------------------------------------

// A context that owns every row may change its split length while no partial sums are pending (NBODY_FORCE_AUTO: the two
// force modes want different ones): everything sized or indexed by splits is released and rebuilt by the next force call.
static int resplit(nbody_ctx *c, int64_t split_len)
{
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->own_stream));
    HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    free_sym_tiles(c);
    if (c->partials) (void)hipFree(c->partials);
    if (c->col_partials) (void)hipFree(c->col_partials);
    c->partials = nullptr;
    c->col_partials = nullptr;
    c->partials_entries = c->col_entries = 0;
    c->split_len = split_len;
    c->n_splits = (int)((c->n_total + split_len - 1) / split_len);
    c->split_done.assign((size_t)c->n_splits, 0);
    c->flags_valid = false;
    if (c->split_mass) (void)hipFree(c->split_mass);
    c->split_mass = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->split_mass, sizeof(float) * (size_t)std::max(1, c->n_splits)));
    c->force_mode = NBODY_FORCE_ONE_SIDED;  // the pair-once geometry (groups of splits) is set up again by the caller
    c->sym_reduced = c->sym_rows_summed = false;
    c->acc_valid = false;
    return NBODY_OK;
}

static int sym_strip_len(const nbody_ctx *c);

int nbody_set_force_mode(nbody_ctx *c, int mode)
{
    if (!c || (mode != NBODY_FORCE_ONE_SIDED && mode != NBODY_FORCE_SYMMETRIC && mode != NBODY_FORCE_AUTO))
        return fail(c, NBODY_ERR_INVALID, "nbody_set_force_mode: unknown mode");
    if (mode == NBODY_FORCE_AUTO) {  // the faster mode for this body count, with the split length that mode wants
        if (c->row_lo != 0 || c->row_count != c->n_total)
            return fail(c, NBODY_ERR_INVALID, "nbody_set_force_mode: NBODY_FORCE_AUTO needs a context that owns every row (the "
                                              "split length of a shard is part of the sharding)");
        mode = c->n_total >= NBODY_PAIR_ONCE_MIN_BODIES ? NBODY_FORCE_SYMMETRIC : NBODY_FORCE_ONE_SIDED;
        const int64_t want = mode == NBODY_FORCE_SYMMETRIC ? nbody_pair_once_split_len(c->n_total) : nbody_default_split_len(c->n_total);
        if (want != c->split_len) {
            int rc = resplit(c, want);
            if (rc != NBODY_OK)
                return rc;
        }
    }
    if (mode == NBODY_FORCE_SYMMETRIC && c->force_mode != NBODY_FORCE_SYMMETRIC) {
        if (c->split_len < 256 || c->split_len > 4096 || c->split_len % kTile != 0)
            return fail(c, NBODY_ERR_INVALID,
                        "nbody_set_force_mode: the pair-once mode needs 256 <= split_len <= 4096, a multiple of 256 (create the context with "
                        "split_len = nbody_pair_once_split_len(n_total))");
        // the canonical summation: kSymGroups groups of ceil(n_splits / kSymGroups) splits; a context owns whole groups
        const int gs = std::max(1, (c->n_splits + kSymGroups - 1) / kSymGroups);
        const int split_lo = (int)(c->row_lo / c->split_len);
        const int split_hi = (int)((c->row_lo + c->row_count + c->split_len - 1) / c->split_len);
        const int64_t row_hi = c->row_lo + c->row_count;
        if (c->row_count && row_hi % c->split_len != 0 && row_hi != c->n_total)
            return fail(c, NBODY_ERR_INVALID, "nbody_set_force_mode: in the pair-once mode a context's rows must end on a split "
                                              "boundary or at n_total (a tile's row side is a whole split)");
        if (c->row_count && (split_lo % gs != 0 || (split_hi % gs != 0 && split_hi != c->n_splits)))
            return fail(c, NBODY_ERR_INVALID,
                        "nbody_set_force_mode: in the pair-once mode a context's rows must be whole groups of " +
                            std::to_string(gs) + " splits (n_splits / 8, rounded up)");
        c->group_splits = gs;
        c->group_lo = split_lo / gs;
        c->group_count = c->row_count ? (split_hi + gs - 1) / gs - c->group_lo : 0;
        c->strip_len = sym_strip_len(c);
        HIP_TRY(c, hipSetDevice(c->device));
        if (!c->sym_acc && c->row_count)
            HIP_TRY(c, hipMalloc((void **)&c->sym_acc, sizeof(float4) * (size_t)c->row_count));
        if (!c->rowsum && c->row_count)
            HIP_TRY(c, hipMalloc((void **)&c->rowsum, sizeof(float4) * (size_t)kSymGroups * (size_t)c->row_count));
        while (c->ev_tiles.size() < (size_t)kSymGroups) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(c, hipEventCreateWithFlags(&a, hipEventDisableTiming));
            c->ev_tiles.push_back(a);
            HIP_TRY(c, hipEventCreateWithFlags(&b, hipEventDisableTiming));
            c->ev_red.push_back(b);
        }
        if (!c->colparts) {
            if (!c->colparts_own && c->n_total)
                HIP_TRY(c, hipMalloc((void **)&c->colparts_own, sizeof(float4) * (size_t)kSymGroups * (size_t)c->n_total));
            c->colparts = c->colparts_own;
        }
        c->sym_reduced = c->sym_rows_summed = false;
    }
    c->force_mode = mode;
    clear_split_done(c);
    c->acc_valid = false;
    drop_step_graph(c);
    return NBODY_OK;
}

// Strips (SymArgs, nbody_kernels.h): from the 2048-body splits on -- N >= 2^20, where the partial sums are gigabytes -- a tile
// workgroup takes four consecutive column splits of its row split and keeps the rows' sums in registers across them: a quarter
// of the row-side partial sums (N = 2^20: 0.8 instead of 3.2 GB per pass; the column side keeps its 3.2 GB -- halving that as well
// takes a workgroup that owns a whole CU, measured 7 % slower: profiles/r04_ab_whole_cu_workgroup.txt).  The blocks of four are
// absolute and the number of splits must be a multiple of 8 x 4, so no strip straddles a summation group or a rank's column
// chunk and which sums exist stays a function of (n_total, split_len) alone; other split counts keep single tiles.
// Automatic: strips of FOUR column splits from 1024 splits on (N >= 2^21), of TWO below (N = 2^20).  A strip is one workgroup, and
// a rank of eight's share of an N = 2^20 pass is only 4088 strips of four on 768 workgroup slots -- 5.3 rounds of 3.5 ms, six in
// practice: 21.6 ms against 18.7 ms with single tiles and 20.0 ms with strips of two, while one GPU runs strips of two and of
// four equally fast (146.2 ms; profiles/r04_shard_rate.txt).  A function of (n_total, split_len) only, like the split length:
// the strips define the order of the row-side sums.
static int sym_strip_len(const nbody_ctx *c)
{
    const int want = c->strip_setting ? c->strip_setting : (c->split_len >= 2048 ? (c->n_splits >= 1024 ? 4 : 2) : 1);
    return want > 1 && c->n_splits % (kSymGroups * want) == 0 && c->split_len == 2048 ? want : 1;
}

int nbody_set_strip_len(nbody_ctx *c, int len)
{
    if (!c || !(len == 0 || len == 1 || len == 2 || len == 4 || len == 8))
        return fail(c, NBODY_ERR_INVALID, "nbody_set_strip_len: expected 0 (automatic), 1, 2, 4 or 8");
    if (c->strip_setting != len) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
        free_sym_tiles(c);  // the cached plans carry the strips
        clear_split_done(c);
    }
    c->strip_setting = len;
    if (c->force_mode == NBODY_FORCE_SYMMETRIC)
        c->strip_len = sym_strip_len(c);
    c->acc_valid = false;
    return NBODY_OK;
}

int nbody_sym_set_colparts(nbody_ctx *c, float *d_buf)
{
    if (!c)
        return NBODY_ERR_INVALID;
    drop_step_graph(c);
    if (d_buf) {
        c->colparts = reinterpret_cast<float4 *>(d_buf);
    } else {
        if (!c->colparts_own && c->n_total) {
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipMalloc((void **)&c->colparts_own, sizeof(float4) * (size_t)kSymGroups * (size_t)c->n_total));
        }
        c->colparts = c->colparts_own;
    }
    c->sym_reduced = c->sym_rows_summed = false;
    return NBODY_OK;
}

int nbody_sym_groups(const nbody_ctx *c, int64_t *group_lo, int64_t *group_count, int64_t *group_splits)
{
    if (!c || c->force_mode != NBODY_FORCE_SYMMETRIC)
        return NBODY_ERR_INVALID;
    if (group_lo) *group_lo = c->group_lo;
    if (group_count) *group_count = c->group_count;
    if (group_splits) *group_splits = c->group_splits;
    return NBODY_OK;
}

static int all_splits_done(nbody_ctx *c, const char *who);
extern "C" int nbody_step_n_on(nbody_ctx *c, float *d_pos, float *d_vel, int k, float dt, float softening);

int nbody_sym_reduce(nbody_ctx *c)
{
    if (!c || c->force_mode != NBODY_FORCE_SYMMETRIC)
        return fail(c, NBODY_ERR_INVALID, "nbody_sym_reduce: the context is not in the pair-once mode");
    int rc = all_splits_done(c, "nbody_sym_reduce");
    if (rc != NBODY_OK)
        return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->pending)
        return fail(c, NBODY_ERR_STATE, "nbody_sym_reduce: no force call since the last update");
    // the groups of the last part; the earlier parts' sums were formed beside the tile launches
    const nbody_ctx::SymPart &p = *c->pending;
    HIP_TRY(c, launch_sym_colparts(c->col_partials + p.col_off, c->colparts, (int)c->n_total, (int)c->split_len, c->n_splits,
                                   p.split_lo, c->group_splits, p.g0, p.g1 - p.g0, c->stream));
    c->sym_reduced = true;
    return NBODY_OK;
}

// The row-side sums of the last part's rows (every earlier part's were formed beside the tile launches).  They need this
// context's own partial sums only, so a sharded host runs them while the column-side sums are on the wire:
// forces -> nbody_sym_reduce -> [start the exchange] -> nbody_sym_rowsum -> [exchange lands] -> nbody_update.
static int sym_rowsum_on(nbody_ctx *c, hipStream_t stream)
{
    if (!c || c->force_mode != NBODY_FORCE_SYMMETRIC)
        return fail(c, NBODY_ERR_INVALID, "nbody_sym_rowsum: the context is not in the pair-once mode");
    int rc = all_splits_done(c, "nbody_sym_rowsum");
    if (rc != NBODY_OK)
        return rc;
    if (!c->pending)
        return fail(c, NBODY_ERR_STATE, "nbody_sym_rowsum: no force call since the last update");
    if (c->sym_rows_summed)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const nbody_ctx::SymPart &p = *c->pending;
    HIP_TRY(c, launch_sym_rowsum(reinterpret_cast<const float3 *>(c->partials) + p.row_off, c->rowsum + p.b0,
                                 (int)(c->row_lo + p.b0), (int)p.rows, (int)c->split_len, c->n_splits, c->group_splits,
                                 (int)c->row_count, c->strip_len, stream));
    c->sym_rows_summed = true;
    return NBODY_OK;
}

int nbody_sym_rowsum(nbody_ctx *c) { return sym_rowsum_on(c, c ? c->stream : nullptr); }

// Pair-once mode: what is still missing of the group sums -- the column-side sums of the last part (unless nbody_sym_reduce
// has run: a shard exchanges them first) and the row-side sums of its rows.  After it rowsum[g][b] and colparts[g][b] hold
// every group's two halves for the context's rows.
static int sym_group_sums(nbody_ctx *c, const char *who)
{
    if (!c->sym_reduced) {
        if (c->row_lo != 0 || c->row_count != c->n_total)
            return fail(c, NBODY_ERR_STATE, std::string(who) + ": pair-once mode on a shard: call nbody_sym_reduce and exchange "
                                                                 "the column sums first");
        // One context, both halves still to do (a pass launched in one part: with strips the whole summation stands behind the
        // force pass): the row-side sums on the auxiliary stream BESIDE the column-side sums -- two HBM-bound walks over
        // different arrays, 0.6 + 0.7 ms one after the other at N = 2^20.
        if (!c->sym_rows_summed) {
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            int rc = sym_rowsum_on(c, c->aux_stream);
            if (rc == NBODY_OK)
                rc = nbody_sym_reduce(c);
            if (rc != NBODY_OK)
                return rc;
            HIP_TRY(c, hipEventRecord(c->ev_join, c->aux_stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
            c->sym_reduced = c->sym_rows_summed = false;
            return NBODY_OK;
        }
        int rc = nbody_sym_reduce(c);
        if (rc != NBODY_OK)
            return rc;
    }
    int rc = nbody_sym_rowsum(c);
    if (rc != NBODY_OK)
        return rc;
    c->sym_reduced = c->sym_rows_summed = false;
    return NBODY_OK;
}

// The partial sums the update kernels add up: the [n_splits][row_count] array of the one-sided kernel, or, in the
// pair-once mode, the one "split" sym_combine produces from the row sums and the (exchanged) column sums.
static int summed_partials(nbody_ctx *c, const char *who, const float4 **partials, int *n_splits)
{
    int rc = all_splits_done(c, who);
    if (rc != NBODY_OK)
        return rc;
    if (c->force_mode != NBODY_FORCE_SYMMETRIC) {
        *partials = c->partials;
        *n_splits = c->n_splits;
        return NBODY_OK;
    }
    rc = sym_group_sums(c, who);
    if (rc != NBODY_OK)
        return rc;
    const int n_groups = (c->n_splits + c->group_splits - 1) / c->group_splits;
    HIP_TRY(c, launch_sym_combine(c->rowsum, c->colparts, c->sym_acc, (int)c->row_lo, (int)c->row_count, (int)c->n_total, n_groups,
                                  c->stream));
    *partials = c->sym_acc;
    *n_splits = 1;
    return NBODY_OK;
}

int nbody_set_particle_softening(nbody_ctx *c, const float *d_eps)
{
    if (!c)
        return NBODY_ERR_INVALID;
    c->eps_pp = d_eps;
    c->acc_valid = false;
    return NBODY_OK;
}

int nbody_upload_particle_softening(nbody_ctx *c, const float *h_eps)
{
    if (!c)
        return NBODY_ERR_INVALID;
    if (!h_eps)
        return nbody_set_particle_softening(c, nullptr);
    if (c->n_total == 0)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->eps_own && hipMalloc(&c->eps_own, sizeof(float) * (size_t)c->n_total) != hipSuccess)
        return fail(c, NBODY_ERR_ALLOC, "nbody_upload_particle_softening: hipMalloc failed");
    HIP_TRY(c, hipMemcpyAsync(c->eps_own, h_eps, sizeof(float) * (size_t)c->n_total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return nbody_set_particle_softening(c, c->eps_own);
}

int nbody_set_equal_mass_path(nbody_ctx *c, int on)
{
    if (!c)
        return NBODY_ERR_INVALID;
    c->equal_mass_path = on != 0;
    c->acc_valid = false;
    c->flags_valid = false;
    return NBODY_OK;
}

static void free_sym_tiles(nbody_ctx *c)
{
    for (auto &kv : c->sym_plans)
        for (auto &p : kv.second.parts) {
            if (p.tiles) (void)hipFree(p.tiles);
            if (p.diag) (void)hipFree(p.diag);
        }
    c->sym_plans.clear();
    c->pending = nullptr;
    drop_step_graph(c);
}

int nbody_set_summation_parts(nbody_ctx *c, int parts)
{
    if (!c || !(parts == 0 || parts == 1 || parts == 2 || parts == 4 || parts == 8))
        return fail(c, NBODY_ERR_INVALID, "nbody_set_summation_parts: expected 0 (default), 1, 2, 4 or 8");
    if (c->sum_parts != parts) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
        free_sym_tiles(c);  // the cached plans carry the cut into parts
        clear_split_done(c);
    }
    c->sum_parts = parts;
    c->acc_valid = false;
    return NBODY_OK;
}

int nbody_set_early_summation(nbody_ctx *c, int on) { return nbody_set_summation_parts(c, on ? 0 : 1); }

int64_t nbody_partial_sum_bytes(const nbody_ctx *c)
{
    return c ? (int64_t)(c->partials_entries * sizeof(float4) + c->col_entries * sizeof(float3)) : 0;
}

int nbody_set_rows_per_lane(nbody_ctx *c, int rpl)
{
    if (!c || !(rpl == 0 || rpl == 1 || rpl == 2 || rpl == 4 || rpl == -4 || rpl == 8 || rpl == 40 || rpl == 41))
        return fail(c, NBODY_ERR_INVALID, "nbody_set_rows_per_lane: expected 0, 1, 2, 4, 8, -4, 40 or 41");
    c->rows_per_lane = rpl;
    return NBODY_OK;
}

// Largest register blocking that still fills the chip evenly; 4 rows per lane (the hand-allocated kernel) is fastest
// once there are enough workgroups.  With short splits (<= 512 columns: one or two LDS tiles per workgroup) a
// workgroup is over quickly and what counts is how evenly the last ones spread: measured at the reference's N = 20 225
// (split 256) 1 row per lane 0.145 ms, 2: 0.152, 4: 0.156; from N = 32 768 on 4 wins (tools/small_n.py).  Speed only:
// each row's sum is the same FMA chain whatever the blocking.
static int pick_rows_per_lane(const nbody_ctx *c, int split_count)
{
    if (c->rows_per_lane)
        return c->rows_per_lane;
    const bool short_splits = c->split_len <= 512;
    const int64_t want[1] = {(short_splits ? 10LL : 4LL) * c->cu_count};
    const int64_t blocks4 = (c->row_count + (int64_t)kTile * 4 - 1) / ((int64_t)kTile * 4) * split_count;
    if (blocks4 >= want[0])
        return 4;
    // too few 1024-row workgroups: the same packed loop with one wave (256 rows) per workgroup -- at every size below (N =
    // 4096 ... 20 225: 29 / 34 / 59 / 78 / 121 us per step against 42 / 38 / 73 / 89 / 137 with the compiler-allocated one-row
    // kernel, profiles/r03_small_n_blocking*.txt).  Per-particle softening: the same two kernels with the softening term in
    // the loop (N = 20 225: 0.165 ms per step with the compiler-allocated one-row kernel it used to take).
    return 41;
}

// The partial-sum array of the one-sided mode (a mode switch may need a larger one).
static int ensure_partials(nbody_ctx *c)
{
    const size_t entries = (size_t)c->n_splits * (size_t)c->row_count;  // [n_splits][rows] float4
    if (entries <= c->partials_entries)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    drop_step_graph(c);  // a captured step would keep launching on the array freed here
    if (c->partials) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->own_stream));
        (void)hipFree(c->partials);
        c->partials = nullptr;
        c->partials_entries = 0;
    }
    if (hipMalloc((void **)&c->partials, sizeof(float4) * entries) != hipSuccess)
        return fail(c, NBODY_ERR_ALLOC, "partial sums: hipMalloc of " + std::to_string(sizeof(float4) * entries >> 20) +
                                            " MiB failed (a longer split_len needs less)");
    c->partials_entries = entries;
    return NBODY_OK;
}

// The two partial-sum arrays of the pair-once mode, in 12-byte entries: what the plan of a force call needs (the whole pass,
// or two parts of it).  They only grow.
static int ensure_sym_buffers(nbody_ctx *c, size_t row_entries, size_t col_entries)
{
    const size_t row_f4 = (row_entries * sizeof(float3) + sizeof(float4) - 1) / sizeof(float4);
    if (row_f4 <= c->partials_entries && col_entries <= c->col_entries)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    drop_step_graph(c);  // a captured step would keep launching on the arrays freed here
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->own_stream));
    HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    if (row_f4 > c->partials_entries) {
        if (c->partials)
            (void)hipFree(c->partials);
        c->partials = nullptr;
        c->partials_entries = 0;
        if (hipMalloc((void **)&c->partials, sizeof(float4) * row_f4) != hipSuccess)
            return fail(c, NBODY_ERR_ALLOC, "row-side partial sums: hipMalloc of " + std::to_string(sizeof(float4) * row_f4 >> 20) +
                                                " MiB failed (more summation parts or a longer split_len need less)");
        c->partials_entries = row_f4;
    }
    if (col_entries > c->col_entries) {
        if (c->col_partials)
            (void)hipFree(c->col_partials);
        c->col_partials = nullptr;
        c->col_entries = 0;
        if (hipMalloc((void **)&c->col_partials, sizeof(float3) * col_entries) != hipSuccess)
            return fail(c, NBODY_ERR_ALLOC, "column-side partial sums: hipMalloc of " +
                                                std::to_string(sizeof(float3) * col_entries >> 20) +
                                                " MiB failed (more summation parts or a longer split_len need less)");
        c->col_entries = col_entries;
    }
    return NBODY_OK;
}

static int forces_impl(nbody_ctx *c, const float *d_pos, int64_t col_lo, int64_t col_count, float softening,
                       bool complement, const char *who)
{
    if (!c || (!d_pos && c->n_total))
        return fail(c, NBODY_ERR_INVALID, std::string(who) + ": NULL argument");
    if (!(softening >= 0.f) || !std::isfinite(softening))
        return fail(c, NBODY_ERR_INVALID, std::string(who) + ": softening must be finite and >= 0");
    if (softening > 0.f && softening < NBODY_MIN_SOFTENING)
        return fail(c, NBODY_ERR_INVALID, std::string(who) + ": 0 < softening < 1e-9 would overflow fp32 (eps^-3 x mass in the "
                                                             "self pair); use 0 (zero-distance pairs then contribute nothing)");
    if (col_lo < 0 || col_count < 0 || col_lo + col_count > c->n_total || col_lo % c->split_len != 0 ||
        ((col_lo + col_count) % c->split_len != 0 && col_lo + col_count != c->n_total))
        return fail(c, NBODY_ERR_INVALID, std::string(who) + ": column range must be split-aligned and inside [0,n_total]");
    if (c->row_count == 0 || c->n_total == 0)
        return NBODY_OK;
    const int first = (int)(col_lo / c->split_len);
    const int count = (int)((col_count + c->split_len - 1) / c->split_len);
    if (complement ? count == c->n_splits : count == 0)
        return NBODY_OK;  // no column selected
    if (c->force_mode == NBODY_FORCE_SYMMETRIC) {
        // this context's tiles with a column split in the range asked for, cut into parts (cached per range)
        const int S = c->n_splits, L = (int)c->split_len;
        const int own_lo = (int)(c->row_lo / L), own_hi = (int)((c->row_lo + c->row_count + L - 1) / L);
        const bool whole = c->row_lo == 0 && c->row_count == c->n_total && !complement && first == 0 && count == S;
        if (!whole && c->pending && c->pending->g1 - c->pending->g0 != c->group_count)
            for (int sp = 0; sp < S; ++sp)
                if (c->split_done[(size_t)sp])
                    return fail(c, NBODY_ERR_STATE, std::string(who) + ": a column range after a call for all columns in several "
                                                                        "summation parts: the earlier parts are already summed");
        const int SL = c->strip_len;
        if (SL > 1 && (first % SL != 0 || ((first + count) % SL != 0 && first + count != S)))
            return fail(c, NBODY_ERR_INVALID, std::string(who) + ": with strips of " + std::to_string(SL) + " column splits a column "
                                                                  "range must start and end on a multiple of that many splits");
        auto key = std::make_tuple(first, count, complement);
        auto it = c->sym_plans.find(key);
        if (it == c->sym_plans.end()) {
            auto selected = [&](int C) { return (C >= first && C < first + count) != complement; };
            // Launch order = L2 locality (speed only; every tile has its own outputs).  The tile of row split R and ring
            // distance d has column split (R + d) mod S.  Blocks of 8 row splits x 8 distances touch 23 splits' bodies
            // instead of 128; workgroups are dealt round-robin to the 8 XCDs (MI355X_MICROARCH.md), so block k's tiles
            // take the launch slots congruent to k mod 8 and meet in one XCD's L2.
            constexpr int kDiagInTilesMinStrip = 1;  // strips longer than this carry the diagonal tiles in the tile launch
            const int B = 8;
            auto list_rows = [&](int r_lo, int r_hi) {  // the strips with a row split in [r_lo, r_hi), in launch order
                std::vector<int4> tiles;
                std::vector<std::vector<int4>> per_xcd(8);
                const int n_blocks = S / SL;  // SL = 1: a "block" is one column split, a strip one tile, its slot the ring distance
                int k = 0;
                for (int Rb = r_lo; Rb < r_hi; Rb += B)
                    for (int jb = 0; jb <= n_blocks / 2 + 1; jb += B, ++k)
                        for (int R = Rb; R < std::min(Rb + B, r_hi); ++R)
                            for (int j = jb; j < std::min(jb + B, n_blocks / 2 + 2); ++j) {
                                const int J = (((R + 1) % S) / SL + j) % n_blocks;  // the j-th block along the ring from R + 1
                                int C0 = -1, cnt = 0;
                                for (int C = J * SL; C < (J + 1) * SL; ++C)
                                    if (selected(C) && sym_rows_side(R, C, S)) {
                                        if (cnt == 0)
                                            C0 = C;
                                        ++cnt;
                                    }
                                // a block met twice along the ring (its head at the start of the half ring, its tail at the end)
                                // cannot happen: the half ring is shorter than the ring by far more than a block
                                if (cnt > 0)
                                    per_xcd[(size_t)k % per_xcd.size()].push_back(make_int4(R, C0, cnt, sym_row_slot(R, C0, S, SL)));
                            }
                // Strips: the DIAGONAL tiles ride in the tile launch, as full squares of the hand-scheduled loops that keep their
                // row side (slot 0) -- 512 one-tile workgroups more for the launch's tail at N = 2^20 instead of a compiler-scheduled
                // launch beside it (0.4 % more pair evaluations; profiles/r04_strips_ab.txt)
                if (SL > kDiagInTilesMinStrip)
                    for (int R = r_lo; R < r_hi; ++R)
                        if (selected(R))
                            per_xcd[(size_t)R % per_xcd.size()].push_back(make_int4(R, R, 1, 0));
                // whole strips first, the shorter ones of the band's edges behind them, longest first (the launch's tail is made
                // of ever shorter workgroups: three-tile strips, then two, then one -- 145.4 against 145.9 ms per N = 2^20 step
                // with the short ones in list order, profiles/r04_strips_ab.txt) -- inside each XCD's sequence, so that a block's
                // strips keep meeting in one L2 (a partition of the interleaved list shifted the launch slots: 4.6 instead of 1.4 GB
                // of fabric reads per N = 2^20 pass)
                for (auto &seq : per_xcd)
                    std::stable_sort(seq.begin(), seq.end(), [](const int4 &x, const int4 &y) { return x.z > y.z; });
                for (size_t j = 0, more = 1; more; ++j) {
                    more = 0;
                    for (auto &seq : per_xcd)
                        if (j < seq.size()) {
                            tiles.push_back(seq[j]);
                            more = 1;
                        }
                }
                return tiles;
            };
            // Summation parts (one context that owns every row, all columns in one call, a system large enough for several
            // launches): the canonical order is by row groups, so a part's sums can be formed as soon as its launch is over,
            // on the auxiliary stream beside the next part's tiles, and nothing about the result changes.  What stays behind
            // the force pass is the last part's share of the summation and the combination.  2 parts = 7 groups + 1 (one
            // extra launch tail, the arrays hold the whole pass); 4 or 8 equal parts keep two parts' arrays.
            const int n_groups = (S + c->group_splits - 1) / c->group_splits;
            int K = c->sum_parts;
            if (K == 0) {  // automatic: one launch is the fastest (profiles/r02_summation_parts_eight_rows.txt); more only for memory
                // the column side and the row side (a strip's rows are summed in registers: 1 / strip_len of the entries)
                const double pass_bytes = 6.0 * (double)c->n_total * (double)c->n_total / (double)L * (1.0 + 1.0 / (double)SL);
                // 4 parts of 3 + 3 + 1 + 1 groups hold two slots of three groups = 3/4 of the pass; 8 equal parts a quarter
                K = pass_bytes <= (double)NBODY_PARTIAL_SUM_BUDGET_BYTES ? 1 : 0.75 * pass_bytes <= (double)NBODY_PARTIAL_SUM_BUDGET_BYTES ? 4 : 8;
            }
            // below 32768 tiles (N = 2^18) an extra launch costs more than the summation it hides: the automatic choice takes
            // one part there (an explicit nbody_set_summation_parts is honoured at every size)
            const int64_t min_tiles = c->sum_parts == 0 ? 32768 : 0;
            if (!whole || (int64_t)S * S / 2 < min_tiles || n_groups < 2)
                K = 1;
            else if (K > 2 && n_groups % K != 0)
                K = 2;
            nbody_ctx::SymPlan plan;
            for (int p = 0; p < K; ++p) {
                nbody_ctx::SymPart part;
                if (K == 1) {
                    part.g0 = c->group_lo;
                    part.g1 = c->group_lo + c->group_count;
                } else if (K == 2) {
                    part.g0 = p ? n_groups - 1 : 0;
                    part.g1 = p ? n_groups : n_groups - 1;
                } else if (K == 4 && n_groups == 8) {
                    // 3 + 3 + 1 + 1 groups: what stays behind the force pass is the LAST part's summation (its partial sums are
                    // read back at HBM speed: 0.2 GB per group and per million bodies^2 / split_len), so the last part is one
                    // group, not two -- update_ms 0.50 -> 0.3 ms at N = 2^20 -- and the two slots hold three groups each
                    static const int cut[5] = {0, 3, 6, 7, 8};
                    part.g0 = cut[p];
                    part.g1 = cut[p + 1];
                } else {
                    part.g0 = p * (n_groups / K);
                    part.g1 = (p + 1) * (n_groups / K);
                }
                part.split_lo = std::max(own_lo, part.g0 * c->group_splits);
                part.split_hi = std::min(own_hi, part.g1 * c->group_splits);
                part.b0 = (int64_t)part.split_lo * L - c->row_lo;
                part.rows = std::min<int64_t>((int64_t)part.split_hi * L, c->row_lo + c->row_count) - (int64_t)part.split_lo * L;
                const size_t row_need = (size_t)sym_row_slots(S, SL) * (size_t)part.rows;
                const size_t col_need = (size_t)(part.split_hi - part.split_lo) * (size_t)(S / 2) * (size_t)L;
                if (K > 2) {  // two slots used in turn
                    plan.row_entries = std::max(plan.row_entries, 2 * row_need);
                    plan.col_entries = std::max(plan.col_entries, 2 * col_need);
                } else {
                    part.row_off = plan.row_entries;
                    part.col_off = plan.col_entries;
                    plan.row_entries += row_need;
                    plan.col_entries += col_need;
                }
                std::vector<int4> tiles = list_rows(part.split_lo, part.split_hi);
                std::vector<int2> diag;
                for (int R = part.split_lo; R < part.split_hi && SL <= kDiagInTilesMinStrip; ++R)
                    if (selected(R))
                        diag.push_back(make_int2(R, R));
                HIP_TRY(c, hipSetDevice(c->device));
                if (!tiles.empty()) {
                    HIP_TRY(c, hipMalloc((void **)&part.tiles, sizeof(int4) * tiles.size()));
                    HIP_TRY(c, hipMemcpy(part.tiles, tiles.data(), sizeof(int4) * tiles.size(), hipMemcpyHostToDevice));
                }
                if (!diag.empty()) {
                    HIP_TRY(c, hipMalloc((void **)&part.diag, sizeof(int2) * diag.size()));
                    HIP_TRY(c, hipMemcpy(part.diag, diag.data(), sizeof(int2) * diag.size(), hipMemcpyHostToDevice));
                }
                part.n_tiles = (int)tiles.size();
                part.n_diag = (int)diag.size();
                plan.parts.push_back(part);
            }
            if (K > 2)
                for (int p = 0; p < K; ++p) {
                    plan.parts[(size_t)p].row_off = (p & 1) * (plan.row_entries / 2);
                    plan.parts[(size_t)p].col_off = (p & 1) * (plan.col_entries / 2);
                }
            it = c->sym_plans.emplace(key, std::move(plan)).first;
        }
        const nbody_ctx::SymPlan &plan = it->second;
        {
            int rc = ensure_sym_buffers(c, plan.row_entries, plan.col_entries);
            if (rc != NBODY_OK)
                return rc;
        }
        const int K = (int)plan.parts.size();
        SymArgs sa;
        sa.pos = reinterpret_cast<const float4 *>(d_pos);
        sa.n_total = (int)c->n_total;
        sa.split_len = L;
        sa.eps2 = softening * softening;
        sa.eps_pp = c->eps_pp;
        sa.split_mass = c->split_mass;
        // 3 (default): packed two-columns-per-step loops, eight rows per lane on every tile of splits of whole 1024 bodies -- one
        // kernel holds the equal-mass loop (S8) and the loop for arbitrary masses (S9), allocated for three waves per SIMD, which
        // costs the equal-mass loop 0.1 % against its own four-waves kernel (144.33 / 144.39 against 144.17 / 144.19 ms per
        // N = 2^20 pass, profiles/r03_ab_packed3_equal_mass.txt) and gives every tile with arbitrary masses the eight-row loop
        // (164.0 against 176.2 ms with the four-row loop, profiles/r03_ab_general_mass_eight_rows.txt).  The other arrangements
        // stay reachable through nbody_set_rows_per_lane for A/B measurement in one process: 2 = round 2's (equal-mass tiles eight
        // rows in a four-waves kernel, the others four rows), 4 = four rows per lane everywhere, 1 = the one-column loops.
        sa.packed = sym_packed(c);
        sa.equal_mass_path = c->equal_mass_path ? 1 : 0;
        const bool quarter = sym_quarter_tiles(L, sa.eps2, sa.eps_pp, sa.packed);  // small systems: no flags, no diagonal launch
        auto part_args = [&](const nbody_ctx::SymPart &p) {
            sa.row_partials = reinterpret_cast<float3 *>(c->partials) + p.row_off;
            sa.col_partials = c->col_partials + p.col_off;
            sa.tiles = p.tiles;
            sa.n_tiles = p.n_tiles;
            sa.strip_len = c->strip_len;
            sa.diag_tiles = p.diag;
            sa.n_diag = p.n_diag;
            sa.row_lo = (int)(c->row_lo + p.b0);
            sa.row_count = (int)p.rows;
        };
        HIP_TRY(c, hipSetDevice(c->device));
        // the equal-mass flags once per step, by its first force call: a later call for other columns (the complement launch on
        // a second stream, the ring's chunks) would rewrite the same values under the kernels that read them
        if (quarter) {
            // the quarter-tile kernel reads the masses itself
        } else if (!c->flags_valid) {
            HIP_TRY(c, launch_split_mass(sa.pos, c->split_mass, sa.n_total, L, c->equal_mass_path, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev_flags, c->stream));
            c->flags_valid = true;
        } else {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_flags, 0));  // written on the stream of the step's first call
        }
        // The auxiliary stream, beside the tile launches: a part's diagonal tiles (pairs inside one split; their own slot of
        // the row-side array: 0.56 ms at N = 2^20 that no longer stands between the tiles and the summation), and, once the
        // part's tile launch is over, its column-side sums per group and the row-side sums of its rows.  With two slots the
        // diagonal tiles of part p + 2 follow the sums of part p on this stream, and the tile launch of part p + 2 waits for
        // them (they ran beside part p + 1's tiles).
        // (small systems, one part: the tile launch serves the diagonal tiles and nothing runs beside it -- no fork, no join:
        // an eagerly enqueued step at N = 1024 took 29 us with them against 18.5 us replayed as a graph)
        const bool aux_idle = quarter && K == 1;
        if (!aux_idle) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        }
        auto diag_launch = [&](int p) -> int {
            if (aux_idle)
                return NBODY_OK;
            part_args(plan.parts[(size_t)p]);
            TimedLaunch t(c, &c->ev_aux, &c->aux_ms, &c->aux_launches, c->aux_stream, true);  // reported separately
            HIP_TRY(c, launch_forces_symmetric_diag(sa, c->aux_stream));
            return NBODY_OK;
        };
        for (int p = 0; p < std::min(K, 2); ++p)
            if (int rc = diag_launch(p))
                return rc;
        // Two parts (7 groups + 1): the last part's tiles go on a stream of the LOWEST priority that does not wait for the first
        // part's -- the dispatcher serves the first part's workgroups while it has any and fills their tail with the second
        // part's, so the extra launch costs no tail, and the first part's summation (7/8 of it) runs beside the second part's
        // tiles instead of behind the pass.
        if (K == 2 && !c->tail_stream) {
            int least = 0, greatest = 0;
            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_TRY(c, hipStreamCreateWithPriority(&c->tail_stream, hipStreamNonBlocking, least));
        }
        for (int p = 0; p < K; ++p) {
            const nbody_ctx::SymPart &part = plan.parts[(size_t)p];
            hipStream_t ts = K == 2 && p == 1 ? c->tail_stream : c->stream;
            if (ts != c->stream)
                HIP_TRY(c, hipStreamWaitEvent(ts, c->ev_fork, 0));  // the positions and the equal-mass flags are in place
            if (K > 2 && p >= 2)
                HIP_TRY(c, hipStreamWaitEvent(ts, c->ev_red[(size_t)p - 2], 0));
            part_args(part);
            {
                TimedLaunch t(c, &c->ev_force, &c->force_ms, &c->force_launches, ts, true);  // the dominant kernel alone
                HIP_TRY(c, launch_forces_symmetric(sa, ts));
            }
            if (p == K - 1) {
                if (ts != c->stream) {  // the last part's sums are formed on the context's stream
                    HIP_TRY(c, hipEventRecord(c->ev_tiles[(size_t)p], ts));
                    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_tiles[(size_t)p], 0));
                }
                break;
            }
            HIP_TRY(c, hipEventRecord(c->ev_tiles[(size_t)p], ts));
            HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_tiles[(size_t)p], 0));
            {
                TimedLaunch t(c, &c->ev_aux, &c->aux_ms, &c->aux_launches, c->aux_stream, true);  // overlapped, like the diagonal
                HIP_TRY(c, launch_sym_colparts(c->col_partials + part.col_off, c->colparts, (int)c->n_total, L, S, part.split_lo,
                                               c->group_splits, part.g0, part.g1 - part.g0, c->aux_stream));
                HIP_TRY(c, launch_sym_rowsum(reinterpret_cast<const float3 *>(c->partials) + part.row_off, c->rowsum + part.b0,
                                             (int)(c->row_lo + part.b0), (int)part.rows, L, S, c->group_splits, (int)c->row_count,
                                             c->strip_len, c->aux_stream));
            }
            HIP_TRY(c, hipEventRecord(c->ev_red[(size_t)p], c->aux_stream));
            if (p + 2 < K)
                if (int rc = diag_launch(p + 2))
                    return rc;
        }
        if (!aux_idle) {
            HIP_TRY(c, hipEventRecord(c->ev_join, c->aux_stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
        c->pending = &plan.parts.back();
        for (int s = 0; s < c->n_splits; ++s)
            if ((s >= first && s < first + count) != complement)
                c->split_done[(size_t)s] = 1;
        c->sym_reduced = c->sym_rows_summed = false;
        return NBODY_OK;
    }
    {
        int rc = ensure_partials(c);
        if (rc != NBODY_OK)
            return rc;
    }
    ForceArgs a;
    a.split_mass = c->split_mass;
    a.pos = reinterpret_cast<const float4 *>(d_pos);
    a.partials = c->partials;
    a.row_lo = (int)c->row_lo;
    a.row_count = (int)c->row_count;
    a.n_total = (int)c->n_total;
    a.split_len = (int)c->split_len;
    a.eps2 = softening * softening;
    a.eps_pp = c->eps_pp;
    if (complement) {  // every split except [first, first+count)
        a.split_first = 0;
        a.split_count = c->n_splits - count;
        a.skip_first = first;
        a.skip_count = count;
    } else {
        a.split_first = first;
        a.split_count = count;
        a.skip_first = c->n_splits;  // never reached
        a.skip_count = 0;
    }
    if (a.split_count <= 0)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int rpl = pick_rows_per_lane(c, a.split_count);
    // the one-wave kernel forms the equal-mass flag of a split of one or two tiles itself (from the tile it holds and, for the
    // second, the masses in memory): no launch in front
    a.own_split_mass = rpl == 41 && a.split_len <= 2 * kTile && c->equal_mass_path;
    if (!a.own_split_mass) {  // once per step, see the pair-once branch
        if (!c->flags_valid) {
            HIP_TRY(c, launch_split_mass(a.pos, c->split_mass, a.n_total, a.split_len, c->equal_mass_path, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev_flags, c->stream));
            c->flags_valid = true;
        } else {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_flags, 0));
        }
    }
    {
        TimedLaunch t(c, &c->ev_force, &c->force_ms, &c->force_launches);
        HIP_TRY(c, launch_forces(a, rpl, c->stream));
    }
    for (int s = 0; s < c->n_splits; ++s)
        if ((s >= first && s < first + count) != complement)
            c->split_done[(size_t)s] = 1;
    return NBODY_OK;
}

int nbody_forces(nbody_ctx *c, const float *d_pos, int64_t col_lo, int64_t col_count, float softening)
{
    return forces_impl(c, d_pos, col_lo, col_count, softening, false, "nbody_forces");
}

int nbody_forces_complement(nbody_ctx *c, const float *d_pos, int64_t col_lo, int64_t col_count, float softening)
{
    return forces_impl(c, d_pos, col_lo, col_count, softening, true, "nbody_forces_complement");
}

int nbody_update(nbody_ctx *c, float *d_pos, float *d_vel, float dt)
{
    if (!c || ((!d_pos || !d_vel) && c->row_count))
        return fail(c, NBODY_ERR_INVALID, "nbody_update: NULL argument");
    if (!std::isfinite(dt))
        return fail(c, NBODY_ERR_INVALID, "nbody_update: dt must be finite");
    if (c->row_count == 0)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    {
        TimedLaunch t(c, &c->ev_update, &c->update_ms, &c->update_launches);
        // Small and mid-size systems (every row, the tiles in one part, nothing summed yet): one finishing kernel instead of
        // column sums + row sums + update (up to 320 splits: -4 % per step at N = 32 768, nothing from 131 072 on,
        // profiles/r03_ab_fused_finish.txt; the bits are the same either way).
        if (sym_fused_finish(c)) {
            int rc = all_splits_done(c, "nbody_update");
            if (rc != NBODY_OK)
                return rc;
            HIP_TRY(c, launch_sym_finish_update(reinterpret_cast<const float3 *>(c->partials), c->col_partials,
                                                reinterpret_cast<float4 *>(d_pos), reinterpret_cast<float4 *>(d_vel), (int)c->n_total,
                                                (int)c->split_len, c->n_splits, c->group_splits, dt, c->stream));
        } else if (c->force_mode == NBODY_FORCE_SYMMETRIC) {  // the combination of the group sums rides in the update kernel
            int rc = all_splits_done(c, "nbody_update");
            if (rc == NBODY_OK)
                rc = sym_group_sums(c, "nbody_update");
            if (rc != NBODY_OK)
                return rc;
            const int n_groups = (c->n_splits + c->group_splits - 1) / c->group_splits;
            HIP_TRY(c, launch_update_sym(reinterpret_cast<float4 *>(d_pos), reinterpret_cast<float4 *>(d_vel), c->rowsum, c->colparts,
                                         (int)c->row_lo, (int)c->row_count, (int)c->n_total, n_groups, dt, c->stream));
        } else {
            const float4 *partials;
            int n_splits;
            int rc = summed_partials(c, "nbody_update", &partials, &n_splits);
            if (rc != NBODY_OK)
                return rc;
            HIP_TRY(c, launch_update(reinterpret_cast<float4 *>(d_pos), reinterpret_cast<float4 *>(d_vel), partials,
                                     (int)c->row_lo, (int)c->row_count, n_splits, dt, c->stream));
        }
    }
    clear_split_done(c);
    return NBODY_OK;
}

// ---- kick-drift-kick (velocity Verlet) with cached accelerations ----------------------------------------------

static int all_splits_done(nbody_ctx *c, const char *who)
{
    for (int s = 0; s < c->n_splits; ++s)
        if (!c->split_done[(size_t)s])
            return fail(c, NBODY_ERR_STATE, std::string(who) + ": split " + std::to_string(s) +
                                                " has no partial sums (call nbody_forces for every column range first)");
    return NBODY_OK;
}

static int ensure_acc(nbody_ctx *c)
{
    if (!c->acc && c->row_count) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipMalloc((void **)&c->acc, sizeof(float4) * (size_t)c->row_count));
    }
    return NBODY_OK;
}

int nbody_set_integrator(nbody_ctx *c, int integrator)
{
    if (!c || (integrator != NBODY_INTEGRATOR_KICK_DRIFT && integrator != NBODY_INTEGRATOR_KDK))
        return fail(c, NBODY_ERR_INVALID, "nbody_set_integrator: unknown integrator");
    c->integrator = integrator;
    c->acc_valid = false;
    return NBODY_OK;
}

int nbody_invalidate_forces(nbody_ctx *c)
{
    if (!c)
        return NBODY_ERR_INVALID;
    c->acc_valid = false;
    clear_split_done(c);  // partial sums of other positions are no partial sums
    return NBODY_OK;
}

int nbody_kdk_prepare(nbody_ctx *c)
{
    if (!c)
        return NBODY_ERR_INVALID;
    if (c->row_count == 0) {
        c->acc_valid = true;
        return NBODY_OK;
    }
    if (sym_fused_finish(c)) {  // one kernel instead of column sums + row sums + combination + copy (the same bits)
        int rc = all_splits_done(c, "nbody_kdk_prepare");
        if (rc == NBODY_OK)
            rc = ensure_acc(c);
        if (rc != NBODY_OK)
            return rc;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, launch_sym_finish_kick(reinterpret_cast<const float3 *>(c->partials), c->col_partials, c->acc, nullptr,
                                          (int)c->n_total, (int)c->split_len, c->n_splits, c->group_splits, 0.f, false, c->stream));
        clear_split_done(c);
        c->acc_valid = true;
        return NBODY_OK;
    }
    const float4 *partials;
    int n_splits;
    int rc = summed_partials(c, "nbody_kdk_prepare", &partials, &n_splits);
    if (rc == NBODY_OK)
        rc = ensure_acc(c);
    if (rc != NBODY_OK)
        return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_kdk_reduce(c->acc, partials, (int)c->row_count, n_splits, c->stream));
    clear_split_done(c);
    c->acc_valid = true;
    return NBODY_OK;
}

int nbody_kdk_kick_drift(nbody_ctx *c, float *d_pos, float *d_vel, float dt)
{
    if (!c || ((!d_pos || !d_vel) && c->row_count))
        return fail(c, NBODY_ERR_INVALID, "nbody_kdk_kick_drift: NULL argument");
    if (!std::isfinite(dt))
        return fail(c, NBODY_ERR_INVALID, "nbody_kdk_kick_drift: dt must be finite");
    if (!c->acc_valid)
        return fail(c, NBODY_ERR_STATE, "nbody_kdk_kick_drift: no cached accelerations (nbody_forces + nbody_kdk_prepare first)");
    if (c->row_count == 0)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    {
        TimedLaunch t(c, &c->ev_update, &c->update_ms, &c->update_launches);
        HIP_TRY(c, launch_kdk_kick_drift(reinterpret_cast<float4 *>(d_pos), reinterpret_cast<float4 *>(d_vel), c->acc,
                                         (int)c->row_lo, (int)c->row_count, dt, c->stream));
    }
    c->acc_valid = false;  // the positions moved: a kick must follow
    return NBODY_OK;
}

int nbody_kdk_kick(nbody_ctx *c, float *d_vel, float dt)
{
    if (!c || (!d_vel && c->row_count))
        return fail(c, NBODY_ERR_INVALID, "nbody_kdk_kick: NULL argument");
    if (!std::isfinite(dt))
        return fail(c, NBODY_ERR_INVALID, "nbody_kdk_kick: dt must be finite");
    if (c->row_count == 0) {
        c->acc_valid = true;
        return NBODY_OK;
    }
    int rc = ensure_acc(c);
    if (rc != NBODY_OK)
        return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    {
        TimedLaunch t(c, &c->ev_update, &c->update_ms, &c->update_launches);
        if (sym_fused_finish(c)) {
            rc = all_splits_done(c, "nbody_kdk_kick");
            if (rc != NBODY_OK)
                return rc;
            HIP_TRY(c, launch_sym_finish_kick(reinterpret_cast<const float3 *>(c->partials), c->col_partials, c->acc,
                                              reinterpret_cast<float4 *>(d_vel), (int)c->n_total, (int)c->split_len, c->n_splits,
                                              c->group_splits, dt, true, c->stream));
        } else {
            const float4 *partials;
            int n_splits;
            rc = summed_partials(c, "nbody_kdk_kick", &partials, &n_splits);
            if (rc != NBODY_OK)
                return rc;
            HIP_TRY(c, launch_kdk_kick(reinterpret_cast<float4 *>(d_vel), c->acc, partials, (int)c->row_count, n_splits, dt,
                                       c->stream));
        }
    }
    clear_split_done(c);
    c->acc_valid = true;
    return NBODY_OK;
}

int nbody_step_async(nbody_ctx *c, float *d_pos, float *d_vel, const float *d_masses, float dt, float softening)
{
    if (!c)
        return NBODY_ERR_INVALID;
    if ((!d_pos && c->n_total) || (!d_vel && c->row_count))
        return fail(c, NBODY_ERR_INVALID, "nbody_step: NULL positions or velocities");
    if (d_masses) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, launch_scatter_mass(reinterpret_cast<float4 *>(d_pos), d_masses, (int)c->n_total, c->stream));
        c->acc_valid = false;
    }
    if (c->n_total == 0)
        return NBODY_OK;
    int rc;
    if (c->integrator == NBODY_INTEGRATOR_KDK) {
        if (c->row_lo != 0 || c->row_count != c->n_total)
            return fail(c, NBODY_ERR_INVALID, "nbody_step: a sharded context drives kick-drift-kick through nbody_kdk_* "
                                              "(the drift must be exchanged before the forces)");
        if (!c->acc_valid) {
            rc = nbody_forces(c, d_pos, 0, c->n_total, softening);
            if (rc == NBODY_OK)
                rc = nbody_kdk_prepare(c);
            if (rc != NBODY_OK)
                return rc;
        }
        rc = nbody_kdk_kick_drift(c, d_pos, d_vel, dt);
        if (rc == NBODY_OK)
            rc = nbody_forces(c, d_pos, 0, c->n_total, softening);
        if (rc == NBODY_OK)
            rc = nbody_kdk_kick(c, d_vel, dt);
        return rc;
    }
    rc = nbody_forces(c, d_pos, 0, c->n_total, softening);
    if (rc != NBODY_OK)
        return rc;
    return nbody_update(c, d_pos, d_vel, dt);
}

int nbody_step(nbody_ctx *c, float *d_pos, float *d_vel, const float *d_masses, float dt, float softening)
{
    int rc = nbody_step_async(c, d_pos, d_vel, d_masses, dt, softening);
    return rc == NBODY_OK ? nbody_sync(c) : rc;
}

// Measured (tools/graph_ab.py, profiles/r02_graph_replay_ab.txt): a graph launch costs ~10 us more than three kernels
// enqueued back to back, so the one-sided step (flags, forces, update) is FASTER eager at every size (N = 256: 29 against
// 40 us per step; N = 20 225: 143 against 153); the pair-once step is seven launches on two streams with events between
// them, and there the replay wins up to a few ten thousand bodies (N = 4096: 96 against 114 us; N = 20 225: 169 against
// 190; N = 65 536: 801 against 788).  Automatic = pair-once mode, at most this many bodies, and a step of more than two kernels.
constexpr int64_t kGraphAutoBodies = 32768;

int nbody_set_graph_replay(nbody_ctx *c, int mode)
{
    if (!c || mode < -1 || mode > 1)
        return fail(c, NBODY_ERR_INVALID, "nbody_set_graph_replay: expected -1 (automatic), 0 or 1");
    c->graph_replay = mode;
    return NBODY_OK;
}

int nbody_step_n(nbody_ctx *c, int k, float dt, float softening)
{
    if (!c || k < 0)
        return fail(c, NBODY_ERR_INVALID, "nbody_step_n: bad argument");
    if ((c->n_total && !c->pos) || (c->row_count && !c->vel))
        return fail(c, NBODY_ERR_STATE, "nbody_step_n: call nbody_set_positions and nbody_set_velocities first");
    return nbody_step_n_on(c, reinterpret_cast<float *>(c->pos), reinterpret_cast<float *>(c->vel), k, dt, softening);
}

// k steps on the given buffers, one synchronisation at the end.  The first step runs eagerly (it may allocate: partial
// sums, tile lists, and in the kick-drift-kick mode it computes the initial accelerations); from the second on, when the
// system is small, ONE step is captured from the stream into a HIP graph and the graph is launched k - 1 times: the same
// kernels with the same arguments in the same order, so the same bits, without the per-launch host work and with the
// dependencies resolved on the device.
int nbody_step_n_on(nbody_ctx *c, float *d_pos, float *d_vel, int k, float dt, float softening)
{
    if (!c || k < 0)
        return fail(c, NBODY_ERR_INVALID, "nbody_step_n: bad argument");
    const bool whole = c->row_lo == 0 && c->row_count == c->n_total;
    // (round 4: where the tile launch serves the diagonal tiles too -- sym_quarter_tiles, one part -- a pair-once step is two
    // kernels on one stream, and those are faster enqueued eagerly as well: N = 1024: 12.5 against 18.5 us per step, 20 225:
    // 85.9 against 91.4, profiles/r04_pair_once_small_n.txt)
    const bool two_kernels = c->force_mode == NBODY_FORCE_SYMMETRIC && c->sum_parts <= 1 &&  // (kick-drift-kick: three, fused finish)
                             sym_quarter_tiles((int)c->split_len, softening * softening, c->eps_pp, sym_packed(c));
    const bool want = c->graph_replay == 1 || (c->graph_replay == -1 && c->force_mode == NBODY_FORCE_SYMMETRIC &&
                                               c->n_total <= kGraphAutoBodies && !two_kernels);
    const bool use_graph = want && whole && !c->timing && k >= 3 && c->n_total > 0;
    int s = 0;
    if (use_graph) {
        int rc = nbody_step_async(c, d_pos, d_vel, nullptr, dt, softening);  // eager: allocations, caches, first forces
        if (rc != NBODY_OK)
            return rc;
        ++s;
        nbody_ctx::GraphKey key;
        key.pos = d_pos; key.vel = d_vel; key.eps_pp = c->eps_pp; key.stream = nullptr;
        key.dt = dt; key.softening = softening;
        key.force_mode = c->force_mode; key.integrator = c->integrator; key.rows_per_lane = c->rows_per_lane;
        key.equal_mass = c->equal_mass_path;
        HIP_TRY(c, hipSetDevice(c->device));
        // The graph lives on the context's own stream (the caller's may be the legacy default stream, which cannot be
        // captured); events order it behind the eager step and the caller's stream behind it.
        hipStream_t user = c->stream;
        if (user != c->own_stream) {
            HIP_TRY(c, hipEventRecord(c->ev_graph_in, user));
            HIP_TRY(c, hipStreamWaitEvent(c->own_stream, c->ev_graph_in, 0));
        }
        if (!c->step_graph || !(c->graph_key == key)) {
            if (c->step_graph) {
                (void)hipGraphExecDestroy(c->step_graph);
                c->step_graph = nullptr;
            }
            hipGraph_t graph = nullptr;
            HIP_TRY(c, hipStreamBeginCapture(c->own_stream, hipStreamCaptureModeThreadLocal));
            c->stream = c->own_stream;
            rc = nbody_step_async(c, d_pos, d_vel, nullptr, dt, softening);
            c->stream = user;
            hipError_t e = hipStreamEndCapture(c->own_stream, &graph);
            if (rc != NBODY_OK || e != hipSuccess || !graph) {
                if (graph)
                    (void)hipGraphDestroy(graph);
                return rc != NBODY_OK ? rc : fail(c, NBODY_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            }
            e = hipGraphInstantiate(&c->step_graph, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) {
                c->step_graph = nullptr;
                return fail(c, NBODY_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
            }
            c->graph_key = key;
        }
        for (; s < k; ++s)
            HIP_TRY(c, hipGraphLaunch(c->step_graph, c->own_stream));
        if (user != c->own_stream) {
            HIP_TRY(c, hipEventRecord(c->ev_graph_out, c->own_stream));
            HIP_TRY(c, hipStreamWaitEvent(user, c->ev_graph_out, 0));
        }
        return nbody_sync(c);
    }
    for (; s < k; ++s) {
        int rc = nbody_step_async(c, d_pos, d_vel, nullptr, dt, softening);
        if (rc != NBODY_OK)
            return rc;
    }
    return nbody_sync(c);
}

// ---- body order on the device (nbody_order.hip) -------------------------------------------------------------------

static int ensure_order_buffers(nbody_ctx *c, size_t tmp_bytes)
{
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->order_perm && c->n_total)
        HIP_TRY(c, hipMalloc((void **)&c->order_perm, sizeof(unsigned) * (size_t)c->n_total));
    const size_t need = order_scratch_bytes((int)c->n_total);
    if (need > c->order_scratch_bytes) {
        if (c->order_scratch) {  // earlier order work may have been enqueued on either stream
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->own_stream));
            (void)hipFree(c->order_scratch);
        }
        c->order_scratch = nullptr;
        c->order_scratch_bytes = 0;
        HIP_TRY(c, hipMalloc(&c->order_scratch, need));
        c->order_scratch_bytes = need;
    }
    if (tmp_bytes > c->order_tmp_bytes) {
        if (c->order_tmp) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->own_stream));
            (void)hipFree(c->order_tmp);
        }
        c->order_tmp = nullptr;
        c->order_tmp_bytes = 0;
        HIP_TRY(c, hipMalloc(&c->order_tmp, tmp_bytes));
        c->order_tmp_bytes = tmp_bytes;
    }
    return NBODY_OK;
}

int nbody_order_compute(nbody_ctx *c, const float *d_xyzm, int64_t n, const int64_t *d_order)
{
    if (!c || n < 0 || n > c->n_total || (n > 0 && !d_xyzm))
        return fail(c, NBODY_ERR_INVALID, "nbody_order_compute: expected 0 <= n <= n_total bodies on the device");
    int rc = ensure_order_buffers(c, 0);
    if (rc != NBODY_OK)
        return rc;
    HIP_TRY(c, launch_morton_order(reinterpret_cast<const float4 *>(d_xyzm), (int)n, c->order_perm, c->order_scratch, c->stream,
                                   d_order));
    HIP_TRY(c, launch_identity_perm(c->order_perm, (int)n, (int)(c->n_total - n), c->stream));
    c->order_n = n;
    return NBODY_OK;
}

int nbody_order_gather(nbody_ctx *c, void *d_dst, const void *d_src, int64_t first, int64_t count, int floats_per_row)
{
    if (!c || first < 0 || count < 0 || first + count > c->n_total || (count > 0 && (!d_dst || !d_src)) ||
        !(floats_per_row == 1 || floats_per_row == 2 || floats_per_row == 4))
        return fail(c, NBODY_ERR_INVALID, "nbody_order_gather: rows [first, first + count) inside [0, n_total), 1, 2 or 4 floats a row");
    if (c->order_n < 0)
        return fail(c, NBODY_ERR_STATE, "nbody_order_gather: no permutation (nbody_order_compute first)");
    if (count == 0)
        return NBODY_OK;
    const bool in_place = d_dst == d_src;
    if (in_place && first != 0)
        return fail(c, NBODY_ERR_INVALID, "nbody_order_gather: in place only from row 0 (the source is indexed by body)");
    const size_t bytes = (size_t)count * (size_t)floats_per_row * sizeof(float);
    void *dst = d_dst;
    if (in_place) {
        int rc = ensure_order_buffers(c, bytes);
        if (rc != NBODY_OK)
            return rc;
        dst = c->order_tmp;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (floats_per_row == 4)
        HIP_TRY(c, launch_gather_float4(static_cast<float4 *>(dst), static_cast<const float4 *>(d_src), c->order_perm, (int)first,
                                        (int)count, c->stream));
    else if (floats_per_row == 2)
        HIP_TRY(c, launch_gather_int64(static_cast<int64_t *>(dst), static_cast<const int64_t *>(d_src), c->order_perm, (int)first,
                                       (int)count, c->stream));
    else
        HIP_TRY(c, launch_gather_float(static_cast<float *>(dst), static_cast<const float *>(d_src), c->order_perm, (int)first,
                                       (int)count, c->stream));
    if (in_place)
        HIP_TRY(c, hipMemcpyAsync(d_dst, dst, bytes, hipMemcpyDeviceToDevice, c->stream));
    return NBODY_OK;
}

int nbody_order_read(nbody_ctx *c, int64_t *d_perm)
{
    if (!c || (c->order_n > 0 && !d_perm))
        return fail(c, NBODY_ERR_INVALID, "nbody_order_read: NULL argument");
    if (c->order_n < 0)
        return fail(c, NBODY_ERR_STATE, "nbody_order_read: no permutation (nbody_order_compute first)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_widen_perm(d_perm, c->order_perm, (int)c->order_n, c->stream));
    return NBODY_OK;
}

int nbody_order_set(nbody_ctx *c, const int64_t *d_order, int64_t n, int inverse)
{
    if (!c || n < 0 || n > c->n_total || (n > 0 && !d_order))
        return fail(c, NBODY_ERR_INVALID, "nbody_order_set: expected 0 <= n <= n_total indices on the device");
    int rc = ensure_order_buffers(c, 0);
    if (rc != NBODY_OK)
        return rc;
    HIP_TRY(c, launch_set_perm(c->order_perm, d_order, (int)n, inverse != 0, c->stream));
    HIP_TRY(c, launch_identity_perm(c->order_perm, (int)n, (int)(c->n_total - n), c->stream));
    c->order_n = n;
    return NBODY_OK;
}

int nbody_order_identity(nbody_ctx *c, int64_t *d_order, int64_t n)
{
    if (!c || n < 0 || (n > 0 && !d_order))
        return fail(c, NBODY_ERR_INVALID, "nbody_order_identity: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_iota64(d_order, (int)n, c->stream));
    return NBODY_OK;
}

int nbody_order_permute_softening(nbody_ctx *c)
{
    if (!c)
        return NBODY_ERR_INVALID;
    if (!c->eps_own || c->eps_pp != c->eps_own)
        return NBODY_OK;  // no copy of its own in use: a borrowed array is the caller's to permute (nbody_order_gather)
    c->acc_valid = false;
    return nbody_order_gather(c, c->eps_own, c->eps_own, 0, c->n_total, 1);
}

int nbody_morton_order_device(nbody_ctx *c, const float *d_xyzm, int64_t n, int64_t *d_perm)
{
    int rc = nbody_order_compute(c, d_xyzm, n, nullptr);
    return rc == NBODY_OK ? nbody_order_read(c, d_perm) : rc;
}

int nbody_reorder(nbody_ctx *c, float *d_pos, float *d_vel, float *d_eps, int64_t *d_order, int64_t n)
{
    if (!c)
        return NBODY_ERR_INVALID;
    if (c->row_lo != 0 || c->row_count != c->n_total)
        return fail(c, NBODY_ERR_INVALID, "nbody_reorder: the context must own every row (shards: nbody_multi_reorder)");
    if (n < 0 || n > c->n_total || (n > 0 && (!d_pos || !d_vel)))
        return fail(c, NBODY_ERR_INVALID, "nbody_reorder: expected 0 <= n <= n_total and device positions and velocities");
    if (n == 0)
        return NBODY_OK;
    int rc = nbody_order_compute(c, d_pos, n, d_order);
    if (rc == NBODY_OK)
        rc = nbody_order_gather(c, d_pos, d_pos, 0, n, 4);
    if (rc == NBODY_OK)
        rc = nbody_order_gather(c, d_vel, d_vel, 0, n, 4);
    if (rc == NBODY_OK && d_eps)
        rc = nbody_order_gather(c, d_eps, d_eps, 0, n, 1);
    if (rc == NBODY_OK && d_eps != c->eps_own)
        rc = nbody_order_permute_softening(c);
    if (rc == NBODY_OK && d_order)
        rc = nbody_order_gather(c, d_order, d_order, 0, n, 2);
    if (rc == NBODY_OK)
        rc = nbody_invalidate_forces(c);
    return rc;
}

// ---- diagnostics ----------------------------------------------------------------------------

static int reduce_blocks(nbody_ctx *c, int blocks, int nvals, double *out)
{
    HIP_TRY(c, hipMemcpyAsync(c->reduce_host.data(), c->reduce_dev, sizeof(double) * (size_t)blocks * nvals,
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int v = 0; v < nvals; ++v)
        out[v] = 0.0;
    for (int b = 0; b < blocks; ++b)
        for (int v = 0; v < nvals; ++v)
            out[v] += c->reduce_host[(size_t)b * nvals + v];
    return NBODY_OK;
}

int nbody_energy(nbody_ctx *c, const float *d_pos, const float *d_vel, float softening, double *out3)
{
    if (!c || !out3 || ((!d_pos || !d_vel) && c->row_count))
        return fail(c, NBODY_ERR_INVALID, "nbody_energy: NULL argument");
    out3[0] = out3[1] = out3[2] = 0.0;
    if (c->row_count == 0)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_energy(reinterpret_cast<const float4 *>(d_pos), reinterpret_cast<const float4 *>(d_vel),
                             c->reduce_dev, (int)c->row_lo, (int)c->row_count, (int)c->n_total,
                             softening * softening, c->eps_pp, c->stream));
    double ku[2];
    int rc = reduce_blocks(c, energy_kernel_blocks((int)c->row_count), 2, ku);
    if (rc != NBODY_OK)
        return rc;
    out3[0] = ku[0];
    out3[1] = ku[1];
    out3[2] = ku[0] + ku[1];
    return NBODY_OK;
}

int nbody_momentum(nbody_ctx *c, const float *d_pos, const float *d_vel, double *out4)
{
    if (!c || !out4 || ((!d_pos || !d_vel) && c->row_count))
        return fail(c, NBODY_ERR_INVALID, "nbody_momentum: NULL argument");
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
    if (c->row_count == 0)
        return NBODY_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_momentum(reinterpret_cast<const float4 *>(d_pos), reinterpret_cast<const float4 *>(d_vel),
                               c->reduce_dev, (int)c->row_lo, (int)c->row_count, c->stream));
    return reduce_blocks(c, energy_blocks((int)c->row_count), 4, out4);
}

int nbody_device_info(nbody_ctx *c, int64_t *out4, char *name, int name_len)
{
    if (!c || !out4)
        return NBODY_ERR_INVALID;
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    out4[0] = prop.multiProcessorCount;
    out4[1] = prop.clockRate / 1000;  // kHz -> MHz
    out4[2] = prop.warpSize;
    out4[3] = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return NBODY_OK;
}

}  // extern "C"
