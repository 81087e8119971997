/*
 * vr_kernels.hip -- gfx950 kernels of the isosurface ray caster (renderImage / traceRay,
 * src/raytrace.zig:11-106, 117-255).  The per-ray arithmetic lives in vr_trace_core.h.
 *
 * The path is latency-bound pointer chasing over 4-byte nodes with no matrix work (no MFMA).
 * Two families of kernels:
 *
 *  per-pixel   one thread per pixel and iso value, 8x8-pixel tiles per wave64.  The literal
 *              root-restart loop (from global memory, or with the top tree levels in LDS) is the
 *              on-GPU anchor; a resume variant runs the same lane code as the pipeline below.
 *
 *  pipeline    (default) what makes the sweep fast:
 *    1. classify + compact: the entry test of traceRay (src/raytrace.zig:122-156) does not depend on
 *       the iso value, so it runs once per view; rays that enter the volume are written, in tile
 *       order, to a dense list (direction, entry t, pixel).
 *    2. frames are zero-filled (OOB and miss pixels are black, src/raytrace.zig:54,67).
 *    3. persistent trace kernel, one 1024-thread block per CU: the top 14 levels of the tree (which
 *       take two thirds of all node reads) sit in LDS, because what bounds this path on the chip is
 *       the latency of the vector L1 path under load (more resident waves make it thrash), not HBM
 *       and not the VALU.  Waves pull rays of one iso value at a time from that iso's queue and refill
 *       idle lanes as their rays finish.  Traversal resumes from the lowest valid ancestor after each
 *       MOVE (path stack in LDS) instead of restarting at the root.  Only hit pixels are stored.
 *    4. for a block of iso values, the sweep kernel: the same machinery, but a lane owns a ray and a
 *       block of iso values, and one traversal serves every iso value that cannot change the outcome
 *       of any density-range test on its way (same pixels and counters as one traversal per iso
 *       value, a quarter of the node reads on the reference's CT volume).
 */
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "vr_measure.h" /* empty macros unless -DVR_MEASURE */
#include "vr_internal.h"
#include "vr_trace_wide.h"
#include "vr_trace_union.h"
#include "vr_trace_union_wide.h"

namespace {

constexpr uint32_t kTileW = 8, kTileH = 8;     /* pixels per wave                         */
constexpr uint32_t kBlockW = 16, kBlockH = 16; /* pixels per 256-thread block (2x2 waves) */
constexpr uint32_t kTopNodes = 4095;           /* heap levels 0..11, 16 KiB of LDS        */
#ifndef VR_REFILL_MIN
#define VR_REFILL_MIN 64
#endif
/* a wave takes new rays once this many of its lanes are idle.  64 = the whole wave: it then always holds ONE tile of 64 neighbouring
 * rays, whose walks read the same nodes and end at about the same time (Skull 4096^2 iso 80: 0.945 ms with 16, 0.834 with 64; C60 1024^2
 * iso 100: 0.147 / 0.140; Skull 1024^2 iso 80, 1.16 rays per lane: the same) */
constexpr uint32_t kRefillMin = VR_REFILL_MIN;
#ifndef VR_DESCEND_TRIPS
#define VR_DESCEND_TRIPS 6
#endif
#ifndef VR_POP_TRIPS
#define VR_POP_TRIPS 8 /* (frames of Skull 1024^2 / C60 1024^2 / Skull 4096^2, kernel ms: 6: 0.303 / 0.147 / 0.953, 8: 0.296 / 0.147 / 0.943, 10: 0.302 / 0.155 / 0.966) */
#endif
/* levels a lane may descend / pop per round before the wave moves on; unfinished lanes continue next round */
constexpr uint32_t kDescendTrips = VR_DESCEND_TRIPS, kPopTrips = VR_POP_TRIPS;

struct GlobalFetch {
    const uint32_t *nodes;
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const
    {
        return nodes[idx];
    }
};

struct TopLdsFetch {
    const uint32_t *nodes;
    const uint32_t *top;
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const
    {
        return idx < kTopNodes ? top[idx] : nodes[idx];
    }
};

/* Path stack cursor.  The stacks sit at the start of the block's dynamic LDS as [row][thread] words: row 0
 * is a spare (looking one level above the root stays inside the allocation), row d + 1 holds the word pushed
 * at depth d.  The cursor is the LDS byte address of row `depth` in this thread's column. */
template <uint32_t STRIDE>
struct LdsStack {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    static constexpr uint32_t kRowBytes = STRIDE * 4u;
    uint32_t origin; /* LDS byte address of row 0 of this thread's column */
    uint32_t cur;    /* LDS byte address of row `depth` */
    __device__ __forceinline__ LdsStack()
    {
        extern __shared__ uint32_t s_dyn[];
        origin = (uint32_t)(uintptr_t)s_dyn + threadIdx.x * 4u; /* low word of a flat LDS address = LDS address */
        cur = origin;
    }
    __device__ __forceinline__ void reset() { cur = origin; }
    __device__ __forceinline__ void push(uint32_t word) { *(lds_u32 *)(uintptr_t)(cur + kRowBytes) = word; cur += kRowBytes; }
    __device__ __forceinline__ void up() { cur -= kRowBytes; }
    __device__ __forceinline__ uint32_t peek_up(uint32_t k) const { return *(lds_u32 *)(uintptr_t)(cur + kRowBytes - k * kRowBytes); }
    __device__ __forceinline__ uint32_t depth() const { return (cur - origin) / kRowBytes; }
};

/* Node fetch of the persistent kernel: heap positions below n_top come from the LDS copy of the top tree
 * levels, deeper ones from global memory.
 * One flat load whose 64-bit address is put together by hand: the low word is 4 * g plus the low word of
 * either base, the high word is either the LDS aperture or the high word of the node array's address.
 * The node array must not cross a 4 GiB boundary (the allocation in vr_capi.hip sees to that), so the
 * low-word add cannot carry. */
struct TopFetch {
    uint32_t g_lo, g_hi, l_lo, l_hi, n_top;
    uint32_t last_pair; /* the last even position whose right neighbour is still inside the array */
    __device__ __forceinline__ TopFetch(const uint32_t *nodes, const uint32_t *lds_top, uint32_t n_top_, uint64_t n_slots = 1)
    {
        last_pair = (uint32_t)(n_slots - 1u) & ~1u; /* positions 0 .. n_slots exist */
        const uint64_t ga = (uint64_t)(uintptr_t)nodes, la = (uint64_t)(uintptr_t)(const void *)lds_top;
        g_lo = (uint32_t)ga; g_hi = (uint32_t)(ga >> 32);
        l_lo = (uint32_t)la; l_hi = (uint32_t)(la >> 32);
        n_top = n_top_;
        /* held in vector registers for the whole kernel: a select needs both of its inputs there, and the
         * compiler would otherwise copy them out of scalar registers at every use */
        asm volatile("" : "+v"(g_lo), "+v"(g_hi), "+v"(l_lo), "+v"(l_hi));
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t g) const
    {
        const bool top = g < n_top;
        const uint32_t lo = (top ? l_lo : g_lo) + (g << 2);
        const uint32_t hi = top ? l_hi : g_hi;
        return *(const uint32_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    /* the words at 2g and 2g + 1 in one 8-byte read (both arrays are 8-byte aligned, n_top is a power of two, so a
     * pair lies in one of them); for a node of the last level -- a leaf, whose children nobody uses -- the last pair
     * of the array instead */
    __device__ __forceinline__ VrNodePair children(uint32_t g) const
    {
        uint32_t c = g << 1;
        c = c < last_pair ? c : last_pair;
        const bool top = c < n_top;
        const uint32_t lo = (top ? l_lo : g_lo) + (c << 2);
        const uint32_t hi = top ? l_hi : g_hi;
        const uint2 w = *(const uint2 *)(uintptr_t)(((uint64_t)hi << 32) | lo);
        return VrNodePair{ w.x, w.y };
    }
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t y = __shfl_xor(x, off, 64); x = y > x ? y : x; }
    return x;
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ void add_stat(uint64_t *p, unsigned long long v)
{
    if (v) atomicAdd((unsigned long long *)p, v);
}

/* ------------------------------------------------------------------ per-pixel kernels ---- */

template <int VARIANT>
__global__ __launch_bounds__(256) void vr_k_render(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t max_depth,
                                                   uint32_t iso_begin, uint32_t row_begin, uint32_t n_rows,
                                                   uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats,
                                                   VrTapsK taps)
{
    __shared__ uint32_t s_top[VARIANT == VR_KERNEL_RESTART_LDS ? kTopNodes : 1];
    extern __shared__ uint32_t s_dyn[]; /* resume variant: path stack, [levels][256] */

    const uint32_t tiles_x = (v.width + kBlockW - 1) / kBlockW;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t xi = bx * kBlockW + (wave & 1u) * kTileW + (lane & 7u);
    const uint32_t yr = by * kBlockH + (wave >> 1) * kTileH + (lane >> 3); /* row relative to row_begin */
    const uint32_t iso = iso_begin + blockIdx.y;
    const bool valid = xi < v.width && yr < n_rows;

    if (VARIANT == VR_KERNEL_RESTART_LDS) {
        const uint32_t n_top = v.n_slots < kTopNodes ? (uint32_t)v.n_slots : kTopNodes;
        for (uint32_t i = threadIdx.x; i < n_top; i += 256u) s_top[i] = nodes[i];
        __syncthreads();
    }

    VrHit h;
    h.t = 0.0f; h.flag = VR_FLAG_OOB; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (valid) {
        VrRay ray = vr_make_ray(v, xi, row_begin + vr_band_row(v, yr));
        if (VARIANT == VR_KERNEL_RESTART_GLOBAL) {
            h = vr_trace_restart(v, ray, iso, GlobalFetch{ nodes });
        } else if (VARIANT == VR_KERNEL_RESTART_LDS) {
            h = vr_trace_restart(v, ray, iso, TopLdsFetch{ nodes, s_top });
        } else {
            LdsStack<256u> st;
            h = vr_trace_resume(v, ray, iso, GlobalFetch{ nodes }, st, 64u);
        }
        const uint64_t pix = ((uint64_t)blockIdx.y * n_rows + yr) * v.width + xi;
        const uint8_t g = h.flag == VR_FLAG_HIT ? vr_shade(h.t) : (uint8_t)0;
        rgb[pix * 3 + 0] = g;
        rgb[pix * 3 + 1] = g;
        rgb[pix * 3 + 2] = g;
        if (taps.cell_flag) taps.cell_flag[pix] = h.cell | (h.flag << 24);
        if (taps.t) taps.t[pix] = h.flag == VR_FLAG_OOB ? 0.0f : h.t;
        if (taps.checked) taps.checked[pix] = h.checked;
        if (taps.node) taps.node[pix] = h.node;
    }

    if (stats) {
        const uint32_t c_checked = wave_sum(valid ? h.checked : 0u);
        const uint32_t c_hit = wave_sum(valid && h.flag == VR_FLAG_HIT ? 1u : 0u);
        const uint32_t c_miss = wave_sum(valid && h.flag == VR_FLAG_MISS ? 1u : 0u);
        const uint32_t c_oob = wave_sum(valid && h.flag == VR_FLAG_OOB ? 1u : 0u);
        const uint32_t c_esc = wave_sum(valid ? h.escapes : 0u);
        const uint32_t c_moves = wave_sum(valid ? h.moves : 0u);
        const uint32_t c_err = wave_sum(valid ? h.error : 0u);
        if (lane == 0) {
            vr_render_stats *s = stats + blockIdx.y;
            add_stat(&s->nodes_checked, c_checked);
            add_stat(&s->hit_count, c_hit);
            add_stat(&s->miss_count, c_miss);
            add_stat(&s->oob_count, c_oob);
            add_stat(&s->escape_help_events, c_esc);
            add_stat(&s->moves, c_moves);
            add_stat(&s->errors, c_err);
        }
    }
}

/* ---------------------------------------------------------------- pipeline: pre-pass ---- */

#ifndef VR_SUPER_TILE
#define VR_SUPER_TILE 4
#endif
/* Tiles are numbered super-tile by super-tile (kSuper x kSuper tiles, row-major inside and across), so that
 * rays next to each other in the compacted list -- which waves of one CU work on at the same time -- come from
 * a compact square of the image and walk the same part of the tree.  The tile grid is padded to whole
 * super-tiles; tiles outside the image hold no rays. */
constexpr uint32_t kSuper = VR_SUPER_TILE;
__host__ __device__ __forceinline__ uint32_t super_tiles_x(uint32_t width) { return (width + kTileW * kSuper - 1) / (kTileW * kSuper); }
__host__ __device__ __forceinline__ uint32_t super_tiles_y(uint32_t n_rows) { return (n_rows + kTileH * kSuper - 1) / (kTileH * kSuper); }
__host__ __device__ __forceinline__ uint32_t padded_tiles(uint32_t width, uint32_t n_rows)
{
    return super_tiles_x(width) * super_tiles_y(n_rows) * kSuper * kSuper;
}

/* column and row (inside the call's rows) of lane `lane` of tile `tile`; they may lie outside the image */
__device__ __forceinline__ void tile_xy(const VrViewK &v, uint32_t tile, uint32_t lane, uint32_t n_rows, bool center_out, uint32_t &xi, uint32_t &yr)
{
    const uint32_t st = tile / (kSuper * kSuper), in = tile % (kSuper * kSuper), sx = super_tiles_x(v.width);
    const uint32_t row_i = st / sx, mid = super_tiles_y(n_rows) / 2u;
    const uint32_t row = !center_out ? row_i : ((row_i & 1u) ? mid - (row_i + 1u) / 2u : mid + row_i / 2u);
    xi = ((st % sx) * kSuper + in % kSuper) * kTileW + (lane & 7u);
    yr = (row * kSuper + in / kSuper) * kTileH + (lane >> 3);
}

__device__ __forceinline__ bool tile_ray(const VrViewK &v, uint32_t tile, uint32_t lane, uint32_t row_begin, uint32_t n_rows, bool center_out,
                                         VrRay &ray, uint32_t &pix)
{
    const uint32_t st = tile / (kSuper * kSuper), in = tile % (kSuper * kSuper), sx = super_tiles_x(v.width);
    /* center_out (the sweep kernel's order): rows of super-tiles from the middle of the image outwards (middle,
     * one above, one below, ...), so that what is still in its queue at the end of a launch is the border of the
     * image, where rays are short.  The kernel with one queue per iso value wants neighbouring rows next to
     * each other in the list instead (its waves share more lines in the L1 that way: 14.3 against 17.8 ms). */
    const uint32_t row_i = st / sx, mid = super_tiles_y(n_rows) / 2u;
    const uint32_t row = !center_out ? row_i : ((row_i & 1u) ? mid - (row_i + 1u) / 2u : mid + row_i / 2u);
    const uint32_t xi = ((st % sx) * kSuper + in % kSuper) * kTileW + (lane & 7u);
    const uint32_t yr = (row * kSuper + in / kSuper) * kTileH + (lane >> 3);
    pix = yr * v.width + xi;
    if (xi >= v.width || yr >= n_rows) return false;
    ray = vr_make_ray(v, xi, row_begin + vr_band_row(v, yr));
    return !ray.oob;
}

/* The pre-pass works in blocks of one super-tile (kSuper x kSuper tiles, one wave per tile): what a block adds
 * to the ray list is contiguous there, and only one number per block goes through the scan. */
constexpr uint32_t kPreThreads = kSuper * kSuper * 64u;
static_assert(kPreThreads <= 1024u, "a super-tile must fit one block");

/* entry test of the block's tiles; returns the number of entering rays in the tiles before this wave's tile
 * (exclusive prefix inside the block) and the block total */
__device__ __forceinline__ bool pre_block(const VrViewK &v, uint32_t row_begin, uint32_t n_rows, bool center_out, VrRay &ray,
                                          uint32_t &pix, unsigned long long &mask, uint32_t &before, uint32_t &total)
{
    __shared__ uint32_t s_cnt[kSuper * kSuper];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * (kSuper * kSuper) + wave;
    const bool active = tile_ray(v, tile, lane, row_begin, n_rows, center_out, ray, pix);
    mask = __ballot(active);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    before = 0; total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kSuper * kSuper; w++) {
        const uint32_t c = s_cnt[w];
        before += w < wave ? c : 0u;
        total += c;
    }
    return active;
}

/* frames are cleared by a memset from this many iso values per call on, or when they hold more than this many pixels together
 * (byte stores of a thread per pixel: 4096^2 takes 0.02 ms longer that way than the memset it saves) */
constexpr uint32_t kCountClearsBelow = 8;
constexpr size_t kCountClearsPixels = (size_t)4 << 20;

/* 1a: number of rays per super-tile that enter the root box.  A call of a few iso values has its frames cleared here as well
 * (clear_isos of them, a thread its own pixel: OOB and miss pixels are black, src/raytrace.zig:54,67) instead of by a memset of
 * its own: one stream operation less in front of a kernel that takes a fraction of a millisecond. */
__global__ __launch_bounds__(kPreThreads) void vr_k_count(VrViewK v, uint32_t row_begin, uint32_t n_rows, uint32_t center_out,
                                                          uint32_t *__restrict__ block_count, uint8_t *__restrict__ rgb, uint32_t clear_isos)
{
    VrRay ray;
    uint32_t pix, before, total;
    unsigned long long mask;
    (void)pre_block(v, row_begin, n_rows, center_out != 0u, ray, pix, mask, before, total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
    if (clear_isos) {
        uint32_t xi, yr;
        tile_xy(v, blockIdx.x * (kSuper * kSuper) + (threadIdx.x >> 6), threadIdx.x & 63u, n_rows, center_out != 0u, xi, yr);
        if (xi < v.width && yr < n_rows) {
            const uint64_t frame_bytes = (uint64_t)v.width * n_rows * 3u;
            uint8_t *p = rgb + ((uint64_t)yr * v.width + xi) * 3u;
            for (uint32_t k = 0; k < clear_isos; k++, p += frame_bytes) { p[0] = 0u; p[1] = 0u; p[2] = 0u; }
        }
    }
}

/* Order of the tiles (groups of 64 consecutive rays of the list) for a union launch that repeats the previous call:
 * longest first.  tile_cost holds, per tile, the largest number of rounds one lane spent on one of the tile's items in
 * the previous launch.  A launch ends with its longest fronts, so they start first, and the queue ends with the
 * shortest (a round takes about 10 us: twenty rounds more at the end of the queue are a fifth of a Skull sweep).
 * Counting sort on the length (in steps of one round up to 255 rounds, of two up to 511, ...), longest first; inside a
 * length the tiles come as the waves get to them.  Work of one block of 1024 threads (the second block of vr_k_scan);
 * leaves tile_cost at zero for the launch that follows. */
constexpr uint32_t kLptClasses = 256;
constexpr uint32_t kLptHeaderWords = 64; /* in front of the lengths; word 0: tiles in the list of the launch that left them,
                                          * word 1: whether the pre-pass of this launch has made an order from them */
constexpr uint32_t kLptMaxRaysPerLane = 8; /* lists beyond this many rays per lane of the grid keep their order */
/* A tile's length has a 256-byte line of its own: the items that end at the same time are those of neighbouring tiles of the
 * list, and their atomics on neighbouring words of a few lines held up every wave behind them (a first call of the Skull
 * sweep cut into 4 blocks per ray: 1.78 ms with packed words against 1.55 without keeping lengths; 8 blocks: 2.96 against 1.86) */
constexpr uint32_t kLptStride = VR_LPT_STRIDE_WORDS;
/* the longest list (in tiles) that is ordered: 8 rays per lane of the grid, and what the array behind the workspace holds */
__host__ __device__ inline uint32_t lpt_max_tiles(uint32_t grid_lanes)
{
    const uint32_t by_grid = grid_lanes / 64u * kLptMaxRaysPerLane;
    return by_grid < VR_LPT_MAX_TILES ? by_grid : VR_LPT_MAX_TILES;
}
constexpr uint32_t kLptPerLane = 16; /* lengths a lane reads at a time: the reads are in flight together */
__device__ void tile_order_block(uint32_t n_tiles, uint32_t *__restrict__ tile_cost, uint32_t *__restrict__ tile_order)
{
    __shared__ uint32_t s_max[16];
    __shared__ uint32_t s_at[kLptClasses]; /* per class: tiles, then where the next tile of the class goes */
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per_wave = ((n_tiles + 15u) / 16u + 63u) & ~63u;
    const uint32_t w0 = wave * per_wave < n_tiles ? wave * per_wave : n_tiles;
    const uint32_t w1 = w0 + per_wave < n_tiles ? w0 + per_wave : n_tiles;
    const bool one_chunk = w1 - w0 <= 64u * kLptPerLane; /* then the lengths stay in registers between the passes */
    uint32_t len[kLptPerLane];
    uint32_t mx = 0;
    if (threadIdx.x < kLptClasses) s_at[threadIdx.x] = 0u;
    for (uint32_t g0 = w0; g0 < w1; g0 += 64u * kLptPerLane) {
#pragma unroll
        for (uint32_t j = 0; j < kLptPerLane; j++) { const uint32_t g = g0 + j * 64u + lane; len[j] = g < w1 ? tile_cost[(size_t)g * kLptStride] : 0u; }
#pragma unroll
        for (uint32_t j = 0; j < kLptPerLane; j++) mx = len[j] > mx ? len[j] : mx;
    }
    mx = wave_max(mx);
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16u; w++) mx = s_max[w] > mx ? s_max[w] : mx;
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(mx | 1u), shift = bits > 8u ? bits - 8u : 0u; /* longest >> shift <= 255 */
    for (uint32_t g0 = w0; g0 < w1; g0 += 64u * kLptPerLane) {
        if (!one_chunk) {
#pragma unroll
            for (uint32_t j = 0; j < kLptPerLane; j++) { const uint32_t g = g0 + j * 64u + lane; len[j] = g < w1 ? tile_cost[(size_t)g * kLptStride] : 0u; }
        }
#pragma unroll
        for (uint32_t j = 0; j < kLptPerLane; j++)
            if (g0 + j * 64u + lane < w1) atomicAdd(&s_at[(kLptClasses - 1u) - (len[j] >> shift)], 1u);
    }
    __syncthreads();
    /* counts to starting positions, longest class first (scan in waves 0-3, then their totals) */
    uint32_t mine = 0, incl = 0;
    if (threadIdx.x < kLptClasses) {
        mine = s_at[threadIdx.x];
        incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off, 64);
            incl += lane >= (uint32_t)off ? y : 0u;
        }
        if (lane == 63u) s_max[wave] = incl; /* every thread has read the lengths' maxima by now */
    }
    __syncthreads();
    if (threadIdx.x < kLptClasses) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += s_max[w];
        s_at[threadIdx.x] = before + incl - mine;
    }
    __syncthreads();
    for (uint32_t g0 = w0; g0 < w1; g0 += 64u * kLptPerLane) {
        if (!one_chunk) {
#pragma unroll
            for (uint32_t j = 0; j < kLptPerLane; j++) { const uint32_t g = g0 + j * 64u + lane; len[j] = g < w1 ? tile_cost[(size_t)g * kLptStride] : 0u; }
        }
#pragma unroll
        for (uint32_t j = 0; j < kLptPerLane; j++) {
            const uint32_t g = g0 + j * 64u + lane;
            if (g < w1) {
                tile_order[atomicAdd(&s_at[(kLptClasses - 1u) - (len[j] >> shift)], 1u)] = g;
                tile_cost[(size_t)g * kLptStride] = 0u;
            }
        }
    }
}

/* The order from what the words behind the header hold for the tiles of the list (word 0 of the header: how many).  A list of
 * many rays per lane of the grid keeps its order: such a launch is not bound by how it ends (32 iso values of the 1024^3 gyroid
 * at 4096^2, 48 rays per lane: 12.5 ms in list order, 12.6 longest first, plus 0.1 ms for the order).  Word 1 of the header says
 * what the order was made from: 0 = there is none, 1 = the item lengths of an earlier launch, 2 = the rays' chords (a first call). */
__device__ void tile_order_step(uint32_t n_tiles, uint32_t *__restrict__ tile_cost, uint32_t *__restrict__ tile_order, uint32_t order_max_tiles,
                                uint32_t made_from)
{
    const uint32_t listed = *(tile_cost - kLptHeaderWords);
    const bool wanted = listed <= order_max_tiles;
    if (wanted) tile_order_block(listed < n_tiles ? listed : n_tiles, tile_cost, tile_order);
    if (threadIdx.x == 0) *(tile_cost - kLptHeaderWords + 1) = wanted ? made_from : 0u;
}

/* the same as a launch of its own: behind vr_k_compact, for the order of a first call (from the chords that kernel leaves) */
__global__ __launch_bounds__(1024) void vr_k_tile_order(uint32_t n_tiles, uint32_t *__restrict__ tile_cost, uint32_t *__restrict__ tile_order,
                                                        uint32_t order_max_tiles)
{
    tile_order_step(n_tiles, tile_cost, tile_order, order_max_tiles, 2u);
}

/* 1b: exclusive scan of the per-block counts in place, by one block; writes the total.  Every wave takes a
 * contiguous share and goes through it 64 values at a time (coalesced), twice: totals first, then the scan. */
__global__ __launch_bounds__(1024) void vr_k_scan(uint32_t *__restrict__ count, uint32_t n, uint32_t *__restrict__ n_active_out,
                                                  uint32_t *__restrict__ heads, uint32_t iso_count,
                                                  vr_render_stats *__restrict__ stats, uint64_t frame_pixels, uint32_t n_tiles,
                                                  uint32_t *__restrict__ tile_cost, uint32_t *__restrict__ tile_order,
                                                  uint32_t order_max_tiles, uint32_t *__restrict__ chord_header, uint32_t list_kept)
{
    if (blockIdx.x == 1u) { /* launched only when the union kernel's tile order is wanted: it rides along on a block of its own */
        tile_order_step(n_tiles, tile_cost, tile_order, order_max_tiles, 1u); /* the same list as this call's: same call */
        return;
    }
    __shared__ uint32_t s_wave[16];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t per = ((n + 15u) / 16u + 63u) & ~63u; /* values per wave, a multiple of 64 */
    /* list_kept: the ray list of the last call on this workspace is this call's (same view, same rows): the offsets and the
     * total are in place, only the set-up work below is done */
    const uint32_t b = wave * per, e = list_kept ? b : (b + per < n ? b + per : n);
    uint32_t sum = 0;
    for (uint32_t i = b + lane; i < e; i += 64u) sum += count[i];
    sum = wave_sum(sum);
    if (lane == 0) s_wave[wave] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16u; w++) {
        const uint32_t c = s_wave[w];
        run += w < wave ? c : 0u;
        total += c;
    }
    if (list_kept) total = *n_active_out;
    for (uint32_t i0 = b; i0 < e; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t c = i < e ? count[i] : 0u;
        uint32_t x = c; /* inclusive scan over the wave */
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            x += lane >= (uint32_t)off ? y : 0u;
        }
        if (i < e) count[i] = run + x - c;
        run += __shfl(x, 63, 64);
    }
    if (threadIdx.x == 0) {
        *n_active_out = total;
        /* the sweep kernel's report (vr_sweep_info, 64 bytes behind the ray count): walks, items, first / queue-empty / last tick */
        unsigned long long *info = (unsigned long long *)(n_active_out + 16);
        info[0] = 0; info[1] = 0; info[2] = ~0ull; info[3] = ~0ull; info[4] = 0; info[5] = 0;
        n_active_out[32] = 0u; /* the union kernel's count of iso runs that left the union walk (128 bytes behind the ray count) */
        n_active_out[33] = 0u; /* ... and its request to render the call again with the sweep kernel (kUnionRedoWord) */
        if (chord_header) chord_header[0] = (total + 63u) >> 6; /* a first call orders its tiles by their rays' chords: the list's tiles */
    }
    /* the small set-up work of a call rides along: empty queues, and the per-iso counters at their starting
     * values (every pixel that never enters the volume is OOB in every frame) */
    for (uint32_t k = threadIdx.x; k < iso_count; k += 1024u) {
        heads[k] = 0u;
        if (stats) {
            vr_render_stats z = { 0, 0, 0, frame_pixels - total, 0, 0, 0 };
            stats[k] = z;
        }
    }
}

/* 1c: dense ray list in tile order: (dx, dy, dz, entry t) and the pixel index inside one frame */
__global__ __launch_bounds__(kPreThreads) void vr_k_compact(VrViewK v, uint32_t row_begin, uint32_t n_rows, uint32_t center_out,
                                                            const uint32_t *__restrict__ block_off, float4 *__restrict__ ray_dt,
                                                            uint32_t *__restrict__ ray_pix, uint32_t *__restrict__ tile_chord,
                                                            uint32_t order_max_tiles, const uint32_t *__restrict__ n_active_ptr)
{
    /* a super-tile none of whose rays enters the volume (71 % of the Skull image) has nothing to write: the scan left its
     * offset equal to its neighbour's */
    if (block_off[blockIdx.x] == (blockIdx.x + 1u < gridDim.x ? block_off[blockIdx.x + 1u] : *n_active_ptr)) return;
    VrRay ray;
    uint32_t pix, before, total;
    unsigned long long mask;
    const bool active = pre_block(v, row_begin, n_rows, center_out != 0u, ray, pix, mask, before, total);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t slot = 0;
    if (active) {
        slot = block_off[blockIdx.x] + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        ray_dt[slot] = make_float4(ray.dx, ray.dy, ray.dz, ray.t);
        ray_pix[slot] = pix;
    }
    /* A first call of a sweep has no item lengths to order its tiles by; the longest chord of a tile's rays through the
     * volume stands in (rank correlation with the lengths: 0.84 on Skull, 0.89 on the 256^3 stand-in, 0.48 on C60; the list's
     * own order: 0.36 / 0.58 / 0.15).  In quarter units, + 1; the rays of a wave lie in one or two tiles of the list. */
    if (tile_chord && *(tile_chord - kLptHeaderWords) <= order_max_tiles && mask != 0ull) {
        uint32_t q = 0;
        if (active) {
            const float ex = ray.dx > 0.0f ? (float)v.rdx : 0.0f, ey = ray.dy > 0.0f ? (float)v.rdy : 0.0f, ez = ray.dz > 0.0f ? (float)v.rdz : 0.0f;
            float te = 3.0e38f;
            if (ray.dx != 0.0f) te = fminf(te, (ex - v.ox) / ray.dx);
            if (ray.dy != 0.0f) te = fminf(te, (ey - v.oy) / ray.dy);
            if (ray.dz != 0.0f) te = fminf(te, (ez - v.oz) / ray.dz);
            const float chord = fminf(fmaxf(te - ray.t, 0.0f), 4096.0f);
            q = (uint32_t)(chord * 4.0f) + 1u;
        }
        const uint32_t first = ~wave_max(active ? ~(slot >> 6) : 0u);
        for (uint32_t g = first; g <= first + 1u; g++) {
            const uint32_t m = wave_max(active && (slot >> 6) == g ? q : 0u);
            if (m && lane == 0u) atomicMax(&tile_chord[(size_t)g * kLptStride], m);
        }
    }
}

/* taps of the pixels no ray list entry writes */
__global__ void vr_k_tap_defaults(VrTapsK taps, uint64_t n, uint32_t oob_word)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (taps.cell_flag) taps.cell_flag[i] = oob_word;
        if (taps.t) taps.t[i] = 0.0f;
        if (taps.checked) taps.checked[i] = 0u;
        if (taps.node) taps.node[i] = 0u;
    }
}

/* --------------------------------------------------------- pipeline: persistent trace ---- */


#ifndef VR_TRACE_THREADS
#define VR_TRACE_THREADS 1024
#endif
constexpr uint32_t kTraceThreads = VR_TRACE_THREADS;
#ifndef VR_TRACE_BLOCKS_PER_CU
#define VR_TRACE_BLOCKS_PER_CU 1
#endif
/* resident blocks per CU.  One block of 16 waves (4 per SIMD) keeps the vector ALUs about 70 % busy; with
 * more waves the per-CU vector L1 starts to thrash (hit rate 95 % -> 85 % at 8 per SIMD) and the sweep
 * gets slower, not faster. */
constexpr uint32_t kTraceBlocksPerCu = VR_TRACE_BLOCKS_PER_CU;
constexpr uint32_t kTraceWavesPerSimd = kTraceThreads / 64u * kTraceBlocksPerCu / 4u;

/* Per-iso counters of the kernels with one traversal per iso value.  With a handful of iso values every wave of
 * the grid would add to the same few words, and device-wide atomics on one address are served in series (about
 * 10 ns each: the backlog of 4096 waves x 6 counters outlasts a short kernel).  There the waves of a block add their
 * sums up in LDS (`s_acc`: [iso value][counter]) and the block adds them to the counters once, at its end: 16 times
 * fewer device-wide atomics, spread over the time in which the blocks end; sweeps of many iso values spread over as
 * many records and use device-wide atomics straight away. */
constexpr uint32_t kSlotIsoMax = 8;

__device__ __forceinline__ void wave_stats_out(vr_render_stats *__restrict__ stats, unsigned long long *s_acc, uint32_t k, uint32_t lane,
                                               unsigned long long a_checked, uint32_t a_hit, uint32_t a_miss, uint32_t a_esc, uint32_t a_moves,
                                               uint32_t a_err)
{
    const unsigned long long c_checked = wave_sum64(a_checked);
    const uint32_t c_hit = wave_sum(a_hit), c_miss = wave_sum(a_miss), c_moves = wave_sum(a_moves);
    const uint32_t c_esc = wave_sum(a_esc), c_err = wave_sum(a_err);
    if (lane != 0) return;
    if (s_acc) {
        unsigned long long *a = s_acc + k * 6u;
        atomicAdd(&a[0], c_checked); atomicAdd(&a[1], (unsigned long long)c_hit); atomicAdd(&a[2], (unsigned long long)c_miss);
        atomicAdd(&a[3], (unsigned long long)c_esc); atomicAdd(&a[4], (unsigned long long)c_moves); atomicAdd(&a[5], (unsigned long long)c_err);
        return;
    }
    vr_render_stats *s = stats + k;
    add_stat(&s->nodes_checked, c_checked);
    add_stat(&s->hit_count, c_hit);
    add_stat(&s->miss_count, c_miss);
    add_stat(&s->escape_help_events, c_esc);
    add_stat(&s->moves, c_moves);
    add_stat(&s->errors, c_err);
}

/* the block's sums of a short iso range go to the counters (oob_count was set by the scan kernel); every thread of the block calls it */
__device__ __forceinline__ void block_stats_out(vr_render_stats *__restrict__ stats, const unsigned long long *s_acc, uint32_t iso_count)
{
    __syncthreads();
    if (threadIdx.x < iso_count * 6u) {
        const uint32_t k = threadIdx.x / 6u, i = threadIdx.x - k * 6u;
        vr_render_stats *s = stats + k;
        uint64_t *dst = i == 0 ? &s->nodes_checked : i == 1 ? &s->hit_count : i == 2 ? &s->miss_count
                      : i == 3 ? &s->escape_help_events : i == 4 ? &s->moves : &s->errors;
        if (s_acc[threadIdx.x]) add_stat(dst, s_acc[threadIdx.x]);
    }
}

/* A wave's private share of a work queue.  Waves take work from the shared queue head in chunks and hand it
 * to their idle lanes from there, so the head sees one atomic per chunk instead of one per refill (a device-wide
 * atomic on one address is served in series: with a single iso value every wave of the grid pulls from the same
 * head: 10 ns per atomic, measured).  Chunks stay small -- neighbouring chunks go to waves that run at the same
 * time, which is what keeps the node lines they share hot in the L1 / L2; large private chunks send every wave
 * to a region of its own and lose more than the atomics cost.  The sweep kernel's items are long and few, its
 * head is far from saturated, and items kept back by one wave are missed by the others at the end of the queue:
 * there a wave takes exactly what its idle lanes need (measured: 4.3 ms against 4.7 - 6.7 ms with chunks of 16 - 128). */
#ifndef VR_TRACE_CHUNK_MAX
#define VR_TRACE_CHUNK_MAX 128
#endif
#ifndef VR_SWEEP_CHUNK
#define VR_SWEEP_CHUNK 0 /* items per chunk of the sweep kernel; 0 = exactly what the idle lanes need, nothing kept back */
#endif
/* rays per chunk of the kernels with one traversal per iso value: one wave's worth, more (up to
 * VR_TRACE_CHUNK_MAX) when there are many rays per wave of the grid, as in a 4096^2 frame */
__device__ __forceinline__ uint32_t trace_chunk(uint32_t n_active, uint32_t n_waves)
{
    uint32_t c = 64u;
    while (c < VR_TRACE_CHUNK_MAX && (uint64_t)c * 16u * n_waves <= n_active) c <<= 1;
    return c;
}
struct WaveChunk {
    uint32_t next, end; /* uniform over the wave */
    uint32_t first;     /* items in front of the queue's own: handed out without it (see seed) */
    bool dry;           /* the shared queue has nothing left */
    __device__ __forceinline__ WaveChunk() : next(0), end(0), first(0), dry(false) {}
    /* The first `per_wave` items of every wave of the grid come without the queue: wave w of n_waves starts with items
     * [w * per_wave, + per_wave), the queue hands out what lies behind n_waves * per_wave.  (A launch begins with every wave
     * asking at once: 4 096 atomics on one word, served one after the other -- the last wave of a Skull sweep launch had its
     * first items 48 us after the first wave's start.) */
    __device__ __forceinline__ void seed(uint32_t w, uint32_t n_waves, uint32_t per_wave, uint32_t n_total)
    {
        const uint64_t all = (uint64_t)n_waves * per_wave, at = (uint64_t)w * per_wave;
        first = all < n_total ? (uint32_t)all : n_total;
        next = at < n_total ? (uint32_t)at : n_total;
        end = at + per_wave < n_total ? (uint32_t)(at + per_wave) : n_total;
    }
    __device__ __forceinline__ void grab(uint32_t *head, uint32_t n_total, uint32_t want, uint32_t lane)
    {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, want);
        base = __builtin_amdgcn_readfirstlane(base) + first;
        dry = base >= n_total;
        next = dry ? 0u : base;
        end = dry ? 0u : (base + want < n_total ? base + want : n_total);
    }
    /* Work for the n_idle idle lanes of the wave (`rank` = this lane's position among them): true and the item
     * in `mine` for the lanes that get one.  Takes what the wave's chunk still holds and, when that is not
     * enough, the rest from one fresh chunk. */
    __device__ __forceinline__ bool take(uint32_t n_idle, uint32_t rank, uint32_t *head, uint32_t n_total, uint32_t want,
                                         uint32_t lane, uint32_t &mine)
    {
        if (next >= end && !dry) grab(head, n_total, want, lane);
        const uint32_t base1 = next, have1 = end - next, take1 = n_idle < have1 ? n_idle : have1;
        next += take1;
        uint32_t base2 = 0, take2 = 0;
        if (take1 < n_idle && !dry) {
            grab(head, n_total, want, lane);
            const uint32_t have2 = end - next;
            base2 = next;
            take2 = n_idle - take1 < have2 ? n_idle - take1 : have2;
            next += take2;
        }
        mine = rank < take1 ? base1 + rank : base2 + (rank - take1);
        return rank < take1 + take2;
    }
    __device__ __forceinline__ bool exhausted() const { return dry && next >= end; }
};

/* A launch begins with every wave asking its queue at once -- thousands of atomics on one word, served one after the other (a
 * third of a 1024^2 frame's 0.3 ms went by before the last wave had rays).  The waves that START with queue k (those with
 * floor(gw * iso_count / n_waves) == k) have their first chunk of rays without it: the one of rank r among them takes rays
 * [p(r) * chunk_rays, + chunk_rays), and the queue hands out what lies behind the starters' chunks.  p spreads neighbouring chunks
 * over the CUs when the list is short (the hard tiles of a frame are neighbours: Skull 1024^2 iso 80 0.304 -> 0.277 ms with
 * p(r) = r, 0.257 spread; C60 0.150 -> 0.123 -> 0.116) and is the identity for long lists, where neighbours share nodes in the
 * CU's L1 (Skull 4096^2: 0.853 -> 0.804, spread 0.812).  A wave that comes to a queue later takes only what the queue has. */
__device__ __forceinline__ void trace_seed(WaveChunk &chunk, uint32_t visit, uint32_t k, uint32_t iso_count, uint32_t gw, uint32_t n_waves,
                                           uint32_t chunk_rays, uint32_t n_active)
{
#ifndef VR_TRACE_NO_SEED
    const uint32_t w0 = (uint32_t)(((uint64_t)k * n_waves + iso_count - 1u) / iso_count);
    const uint32_t w1 = (uint32_t)(((uint64_t)(k + 1u) * n_waves + iso_count - 1u) / iso_count);
    const uint32_t n = w1 - w0, r = gw - w0; /* r < n only on the wave's first visit */
    const uint32_t per = kTraceThreads / 64u;
    const bool spread = chunk_rays == 64u && n % per == 0u;
    chunk.seed(spread ? (r % per) * (n / per) + r / per : r, n, chunk_rays, n_active);
    if (visit != 0u) { chunk.next = 0u; chunk.end = 0u; }
#endif
}

template <bool TAPS>
__global__ __launch_bounds__(kTraceThreads, kTraceWavesPerSimd) void vr_k_trace(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t n_top,
                                                            uint32_t stack_rows,
                                                            const float4 *__restrict__ ray_dt,
                                                            const uint32_t *__restrict__ ray_pix,
                                                            const uint32_t *__restrict__ n_active_ptr,
                                                            uint32_t *__restrict__ heads, uint32_t iso_begin,
                                                            uint32_t iso_count, uint64_t frame_pixels,
                                                            uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats,
                                                            uint32_t block_counters, VrTapsK taps)
{
    /* dynamic LDS: [path stacks: stack_rows x kTraceThreads words][n_top node words: heap positions 0 .. n_top-1] */
    extern __shared__ uint32_t s_dyn[];
    __shared__ unsigned long long s_acc_mem[kSlotIsoMax * 6u]; /* the block's counter sums of a short iso range (block_counters) */
    unsigned long long *const s_acc = block_counters ? s_acc_mem : nullptr;
    if (threadIdx.x < kSlotIsoMax * 6u) s_acc_mem[threadIdx.x] = 0ull;
    uint32_t *s_top = s_dyn + stack_rows * kTraceThreads;
    for (uint32_t i = threadIdx.x; i < n_top; i += kTraceThreads) s_top[i] = nodes[i];
    __syncthreads();
    LdsStack<kTraceThreads> stack;
    const TopFetch fetch(nodes, s_top, n_top, v.n_slots);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_active = *n_active_ptr;
    const uint32_t waves_per_block = kTraceThreads / 64u;
    const uint32_t n_waves = gridDim.x * waves_per_block, gw = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    /* spread the waves over the iso queues; each wave then walks all queues cyclically, so waves that
     * run out of work on a cheap iso value move on to help with the expensive ones */
    const uint32_t k0 = (uint32_t)(((uint64_t)gw * iso_count) / n_waves);
    const uint32_t chunk_rays = trace_chunk(n_active, n_waves);

    for (uint32_t visit = 0; visit < iso_count; visit++) {
        uint32_t k = k0 + visit;
        if (k >= iso_count) k -= iso_count;
        const uint32_t iso = iso_begin + k;
        uint8_t *frame = rgb + (uint64_t)k * frame_pixels * 3u;
        const uint64_t tap_base = (uint64_t)k * frame_pixels;

        VrLane L;
        WaveChunk chunk;
        trace_seed(chunk, visit, k, iso_count, gw, n_waves, chunk_rays, n_active);
        uint32_t pix = 0;
        bool alive = false, exhausted = false, popping = false;
        unsigned long long a_checked = 0;
        uint32_t a_hit = 0, a_miss = 0, a_moves = 0, a_esc = 0, a_err = 0;

        for (;;) {
            /* ---- refill idle lanes from this iso value's queue ---- */
            const unsigned long long idle = __ballot(!alive);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (!exhausted && n_idle >= kRefillMin) {
                uint32_t mine;
                const bool got = chunk.take(n_idle, (uint32_t)__popcll(idle & ((1ull << lane) - 1ull)), &heads[k], n_active, chunk_rays,
                                            lane, mine);
                if (!alive && got) {
                    const float4 r = ray_dt[mine];
                    pix = ray_pix[mine];
                    vr_lane_begin(L, stack, v, r.x, r.y, r.z, r.w);
                    alive = true;
                    popping = false;
                }
                exhausted = chunk.exhausted();
            }
            if (__ballot(alive) == 0ull) break; /* only possible once the queue is exhausted */

            if (alive) {
                uint32_t go = VR_GO_MORE;
                if (!popping) {
                    /* ---- descend while the node's density range holds the iso value (bounded) ---- */
#ifdef VR_TRACE_NO_AHEAD
                    go = vr_lane_descend(L, iso, fetch, stack, kDescendTrips);
#else
                    { VrNoIsoTrack none; go = vr_lane_descend_ahead(L, iso, fetch, stack, kDescendTrips, none); }
#endif
                }
                if (go == VR_GO_HIT) { /* shade and store (src/raytrace.zig:56-61) */
                    const uint8_t g = vr_shade(L.t);
                    frame[(uint64_t)pix * 3u + 0u] = g;
                    frame[(uint64_t)pix * 3u + 1u] = g;
                    frame[(uint64_t)pix * 3u + 2u] = g;
                    a_hit++;
                    alive = false;
                    if (TAPS) {
                        if (taps.cell_flag) taps.cell_flag[tap_base + pix] = vr_unpack_cell(L.cell) | (VR_FLAG_HIT << 24);
                        if (taps.node) taps.node[tap_base + pix] = L.g - 1u;
                    }
                } else if (go == VR_GO_EXCLUDED) {
                    a_moves++;
                    bool in_root = vr_lane_move(L, v, a_esc);
                    if (L.checked > VR_MAX_CHECKED) { in_root = false; a_err++; }
                    if (in_root) {
                        popping = true;
                    } else { /* left the volume: a miss, the pixel stays black (src/raytrace.zig:66-68) */
                        a_miss++;
                        alive = false;
                        if (TAPS) {
                            if (taps.cell_flag) taps.cell_flag[tap_base + pix] = VR_FLAG_MISS << 24;
                            if (taps.node) taps.node[tap_base + pix] = 0u;
                        }
                    }
                }
                if (alive && popping) {
                    /* ---- pop to the lowest ancestor that holds the new point (bounded) ---- */
                    popping = !vr_lane_pop(L, stack, kPopTrips);
                }
                if (!alive) {
                    a_checked += L.checked;
                    if (TAPS) {
                        if (taps.t) taps.t[tap_base + pix] = L.t;
                        if (taps.checked) taps.checked[tap_base + pix] = L.checked;
                    }
                }
            }
        }

        /* ---- per-iso counters ---- */
        if (stats) wave_stats_out(stats, s_acc, k, lane, a_checked, a_hit, a_miss, a_esc, a_moves, a_err);
    }
    if (stats && s_acc) block_stats_out(stats, s_acc, iso_count);
}

/* --------------------------------------------------------- pipeline: iso sweep kernel ---- */

/* The sweep over iso values with traversals shared between them.  A lane owns one ray and a block of iso
 * values; it traces the lowest open iso value while tracking the largest iso value that cannot change
 * the outcome of any density-range test on the way (VrIsoUpper), gives the result to every iso value up
 * to there -- pixels, taps and counters are exactly those of a separate traversal per iso value -- and
 * goes on behind it.  On the Skull sweep a ray has 28 distinct traversals in 256 iso values.
 *
 * Work items are (ray, iso block) pairs from one queue; the iso range is cut into as many blocks as it
 * takes to have several items per lane of the grid (a ray's 256 iso values cost between 5 and >100
 * traversals, so whole rays would balance badly when there are about as many rays as lanes).
 *
 * Per-iso counters go through difference arrays in LDS: +x at the first iso value of an interval, -x
 * behind its last; a prefix sum per block at the end, then one global atomic per iso value and counter. */
/* (ray, iso block) items per lane of the grid that the iso range is cut for: a launch argument (vr_internal.h:
 * VR_SWEEP_ITEMS_DEFAULT; the context tunes it per workload from the kernel's own reports) */
#ifndef VR_SWEEP_DESCEND_TRIPS
#define VR_SWEEP_DESCEND_TRIPS 5
#endif
#ifndef VR_SWEEP_POP_TRIPS
#define VR_SWEEP_POP_TRIPS 7
#endif
#ifndef VR_SWEEP_REFILL_MIN
#define VR_SWEEP_REFILL_MIN 16
#endif
#ifndef VR_SWEEP_MAX_BLOCKS
#define VR_SWEEP_MAX_BLOCKS 32
#endif
#ifndef VR_SWEEP_STEALS
#define VR_SWEEP_STEALS 2
#endif
#ifndef VR_SWEEP_STEAL_MIN
#define VR_SWEEP_STEAL_MIN 2
#endif
#ifndef VR_SWEEP_KEEP
#define VR_SWEEP_KEEP 2 /* lane states kept for the next walks of the same ray to carry on from */
#endif
/* once the queue is empty: hand-overs of half an iso range to an idle lane per wave and round, and the
 * shortest range (iso values behind the current one) worth splitting */
constexpr uint32_t kSweepSteals = VR_SWEEP_STEALS, kSweepStealMin = VR_SWEEP_STEAL_MIN;
constexpr uint32_t kSweepDescendTrips = VR_SWEEP_DESCEND_TRIPS, kSweepPopTrips = VR_SWEEP_POP_TRIPS; /* lanes of a wave are less alike here */
constexpr uint32_t kSweepRefillMin = VR_SWEEP_REFILL_MIN; /* items last long here: refills are rare */
constexpr uint32_t kSweepCounters = 5; /* hit, miss, moves, escape help, errors (u32 each); checks are u64 */

__host__ __device__ __forceinline__ uint32_t sweep_stat_words(uint32_t iso_count)
{
    const uint32_t words = (2u + kSweepCounters) * (iso_count + 1u) + 1u; /* + the block's traversal count */
    return words + (words & 1u);
}

/* Shared beginnings of a ray's walks.  The walk for the next open iso value, upper + 1, makes the same tests with
 * the same outcomes as the one just done up to the first node whose bound equals `upper`.  The reference returns to
 * the root after every MOVE (src/raytrace.zig:246), so everything before the last such return ahead of that node --
 * ahead of the stretch of the walk in which `upper` last went down -- is common to both, and the next walk may start
 * from the lane's state at that return instead of from the ray's entry point: t, cell, and the checks, moves and escape
 * helps so far, with `upper` as it stood there.  States older than that one lie in the common part, where they are the
 * next walk's own, so a short stack of them (newest on top) serves a run of walks. */
struct SweepKept {
    float t;
    uint32_t cell, checked, moves, esc, upper; /* upper = kNone: empty */
    static constexpr uint32_t kNone = 0xffffffffu;
};

/* ---- the sweep kernel's lane code for the two tree formats ---- */
struct WideFetch {
    const uint32_t *nodes;
    uint32_t last_pair; /* the last index whose right neighbour is still inside the array */
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const { return nodes[idx]; }
    /* the words at 2i + 1 and 2i + 2 in one 8-byte read (4-byte aligned) */
    __device__ __forceinline__ VrNodePair children(uint32_t idx) const
    {
        struct __attribute__((packed, aligned(4))) Pair { uint32_t a, b; };
        uint32_t c = 2u * idx + 1u;
        c = c < last_pair ? c : last_pair;
        const Pair w = *(const Pair *)(nodes + c);
        return VrNodePair{ w.a, w.b };
    }
};

template <uint32_t STRIDE>
struct LdsPath {
    uint32_t *col; /* this thread's column of the [VR_WIDE_MAX_DEPTH][STRIDE] path store */
    __device__ __forceinline__ uint32_t get(uint32_t d) const { return col[d * STRIDE]; }
    __device__ __forceinline__ void set(uint32_t d, uint32_t w) { col[d * STRIDE] = w; }
};


__device__ __forceinline__ float lane_f(float x, uint32_t src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); }

/* trees within the reference's range: flattened array, top levels in LDS, path stack of bound words */
struct SweepNarrow {
    typedef VrLane Lane;
    typedef LdsStack<kTraceThreads> Store;
    typedef TopFetch Fetch;
    static constexpr uint32_t kFlagShift = 24u;
    __device__ static __forceinline__ Store store(uint32_t *) { return Store(); }
    __device__ static __forceinline__ Fetch fetch(const uint32_t *nodes, const uint32_t *s_top, uint32_t n_top, uint64_t n_slots) { return TopFetch(nodes, s_top, n_top, n_slots); }
    __device__ static __forceinline__ void begin(Lane &L, Store &st, const VrViewK &v, float dx, float dy, float dz, float t) { vr_lane_begin(L, st, v, dx, dy, dz, t); }
    __device__ static __forceinline__ uint32_t cell(const Lane &L) { return L.cell; }
    __device__ static __forceinline__ void restart(Lane &L, Store &st, const VrViewK &v, float t0, uint32_t c0) { vr_lane_restart(L, st, v, t0, c0); }
    __device__ static __forceinline__ uint32_t descend(Lane &L, const VrViewK &, uint32_t iso, const Fetch &f, Store &st, VrIsoUpper &track)
    {
#ifdef VR_SWEEP_NO_AHEAD
        return vr_lane_descend(L, iso, f, st, kSweepDescendTrips, track);
#else
        return vr_lane_descend_ahead(L, iso, f, st, kSweepDescendTrips, track);
#endif
    }
    __device__ static __forceinline__ bool move(Lane &L, const VrViewK &v, uint32_t &esc) { return vr_lane_move(L, v, esc); }
    __device__ static __forceinline__ bool pop(Lane &L, Store &st) { return vr_lane_pop(L, st, kSweepPopTrips); }
    __device__ static __forceinline__ uint32_t cell_tap(const Lane &L) { return vr_unpack_cell(L.cell); }
    __device__ static __forceinline__ uint32_t node_tap(const Lane &L) { return L.g - 1u; }
    /* the victim's ray, read by every lane, kept by the thief */
    __device__ static __forceinline__ void adopt(Lane &L, uint32_t victim, bool thief)
    {
        const float dx = lane_f(L.dx, victim), dy = lane_f(L.dy, victim), dz = lane_f(L.dz, victim);
        const float rx = lane_f(L.rx, victim), ry = lane_f(L.ry, victim), rz = lane_f(L.rz, victim);
        const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane((int)L.posmask, victim);
        if (thief) { L.dx = dx; L.dy = dy; L.dz = dz; L.rx = rx; L.ry = ry; L.rz = rz; L.posmask = pm; }
    }
};

/* volumes beyond 256 voxels per axis: the wide node array as it is, one path word per level in LDS */
struct SweepWide {
    typedef VrWideLane Lane;
    typedef LdsPath<kTraceThreads> Store;
    typedef WideFetch Fetch;
    static constexpr uint32_t kFlagShift = VR_WIDE_FLAG_SHIFT;
    __device__ static __forceinline__ Store store(uint32_t *s_dyn) { return Store{ s_dyn + threadIdx.x }; }
    __device__ static __forceinline__ Fetch fetch(const uint32_t *nodes, const uint32_t *, uint32_t, uint64_t n_slots) { return WideFetch{ nodes, n_slots >= 2u ? (uint32_t)(n_slots - 2u) : 0u }; }
    __device__ static __forceinline__ void begin(Lane &L, Store &, const VrViewK &v, float dx, float dy, float dz, float t) { vr_wide_begin(L, v, dx, dy, dz, t); }
    __device__ static __forceinline__ uint32_t cell(const Lane &L) { return L.c; }
    __device__ static __forceinline__ void restart(Lane &L, Store &, const VrViewK &v, float t0, uint32_t c0) { vr_wide_restart(L, v, t0, c0); }
    __device__ static __forceinline__ uint32_t descend(Lane &L, const VrViewK &v, uint32_t iso, const Fetch &f, Store &st, VrIsoUpper &track)
    {
#ifdef VR_WIDE_NO_AHEAD
        return vr_wide_descend(L, v, iso, f, st, kSweepDescendTrips, track);
#else
        return vr_wide_descend_ahead(L, v, iso, f, st, kSweepDescendTrips, track);
#endif
    }
    __device__ static __forceinline__ bool move(Lane &L, const VrViewK &v, uint32_t &esc) { return vr_wide_move(L, v, esc); }
    __device__ static __forceinline__ bool pop(Lane &L, Store &st) { return vr_wide_pop(L, st, kSweepPopTrips); }
    __device__ static __forceinline__ uint32_t cell_tap(const Lane &L) { return L.c; }
    __device__ static __forceinline__ uint32_t node_tap(const Lane &L) { return L.idx; }
    __device__ static __forceinline__ void adopt(Lane &L, uint32_t victim, bool thief)
    {
        const float dx = lane_f(L.dx, victim), dy = lane_f(L.dy, victim), dz = lane_f(L.dz, victim);
        const float rx = lane_f(L.rx, victim), ry = lane_f(L.ry, victim), rz = lane_f(L.rz, victim);
        if (thief) { L.dx = dx; L.dy = dy; L.dz = dz; L.rx = rx; L.ry = ry; L.rz = rz; }
    }
};

/* Words of the 256-byte block behind the ray count (n_active): [16..27] the sweep kernels' report, [32] the union kernel's count of
 * iso runs that left its walks, [33] set by the union kernel when it could not finish a call exactly -- its list of such runs was
 * full, or a front passed the bound on its MOVEs: the frames it leaves are then incomplete, vr_k_union_fix clears them and the
 * counters instead of finishing the list, and a launch of vr_k_sweep that is enqueued behind every union launch (and returns at
 * once while the word is zero) renders the whole call again. */
constexpr uint32_t kUnionRedoWord = 33u;

template <bool TAPS, class OPS>
__global__ __launch_bounds__(kTraceThreads, kTraceWavesPerSimd) void vr_k_sweep(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t n_top,
                                                            uint32_t stack_rows,
                                                            const float4 *__restrict__ ray_dt,
                                                            const uint32_t *__restrict__ ray_pix,
                                                            const uint32_t *__restrict__ n_active_ptr,
                                                            uint32_t *__restrict__ head, uint32_t iso_begin,
                                                            uint32_t iso_count, uint64_t frame_pixels,
                                                            uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats,
                                                            VrTapsK taps, uint32_t items_per_lane, uint32_t launch_seq,
                                                            const uint32_t *__restrict__ run_if)
{
    if (run_if && *run_if == 0u) return; /* the launch behind a union launch: only when that one asked for it */
    /* dynamic LDS: [path stacks][check sums: (iso_count + 1) x u64][counters: 5 x (iso_count + 1) x u32][top node words] */
    extern __shared__ uint32_t s_dyn[];
    unsigned long long *s_checks = (unsigned long long *)(s_dyn + stack_rows * kTraceThreads);
    uint32_t *s_cnt = (uint32_t *)(s_checks + (iso_count + 1u));
    uint32_t *s_top = s_dyn + stack_rows * kTraceThreads + sweep_stat_words(iso_count);
    const uint32_t n_k = iso_count + 1u;
    for (uint32_t i = threadIdx.x; i < sweep_stat_words(iso_count); i += kTraceThreads) s_dyn[stack_rows * kTraceThreads + i] = 0u;
    for (uint32_t i = threadIdx.x; i < n_top; i += kTraceThreads) s_top[i] = nodes[i];
    __syncthreads();
    typename OPS::Store stack = OPS::store(s_dyn);
    const typename OPS::Fetch fetch = OPS::fetch(nodes, s_top, n_top, v.n_slots);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_active = *n_active_ptr;
    /* iso blocks: enough (ray, block) items for several per lane of the grid.  Items are numbered group by
     * group of 64 consecutive rays (one tile of the image): the 64 rays on the first iso block, the same rays
     * on the next block, ... */
    const uint64_t want = (uint64_t)gridDim.x * kTraceThreads * items_per_lane;
    uint32_t nb = (uint32_t)((want + n_active - 1u) / (n_active ? n_active : 1u));
    nb = nb < 1u ? 1u : (nb > VR_SWEEP_MAX_BLOCKS ? VR_SWEEP_MAX_BLOCKS : nb);
    nb = nb > iso_count ? iso_count : nb;
    const uint32_t block_isos = (iso_count + nb - 1u) / nb, n_blocks = (iso_count + block_isos - 1u) / block_isos;
    const uint32_t n_items = ((n_active + 63u) & ~63u) * n_blocks; /* more than one block only while n_active is far below 2^32 / 64 */
    const uint64_t frame_bytes = frame_pixels * 3u;

    /* the kernel's report on itself (read back with vr_sweep_info_read): traversals done, and the 100 MHz ticks at
     * which the first wave started, the first wave found the queue empty, and the last wave ended */
    unsigned long long *info = (unsigned long long *)(n_active_ptr + 16);
    const unsigned long long tick_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long tick_empty = 0;
    uint32_t n_walks = 0;
    typename OPS::Lane L;
    WaveChunk chunk;
    VrIsoUpper track;
    track.upper = 255u;
    SweepKept kept[VR_SWEEP_KEEP + 1]; /* [0]: the newest; constant indices only, so these are registers */
#pragma unroll
    for (int q = 0; q <= VR_SWEEP_KEEP; q++) kept[q].upper = SweepKept::kNone;
    uint32_t seg_checked = 0, seg_upper = 255u; /* at the walk's last return to the root */
    uint32_t pix = 0, iso = 0, iso_last = 0, cell0 = 0, moves = 0, esc = 0;
    float t0 = 0.0f;
    bool alive = false, exhausted = false, popping = false;

    for (;;) {
        /* ---- refill idle lanes from the item queue ---- */
        const unsigned long long idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!exhausted && n_idle >= kSweepRefillMin) {
            uint32_t mine;
            const bool got = chunk.take(n_idle, (uint32_t)__popcll(idle & ((1ull << lane) - 1ull)), head, n_items, VR_SWEEP_CHUNK ? VR_SWEEP_CHUNK : n_idle, lane, mine);
            if (!alive && got) {
                const uint32_t grp = mine / (64u * n_blocks), in = mine - grp * 64u * n_blocks;
                const uint32_t blk = in >> 6, ray = grp * 64u + (in & 63u);
                const float4 r = ray_dt[ray < n_active ? ray : 0u];
                pix = ray_pix[ray < n_active ? ray : 0u];
                OPS::begin(L, stack, v, r.x, r.y, r.z, r.w);
                t0 = r.w;
                cell0 = OPS::cell(L);
                iso = iso_begin + blk * block_isos;
                iso_last = iso_begin + (blk * block_isos + block_isos < iso_count ? blk * block_isos + block_isos : iso_count) - 1u;
                track.upper = 255u;
                moves = 0; esc = 0;
#pragma unroll
                for (int q = 0; q < VR_SWEEP_KEEP; q++) kept[q].upper = SweepKept::kNone;
                seg_checked = 0; seg_upper = 255u;
                alive = ray < n_active; /* the last group of 64 may be short */
                popping = false;
            }
            exhausted = chunk.exhausted();
            if (exhausted && !tick_empty) tick_empty = __builtin_amdgcn_s_memrealtime();
        }
        if (__ballot(alive) == 0ull) break; /* only possible once the queue is exhausted */
        /* ---- queue empty: idle lanes take over the upper half of the longest iso range left in the wave ---- */
        if (exhausted) {
#pragma unroll 1
            for (uint32_t steal = 0; steal < kSweepSteals; steal++) {
                const unsigned long long idle_now = __ballot(!alive);
                if (idle_now == 0ull) break;
                const uint32_t rem = alive ? iso_last - iso : 0u; /* iso values behind the one being traced */
                const uint32_t best = wave_max((rem << 6) | lane);
                if ((best >> 6) < kSweepStealMin) break;
                const uint32_t victim = __builtin_amdgcn_readfirstlane(best & 63u);
                const uint32_t thief = (uint32_t)__ffsll((long long)idle_now) - 1u;
                const float s_t0 = lane_f(t0, victim);
                const uint32_t s_c0 = (uint32_t)__builtin_amdgcn_readlane((int)cell0, victim);
                const uint32_t s_px = (uint32_t)__builtin_amdgcn_readlane((int)pix, victim);
                const uint32_t s_iso = (uint32_t)__builtin_amdgcn_readlane((int)iso, victim);
                const uint32_t s_last = (uint32_t)__builtin_amdgcn_readlane((int)iso_last, victim);
                const uint32_t mid = s_iso + ((s_last - s_iso) >> 1); /* the victim keeps [iso, mid] */
                if (lane == victim) iso_last = mid;
                OPS::adopt(L, victim, lane == thief);
                if (lane == thief) {
                    t0 = s_t0; cell0 = s_c0; pix = s_px;
                    iso = mid + 1u; iso_last = s_last;
                    OPS::restart(L, stack, v, t0, cell0);
                    track.upper = 255u;
                    moves = 0; esc = 0;
#pragma unroll
                    for (int q = 0; q < VR_SWEEP_KEEP; q++) kept[q].upper = SweepKept::kNone;
                    seg_checked = 0; seg_upper = 255u;
                    alive = true;
                    popping = false;
                }
            }
        }

        if (alive) {
            uint32_t go = VR_GO_MORE;
            /* ---- descend while the node's density range holds the iso value (bounded) ---- */
            if (!popping) go = OPS::descend(L, v, iso, fetch, stack, track);
            bool done = false, hit = false;
            uint32_t err = 0;
            if (VR_SWEEP_KEEP && go != VR_GO_MORE && track.upper < seg_upper) {
                /* `upper` went down since the last return to the root: the state at that return is the newest one
                 * the next walk shares (t, cell, moves and esc have not changed since) */
#pragma unroll
                for (int q = VR_SWEEP_KEEP - 1; q > 0; q--) kept[q] = kept[q - 1];
                kept[0].t = L.t; kept[0].cell = OPS::cell(L); kept[0].checked = seg_checked;
                kept[0].moves = moves; kept[0].esc = esc; kept[0].upper = seg_upper;
            }
            if (go == VR_GO_HIT) {
                done = true; hit = true;
            } else if (go == VR_GO_EXCLUDED) {
                moves++;
                seg_checked = L.checked; seg_upper = track.upper;
                bool in_root = OPS::move(L, v, esc);
                if (L.checked > VR_MAX_CHECKED) { in_root = false; err = 1; }
                if (in_root) popping = true; else done = true; /* left the volume: a miss (src/raytrace.zig:66-68) */
            }
            /* ---- pop to the lowest ancestor that holds the new point (bounded) ---- */
            if (popping && !done) popping = !OPS::pop(L, stack);

            if (done) {
                n_walks++;
                /* ---- the traversal stands for every iso value from `iso` to `e` ---- */
                const uint32_t e = track.upper < iso_last ? track.upper : iso_last;
                const uint32_t k0 = iso - iso_begin, k1 = e + 1u - iso_begin;
                atomicAdd(&s_checks[k0], (unsigned long long)L.checked);
                atomicAdd(&s_checks[k1], 0ull - (unsigned long long)L.checked);
                uint32_t *c = s_cnt + (hit ? 0u : n_k);
                atomicAdd(&c[k0], 1u); atomicAdd(&c[k1], 0u - 1u);
                if (moves) { atomicAdd(&s_cnt[2u * n_k + k0], moves); atomicAdd(&s_cnt[2u * n_k + k1], 0u - moves); }
                if (esc) { atomicAdd(&s_cnt[3u * n_k + k0], esc); atomicAdd(&s_cnt[3u * n_k + k1], 0u - esc); }
                if (err) { atomicAdd(&s_cnt[4u * n_k + k0], 1u); atomicAdd(&s_cnt[4u * n_k + k1], 0u - 1u); }
                if (hit) { /* shade and store (src/raytrace.zig:56-61), one frame per iso value */
                    const uint8_t g = vr_shade(L.t);
                    uint8_t *p = rgb + (uint64_t)k0 * frame_bytes + (uint64_t)pix * 3u;
                    for (uint32_t k = k0; k < k1; k++, p += frame_bytes) { p[0] = g; p[1] = g; p[2] = g; }
                }
                if (TAPS) {
                    const uint32_t cf = hit ? (OPS::cell_tap(L) | (VR_FLAG_HIT << OPS::kFlagShift)) : (VR_FLAG_MISS << OPS::kFlagShift);
                    const uint32_t nd = hit ? OPS::node_tap(L) : 0u;
                    for (uint32_t k = k0; k < k1; k++) {
                        const uint64_t at = (uint64_t)k * frame_pixels + pix;
                        if (taps.cell_flag) taps.cell_flag[at] = cf;
                        if (taps.node) taps.node[at] = nd;
                        if (taps.t) taps.t[at] = L.t;
                        if (taps.checked) taps.checked[at] = L.checked;
                    }
                }
                iso = e + 1u;
                if (iso > iso_last) {
                    alive = false;
                } else { /* same ray, next open iso value: from the newest kept state, else from the entry point */
                    const bool on = VR_SWEEP_KEEP && kept[0].upper != SweepKept::kNone;
                    OPS::restart(L, stack, v, on ? kept[0].t : t0, on ? kept[0].cell : cell0);
                    L.checked = on ? kept[0].checked : 0u;
                    track.upper = on ? kept[0].upper : 255u;
                    moves = on ? kept[0].moves : 0u; esc = on ? kept[0].esc : 0u;
                    seg_checked = L.checked; seg_upper = track.upper;
#pragma unroll
                    for (int q = 0; q < VR_SWEEP_KEEP; q++) kept[q] = kept[q + 1]; /* [KEEP] stays empty */
                    popping = false;
                }
            }
        }
    }

    {
        const uint32_t w_walks = wave_sum(n_walks);
        if (lane == 0) {
            atomicAdd(&s_cnt[kSweepCounters * n_k], w_walks); /* the word behind the counter arrays */
            if ((threadIdx.x >> 6) == 0u) { /* one wave per block reports the times */
                atomicMin(&info[2], tick_begin);
                if (tick_empty) atomicMin(&info[3], tick_empty);
            }
        }
    }
    /* ---- per-iso counters: prefix sums of the difference arrays, then one atomic per iso value and counter ---- */
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&info[0], (unsigned long long)s_cnt[kSweepCounters * n_k]);
        if (blockIdx.x == 0) { info[1] = n_items; info[5] = launch_seq; }
        atomicMax(&info[4], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    if (threadIdx.x < kSweepCounters) {
        uint32_t *c = s_cnt + threadIdx.x * n_k;
        uint32_t run = 0;
        for (uint32_t k = 0; k < iso_count; k++) { run += c[k]; c[k] = run; }
    } else if (threadIdx.x == kSweepCounters) {
        unsigned long long run = 0;
        for (uint32_t k = 0; k < iso_count; k++) { run += s_checks[k]; s_checks[k] = run; }
    }
    __syncthreads();
    if (stats) {
        for (uint32_t k = threadIdx.x; k < iso_count; k += kTraceThreads) {
            vr_render_stats *s = stats + k;
            add_stat(&s->nodes_checked, s_checks[k]);
            add_stat(&s->hit_count, s_cnt[k]);
            add_stat(&s->miss_count, s_cnt[n_k + k]);
            add_stat(&s->moves, s_cnt[2u * n_k + k]);
            add_stat(&s->escape_help_events, s_cnt[3u * n_k + k]);
            add_stat(&s->errors, s_cnt[4u * n_k + k]);
        }
    }
}


/* ------------------------------------------------------------------ union walk ---- */

/* vr_k_union: the sweep as ONE traversal per ray (lane code and the idea: vr_trace_union.h).  Work items are rays;
 * a lane keeps the 256 done bits of its ray in LDS next to its path stack.  Per-iso counters go through the same
 * difference arrays in LDS as in vr_k_sweep.  Iso values whose own MOVE does not land on the front's state (a few in
 * a thousand rays) are written to a list as (ray, run of iso values, own state) and finished by vr_k_union_fix.
 * Needs node ranges that are nested along every path (a node's range lies inside its parent's), which holds for
 * every tree built from a volume; uploaded trees are checked (vr_host_ranges_nested). */
#ifndef VR_UNION_DESCEND_TRIPS
#define VR_UNION_DESCEND_TRIPS 5
#endif
#ifndef VR_UNION_POP_TRIPS
#define VR_UNION_POP_TRIPS 11 /* levels up per round (Skull sweep, with one exit-plane check per pop: 1.000 ms with 9, 0.986 with 11, 0.990 with 13;
                               * C60 0.733 / 0.724 with 9 / 11, 256^3 stand-in 2.51 / 2.49) */
#endif
#ifndef VR_UNION_FINE_SHIFT
#define VR_UNION_FINE_SHIFT 4 /* the first 1/16 of the tiles of a settled order ... */
#endif
#ifndef VR_UNION_FINE_MUL
#define VR_UNION_FINE_MUL 2   /* ... are cut into twice as many blocks of iso values (1: one cut for all).  Skull / C60 / 256^3 stand-in sweep,
                               * kernel ms: one cut 0.846 / 0.684 / 2.20; first 1/16 x 2: 0.845 / 0.609 / 2.11; 1/8 x 2: 0.839 / 0.617 / 2.11;
                               * 1/4 x 2: 0.875 / 0.634 / 2.13; 1/16 x 4: 0.874 / 0.630 / 2.21 */
#endif
#ifndef VR_UNION_REFILL_MIN
#define VR_UNION_REFILL_MIN 64 /* idle lanes from which a wave takes new items.  64 = only when the whole wave is idle: a wave then always holds
                                * the 64 neighbouring rays of ONE tile for ONE block of iso values, fronts that read the same nodes and end at about
                                * the same time.  Skull / C60 / 256^3 stand-in sweep, kernel ms: 24: 0.905 / 0.717 / 2.48, 32: 0.884 / 0.711 / 2.53,
                                * 48: 0.867 / 0.701 / 2.64, 56: 0.875 / 0.701 / 2.73 (idle lanes without the coherence), 64: 0.867 / 0.697 / 2.34;
                                * 32 iso values of the 1024^3 gyroid at 4096^2: 10.19 / 9.80 with 24 / 64 */
#endif
constexpr uint32_t kUnionDescendTrips = VR_UNION_DESCEND_TRIPS, kUnionPopTrips = VR_UNION_POP_TRIPS;
constexpr uint32_t kUnionRefillMin = VR_UNION_REFILL_MIN;
constexpr uint32_t kParkMoveShift = 35u; /* see UnionSink::park */
constexpr uint32_t kUnionFineShift = VR_UNION_FINE_SHIFT, kUnionFineMul = VR_UNION_FINE_MUL;
/* MOVEs of one front: the safety valve (VR_MAX_CHECKED in the other kernels).  A valid front makes one MOVE per cell its ray crosses
 * at most -- a few thousand on the largest volume; a front with a ray on which the reference's loop does not end would go on for
 * ever: it is given up here (half a second) and the call rendered again by the kernel that bounds every ray by its checks. */
constexpr uint32_t kUnionMaxStates = 1u << 17;

struct VrUnionRecord { uint32_t ray, run, t_bits, cell, esc, pad[3]; }; /* run = r0 | r1 << 8 | in_root << 16 */

template <uint32_t STRIDE>
struct LdsDone {
    uint32_t *col; /* this thread's column of the [8][STRIDE] done words */
    __device__ __forceinline__ uint32_t word(uint32_t w) const { return col[w * STRIDE]; }
    __device__ __forceinline__ void set(uint32_t w, uint32_t bits) { col[w * STRIDE] |= bits; }
    __device__ __forceinline__ void unset(uint32_t w, uint32_t bits) { col[w * STRIDE] &= ~bits; }
    __device__ __forceinline__ void clear(uint32_t w) { col[w * STRIDE] = 0u; }
    __device__ __forceinline__ void fill(uint32_t w) { col[w * STRIDE] = 0xffffffffu; }
};

/* STATS = false: the caller passed no counters -- the frames are all it wants -- and everything that only serves them is compiled
 * out: the parked sets of every level, the checks and counts of hits, misses and escape helps (Skull sweep kernel 0.755 -> 0.645
 * ms, C60 0.565 -> 0.404, 256^3 stand-in 1.95 -> 1.41).  The reference's renderImage returns nodes_checked (src/raytrace.zig:21,105):
 * a caller that stands in for it passes counters. */
template <bool STATS>
struct UnionSinkT {
    unsigned long long *s_checks;
    uint32_t *s_cnt;
    uint32_t n_k, iso_begin;
    uint8_t *rgb;
    uint64_t frame_bytes;
    uint32_t pix, ray;
    uint32_t *except_count; /* [0] the count, [1] (kUnionRedoWord): set when the call has to be rendered again */
    VrUnionRecord *except_list;
    uint32_t except_cap;
    /* iso values that left the front, kept by the lane until an idle lane of the wave takes them or the ray's current
     * front is finished and the lane walks them itself, as a front of their own; three such states of up to four runs each
     * fit, more go to the list */
    bool pend, pend2, pend3;
    float pend_t, pend2_t, pend3_t;
    uint32_t pend_cell, pend2_cell, pend3_cell;
    VrUnionRuns pend_runs, pend2_runs, pend3_runs;
    __device__ __forceinline__ void pend_shift()
    {
        pend = pend2; pend_t = pend2_t; pend_cell = pend2_cell; pend_runs = pend2_runs;
        pend2 = pend3; pend2_t = pend3_t; pend2_cell = pend3_cell; pend2_runs = pend3_runs;
        pend3 = false;
    }
    __device__ __forceinline__ void add(uint32_t counter, uint32_t r0, uint32_t r1, uint32_t amount)
    {
        if constexpr (!STATS) return;
        atomicAdd(&s_cnt[counter * n_k + (r0 - iso_begin)], amount);
        atomicAdd(&s_cnt[counter * n_k + (r1 + 1u - iso_begin)], 0u - amount);
    }
    __device__ __forceinline__ void checks(uint32_t r0, uint32_t r1, uint32_t amount)
    {
        if constexpr (!STATS) return;
        atomicAdd(&s_checks[r0 - iso_begin], (unsigned long long)amount);
        atomicAdd(&s_checks[r1 + 1u - iso_begin], 0ull - (unsigned long long)amount);
    }
    /* excluded at a node: the checks of this round and the MOVE out of the node's box, in one 64-bit add each way (the MOVE
     * count rides in the bits from kParkMoveShift on of the check sums until the block's prefix sums split them).  A parked set
     * has checked at most 32 nodes (the deepest level + 1) per MOVE, so 29 bits of MOVEs and 35 bits of checks fill up together:
     * exact while one block counts fewer than 2^29 such MOVEs for one iso value in one launch, which 8192 x 8192 rays with 2 000
     * MOVEs each, spread over the 256 blocks of the grid, stay below. */
    __device__ __forceinline__ void park(uint32_t r0, uint32_t r1, uint32_t n_checks)
    {
        if constexpr (!STATS) return;
        const unsigned long long both = (unsigned long long)n_checks + (1ull << kParkMoveShift);
        atomicAdd(&s_checks[r0 - iso_begin], both);
        atomicAdd(&s_checks[r1 + 1u - iso_begin], 0ull - both);
    }
    __device__ __forceinline__ void hit(uint32_t r0, uint32_t r1, uint32_t n_checks, float t, uint32_t, uint32_t)
    {
        checks(r0, r1, n_checks);
        add(0u, r0, r1, 1u);
        const uint8_t g = vr_shade(t); /* shade and store (src/raytrace.zig:56-61), one frame per iso value */
        uint8_t *p = rgb + (uint64_t)(r0 - iso_begin) * frame_bytes + (uint64_t)pix * 3u;
        VR_WC(16);
        uint32_t n = r1 - r0 + 1u; /* two frames per turn: half the loop's own instructions (-1 % of the kernel) */
        for (; n >= 2u; n -= 2u) {
            VR_WC(6);
            uint8_t *q = p + frame_bytes;
            p[0] = g; p[1] = g; p[2] = g;
            q[0] = g; q[1] = g; q[2] = g;
            p = q + frame_bytes;
        }
        if (n) { p[0] = g; p[1] = g; p[2] = g; }
    }
    __device__ __forceinline__ void miss(uint32_t r0, uint32_t r1) { add(1u, r0, r1, 1u); }
    __device__ __forceinline__ void esc(uint32_t r0, uint32_t r1, uint32_t n) { add(3u, r0, r1, n); }
    __device__ __forceinline__ void error(uint32_t r0, uint32_t r1) { add(4u, r0, r1, 1u); }
    __device__ __forceinline__ void except(uint32_t r0, uint32_t r1, float t, uint32_t cell, bool in_root, uint32_t n_esc)
    {
        if (!in_root) { /* their own MOVE left the volume: misses (src/raytrace.zig:66-68) */
            if (n_esc) esc(r0, r1, n_esc);
            miss(r0, r1);
            return;
        }
        const uint32_t tb = __builtin_bit_cast(uint32_t, t);
        if (!pend) {
            pend = true; pend_t = t; pend_cell = cell; pend_runs.clear();
        }
        if (cell == pend_cell && tb == __builtin_bit_cast(uint32_t, pend_t) && pend_runs.room()) {
            if (n_esc) esc(r0, r1, n_esc);
            pend_runs.add(r0, r1);
            return;
        }
        if (!pend2) {
            pend2 = true; pend2_t = t; pend2_cell = cell; pend2_runs.clear();
        }
        if (cell == pend2_cell && tb == __builtin_bit_cast(uint32_t, pend2_t) && pend2_runs.room()) {
            if (n_esc) esc(r0, r1, n_esc);
            pend2_runs.add(r0, r1);
            return;
        }
        if (!pend3) {
            pend3 = true; pend3_t = t; pend3_cell = cell; pend3_runs.clear();
        }
        if (cell == pend3_cell && tb == __builtin_bit_cast(uint32_t, pend3_t) && pend3_runs.room()) {
            if (n_esc) esc(r0, r1, n_esc);
            pend3_runs.add(r0, r1);
            return;
        }
        const uint32_t at = atomicAdd(except_count, 1u);
        if (at >= except_cap) { except_count[1] = 1u; return; } /* the list is full (VR_UNION_EXCEPT_CAP runs): the call is rendered again */
        VrUnionRecord rec;
        rec.ray = ray; rec.run = r0 | (r1 << 8) | ((in_root ? 1u : 0u) << 16);
        rec.t_bits = __builtin_bit_cast(uint32_t, t); rec.cell = cell; rec.esc = n_esc;
        rec.pad[0] = rec.pad[1] = rec.pad[2] = 0u;
        except_list[at] = rec;
    }
};
typedef UnionSinkT<true> UnionSink;

/* the union kernel's lane code for the two tree formats */
struct UnionNarrow {
    typedef VrUnionLane ULane;
    typedef LdsStack<kTraceThreads> Store;
    typedef TopFetch Fetch;
    typedef VrUnionExit Exit;
    typedef SweepNarrow Ray;
    __device__ static __forceinline__ Store store(uint32_t *) { return Store(); }
    __device__ static __forceinline__ Fetch fetch(const uint32_t *nodes, const uint32_t *s_top, uint32_t n_top, uint64_t n_slots) { return TopFetch(nodes, s_top, n_top, n_slots); }
    template <class Done>
    __device__ static __forceinline__ void begin(ULane &U, Store &st, Done &done, const VrViewK &v, const float4 &r, uint32_t b, uint32_t n) { vr_union_begin(U, st, done, v, r.x, r.y, r.z, r.w, b, n); }
    template <class Done>
    __device__ static __forceinline__ void restart(ULane &U, Store &st, Done &done, const VrViewK &v, float t, uint32_t cell, const VrUnionRuns &runs) { vr_union_restart(U, st, done, v, t, cell, runs); }
    /* the lanes of a wave that go down together take the loop without the done bits when none of them has a hole in its alive range */
    template <class Done, class Sink>
    __device__ static __forceinline__ uint32_t descend(ULane &U, const Fetch &f, Store &st, Done &done, Sink &sink)
    {
        if (__ballot(U.holes) == 0ull) return vr_union_descend<false>(U, f, st, done, sink, kUnionDescendTrips);
        return vr_union_descend<true>(U, f, st, done, sink, kUnionDescendTrips);
    }
    __device__ static __forceinline__ uint32_t depth(const ULane &, const Store &st) { return st.depth(); }
    __device__ static __forceinline__ uint32_t cell(const ULane &U) { return U.L.cell; }
    __device__ static __forceinline__ Exit move(ULane &U, const VrViewK &v) { return vr_union_move(U.L, v); }
    __device__ static __forceinline__ bool pop(ULane &U, Store &st, const Exit &x, bool &suspect, VrUnionFrom &from) { return vr_union_pop(U.L, st, kUnionPopTrips, x, suspect, from); }
    __device__ static __forceinline__ VrUnionFrom root(const VrViewK &v) { return VrUnionFrom{ 1u, 0u, v.root_hi }; }
    __device__ static __forceinline__ void leave(ULane &U, Store &st, const Exit &x, bool &suspect, const VrViewK &v) { vr_union_leave(U.L, st, x, suspect, v.root_hi); }
    template <class Done, class Sink>
    __device__ static __forceinline__ void resolve(ULane &U, const VrViewK &v, const Fetch &f, Done &done, Sink &sink, uint32_t depth_k,
                                                   uint32_t cell_before, int a_level, float t_new, uint32_t cell_new, bool in_root,
                                                   const VrUnionFrom &from)
    {
        vr_union_resolve(U, v, f, done, sink, depth_k, cell_before, a_level, t_new, cell_new, in_root, from);
    }
    __device__ static __forceinline__ void after_pop(ULane &U, const Fetch &f) { vr_union_after_pop(U, f); }
};

struct UnionWide {
    typedef VrUnionWideLane ULane;
    typedef LdsPath<kTraceThreads> Store;
    typedef WideFetch Fetch;
    typedef VrUnionWideExit Exit;
    typedef SweepWide Ray;
    __device__ static __forceinline__ Store store(uint32_t *s_dyn) { return Store{ s_dyn + threadIdx.x }; }
    __device__ static __forceinline__ Fetch fetch(const uint32_t *nodes, const uint32_t *, uint32_t, uint64_t n_slots) { return WideFetch{ nodes, n_slots >= 2u ? (uint32_t)(n_slots - 2u) : 0u }; }
    template <class Done>
    __device__ static __forceinline__ void begin(ULane &U, Store &, Done &done, const VrViewK &v, const float4 &r, uint32_t b, uint32_t n) { vr_uw_begin(U, done, v, r.x, r.y, r.z, r.w, b, n); }
    template <class Done>
    __device__ static __forceinline__ void restart(ULane &U, Store &, Done &done, const VrViewK &v, float t, uint32_t cell, const VrUnionRuns &runs) { vr_uw_restart(U, done, v, t, cell, runs); }
    template <class Done, class Sink>
    __device__ static __forceinline__ uint32_t descend(ULane &U, const Fetch &f, Store &st, Done &done, Sink &sink)
    {
        if (__ballot(U.holes) == 0ull) return vr_uw_descend<false>(U, f, st, done, sink, kUnionDescendTrips);
        return vr_uw_descend<true>(U, f, st, done, sink, kUnionDescendTrips);
    }
    __device__ static __forceinline__ uint32_t depth(const ULane &U, const Store &) { return U.L.d; }
    __device__ static __forceinline__ uint32_t cell(const ULane &U) { return U.L.c; }
    __device__ static __forceinline__ Exit move(ULane &U, const VrViewK &v) { return vr_uw_move(U.L, v); }
    __device__ static __forceinline__ bool pop(ULane &U, Store &st, const Exit &x, bool &suspect, VrUnionFrom &from) { return vr_uw_pop(U.L, st, kUnionPopTrips, x, suspect, from); }
    __device__ static __forceinline__ VrUnionFrom root(const VrViewK &v) { return VrUnionFrom{ 0u, 0u, vr_pack10(v.rdx, v.rdy, v.rdz) }; }
    __device__ static __forceinline__ void leave(ULane &U, Store &st, const Exit &x, bool &suspect, const VrViewK &v) { vr_uw_leave(U.L, st, x, suspect, vr_pack10(v.rdx, v.rdy, v.rdz)); }
    template <class Done, class Sink>
    __device__ static __forceinline__ void resolve(ULane &U, const VrViewK &v, const Fetch &f, Done &done, Sink &sink, uint32_t depth_k,
                                                   uint32_t cell_before, int a_level, float t_new, uint32_t cell_new, bool in_root,
                                                   const VrUnionFrom &from)
    {
        vr_uw_resolve(U, v, f, done, sink, depth_k, cell_before, a_level, t_new, cell_new, in_root, from);
    }
    __device__ static __forceinline__ void after_pop(ULane &U, const Fetch &f) { vr_uw_after_pop(U, f); }
};

/* The front of a lane is finished: on with the iso values it has kept, if any (a front of their own from their own state),
 * else the lane is idle and the item's length goes to its tile.  Returns what becomes of the lane's `alive` flag; `popping`
 * is cleared by the caller when the lane goes on (the flags are passed and returned by value: taken by reference they ended
 * up in scratch memory). */
template <class UOPS, class Store, class Done, class Sink>
__device__ __forceinline__ bool union_front_done(typename UOPS::ULane &U, Store &stack, Done &done, const VrViewK &v, Sink &sink,
                                                 uint32_t *length_out, uint32_t n_rounds)
{
    if (sink.pend) {
        VR_WC(12);
        UOPS::restart(U, stack, done, v, sink.pend_t, sink.pend_cell, sink.pend_runs);
        sink.pend_shift();
        return true;
    }
    if (length_out) atomicMax(&length_out[(size_t)(sink.ray >> 6) * kLptStride], n_rounds);
    return false;
}

/* (parameters: what every level of a walk needs first, what only a refill or the end of the launch needs behind it -- the
 * compiler loads neighbouring arguments as one register tuple and spills and reloads such a tuple whole) */
template <class UOPS, bool STATS = true>
__global__ __launch_bounds__(kTraceThreads, kTraceWavesPerSimd) void vr_k_union(
    VrViewK v, const uint32_t *__restrict__ nodes, uint32_t n_top, uint32_t iso_begin, uint32_t iso_count, uint32_t stack_rows,
    uint8_t *__restrict__ rgb, uint64_t frame_pixels, const float4 *__restrict__ ray_dt, const uint32_t *__restrict__ ray_pix,
    const uint32_t *__restrict__ n_active_ptr, uint32_t *__restrict__ head, vr_render_stats *__restrict__ stats,
    VrUnionRecord *__restrict__ except_list, uint32_t except_cap, uint32_t items_per_lane, uint32_t launch_seq,
    uint32_t keep_lengths, const uint32_t *__restrict__ tile_order, uint32_t *__restrict__ tile_cost)
{
    /* dynamic LDS: [path stacks][check sums and counters as in vr_k_sweep][done bits: 8 x kTraceThreads words][top node words] */
    extern __shared__ uint32_t s_dyn[];
    unsigned long long *s_checks = (unsigned long long *)(s_dyn + stack_rows * kTraceThreads);
    uint32_t *s_cnt = (uint32_t *)(s_checks + (iso_count + 1u));
    uint32_t *s_done = s_dyn + stack_rows * kTraceThreads + sweep_stat_words(iso_count);
    uint32_t *s_top = s_done + 8u * kTraceThreads;
    const uint32_t n_k = iso_count + 1u;
    for (uint32_t i = threadIdx.x; i < sweep_stat_words(iso_count); i += kTraceThreads) s_dyn[stack_rows * kTraceThreads + i] = 0u;
    for (uint32_t i = threadIdx.x; i < n_top; i += kTraceThreads) s_top[i] = nodes[i];
    __syncthreads();
    typename UOPS::Store stack = UOPS::store(s_dyn);
    LdsDone<kTraceThreads> done{ s_done + threadIdx.x };
    const typename UOPS::Fetch fetch = UOPS::fetch(nodes, s_top, n_top, v.n_slots);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_active = *n_active_ptr;
    /* work items: (ray, block of iso values), numbered as in vr_k_sweep; one front per item.  Blocks shorten the longest
     * walk and fill the lanes when there are about as many rays as lanes; each costs the states its iso values share
     * with the other blocks of the ray once more. */
    const uint64_t want = (uint64_t)gridDim.x * kTraceThreads * items_per_lane;
    uint32_t nb = (uint32_t)((want + n_active - 1u) / (n_active ? n_active : 1u));
    nb = nb < 1u ? 1u : (nb > VR_SWEEP_MAX_BLOCKS ? VR_SWEEP_MAX_BLOCKS : nb);
    nb = nb > iso_count ? iso_count : nb;
    const uint32_t block_isos = (iso_count + nb - 1u) / nb, n_blocks = (iso_count + block_isos - 1u) / block_isos;
    uint32_t n_items = ((n_active + 63u) & ~63u) * n_blocks;
    unsigned long long *info = (unsigned long long *)(n_active_ptr + 16);
    const unsigned long long tick_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long tick_empty = 0;
    uint32_t n_walks = 0;

    UnionSinkT<STATS> sink;
    sink.s_checks = s_checks; sink.s_cnt = s_cnt; sink.n_k = n_k; sink.iso_begin = iso_begin;
    sink.rgb = rgb; sink.frame_bytes = frame_pixels * 3u; sink.pix = 0; sink.ray = 0;
    sink.except_count = (uint32_t *)n_active_ptr + 32; sink.except_list = except_list; sink.except_cap = except_cap;
    sink.pend = false; sink.pend_t = 0.0f; sink.pend_cell = 0; sink.pend_runs.clear();
    sink.pend2 = false; sink.pend2_t = 0.0f; sink.pend2_cell = 0; sink.pend2_runs.clear();
    sink.pend3 = false; sink.pend3_t = 0.0f; sink.pend3_cell = 0; sink.pend3_runs.clear();
    typename UOPS::ULane U;
    typename UOPS::Exit x = {};
    WaveChunk chunk;
    uint32_t depth_k = 0, cell_before = 0, n_states = 0;
    const bool ordered = tile_order && *(tile_cost - kLptHeaderWords + 1) != 0u;
    /* lengths are kept only for lists the pre-pass would order (one atomic per item adds up on long lists) */
    uint32_t *const length_out = tile_cost && keep_lengths && ((n_active + 63u) >> 6) <= lpt_max_tiles(gridDim.x * kTraceThreads) ? tile_cost : nullptr;
    uint32_t n_rounds = 0; /* what this lane has spent on its item, in half rounds: the item's length, kept per tile for the next launch's order */
    VR_MEASURE_LANE(measure);
    VR_MEASURE_DIMS(n_active, n_blocks, block_isos, iso_begin);
    bool alive = false, exhausted = false, popping = false, suspect = false;
    const uint32_t n_tiles_all = (n_active + 63u) >> 6;
    uint32_t wave_age = 0u; /* rounds since the wave took its tile */
    (void)wave_age;
    /* Two cuts in one launch: the first tiles of an order that is in place and no longer measured -- the longest of the launch --
     * are cut into kUnionFineMul times as many blocks of iso values as the rest.  A launch lasts as long as its longest items, and
     * where those are long because of how many iso values they carry, the finer cut shortens them, while cutting every tile
     * finer costs more in all than it gains (C60 sweep 0.684 -> 0.609 ms, 256^3 stand-in 2.20 -> 2.11; Skull, whose longest item
     * is one ray's walk for a few iso values: 0.846 -> 0.844).  Items [0, fine_items) are the fine ones.  (The measuring build
     * keeps one cut: its item table has one shape.) */
#ifdef VR_MEASURE
    const uint32_t fine_tiles = 0u;
#else
    const uint32_t fine_tiles = ordered && !length_out ? n_tiles_all >> kUnionFineShift : 0u;
#endif
    uint32_t nbf = n_blocks * kUnionFineMul;
    nbf = nbf > iso_count ? iso_count : nbf;
    const uint32_t fine_isos = (iso_count + nbf - 1u) / nbf, n_fine_blocks = (iso_count + fine_isos - 1u) / fine_isos;
    const uint32_t fine_items = fine_tiles * 64u * n_fine_blocks;
    n_items = fine_items + (((n_active + 63u) >> 6) - fine_tiles) * 64u * n_blocks;
#ifndef VR_UNION_NO_SEED
    /* wave k of block b is wave k * blocks + b of the grid: the first tiles of the order go to different CUs and SIMDs */
    chunk.seed((threadIdx.x >> 6) * gridDim.x + blockIdx.x, gridDim.x * (kTraceThreads / 64u), 64u, n_items);
#endif

    for (;;) {
        /* ---- refill idle lanes from the ray queue ---- */
        const unsigned long long idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        bool start = false;            /* this lane begins a front: on ray s_ray, for s_n iso values from s_iso on */
        uint32_t s_ray = 0, s_iso = 1u, s_n = 0;
        VR_WC(15);
        if (!exhausted && n_idle >= kUnionRefillMin) {
            VR_WC(13);
            uint32_t mine;
            const bool got = chunk.take(n_idle, (uint32_t)__popcll(idle & ((1ull << lane) - 1ull)), head, n_items, n_idle, lane, mine);
            if (!alive && got) {
                const bool fine = mine < fine_items;
                const uint32_t per_tile = 64u * (fine ? n_fine_blocks : n_blocks), rel = fine ? mine : mine - fine_items;
                const uint32_t at = (fine ? 0u : fine_tiles) + rel / per_tile, in = rel % per_tile;
                const uint32_t u_isos = fine ? fine_isos : block_isos;
                const uint32_t grp = ordered ? tile_order[at] : at; /* longest tiles of the previous launch first */
                const uint32_t blk = in >> 6, ray = grp * 64u + (in & 63u);
                if (ray < n_active) { /* the last group of 64 may be short */
                    const uint32_t b0 = blk * u_isos, b1 = b0 + u_isos < iso_count ? b0 + u_isos : iso_count;
                    start = true; s_ray = ray; s_iso = iso_begin + b0; s_n = b1 - b0;
                    n_walks++;
                }
            }
#ifndef VR_UNION_NO_PRIO
            wave_age = 0u;
            __builtin_amdgcn_s_setprio(0);
#endif
            exhausted = chunk.exhausted();
            if (exhausted && !tick_empty) tick_empty = __builtin_amdgcn_s_memrealtime();
        }
        if (start) {
            VR_WC(17);
            sink.pix = ray_pix[s_ray];
            sink.ray = s_ray;
            UOPS::begin(U, stack, done, v, ray_dt[s_ray], s_iso, s_n);
            alive = true; popping = false; n_states = 0; n_rounds = 0;
            VR_MEASURE_DO(measure.begin(s_ray * n_blocks + (s_iso - iso_begin) / block_isos, tick_begin));
        }
        if (__ballot(alive) == 0ull) {
            if (exhausted) break;
            continue; /* fewer idle lanes than a refill takes cannot happen with nobody alive, but stay safe */
        }
        VR_MEASURE_DO(measure.round(alive, popping, tick_begin));
        /* The launch lasts as long as its longest items: a wave that has been at its tile for 16 rounds issues ahead of the younger
         * ones on its SIMD, from 48 rounds on ahead of everybody (a wave holds one tile at a time; the priority goes back to 0 with
         * the next tile).  Skull / C60 / 256^3 stand-in sweep, kernel ms: no priorities (-DVR_UNION_NO_PRIO) 0.841 / 0.616 / 2.115;
         * these 0.848 / 0.622 / 1.954; by the tile's place in the order instead (first 1/32: 3, first 1/8: 1) 0.845 / 0.612 / 2.105;
         * both: 0.865 / 0.655 / 2.00. */
#ifndef VR_UNION_NO_PRIO
        wave_age++;
        if (wave_age == 16u) __builtin_amdgcn_s_setprio(2);
        else if (wave_age == 48u) __builtin_amdgcn_s_setprio(3);
#endif
        if (alive) {
            /* the item's length for the next launch's order, in half rounds: a round of a wave in which some lane has a hole in its
             * open iso values goes through the done-bit code and takes about one and a half times as long (Skull sweep: tiles ordered
             * by plain rounds 0.985 ms, by these 0.908; C60 and the 256^3 stand-in, where nearly every wave has holes: the same) */
            n_rounds += 2u + (__ballot(U.holes) != 0ull ? 1u : 0u);
            uint32_t go = VR_GO_MORE;
            if (!popping) go = UOPS::descend(U, fetch, stack, done, sink);
            if (go == VR_GO_EXCLUDED) {
                VR_WC(19);
                if (U.alive_lo > U.alive_hi) { /* every iso value of the front has hit */
                    alive = union_front_done<UOPS>(U, stack, done, v, sink, length_out, n_rounds);
                    popping = false;
                    if (!alive) VR_MEASURE_DO(measure.end());
                } else {
                    depth_k = UOPS::depth(U, stack); cell_before = UOPS::cell(U);
                    x = UOPS::move(U, v);
                    suspect = false; /* set where a box that is left does not share the front's exit plane */
                    if (++n_states > kUnionMaxStates) { /* safety valve: the front is given up and the call rendered again */
                        sink.except_count[1] = 1u;
                        sink.pend = false; sink.pend2 = false; sink.pend3 = false;
                        alive = false;
                        if (length_out) atomicMax(&length_out[(size_t)(sink.ray >> 6) * kLptStride], n_rounds);
                        VR_MEASURE_DO(measure.end());
                    } else if (!x.in_root) {
                        /* the front has left the volume: what is not done is a miss (src/raytrace.zig:66-68) */
                        const float t_new = U.L.t;
                        UOPS::leave(U, stack, x, suspect, v);
                        if (suspect) UOPS::resolve(U, v, fetch, done, sink, depth_k, cell_before, -1, t_new, 0u, false, UOPS::root(v));
                        else vr_union_credit_esc(done, U, sink, 0u, 255u, x.esc);
                        if (U.alive_lo <= U.alive_hi)
                            vr_union_runs(done, U.alive_lo, U.alive_hi, [&](uint32_t r0, uint32_t r1) { sink.miss(r0, r1); }, U.holes);
                        alive = union_front_done<UOPS>(U, stack, done, v, sink, length_out, n_rounds);
                        popping = false;
                        if (!alive) VR_MEASURE_DO(measure.end());
                    } else {
                        popping = true;
                    }
                }
            }
            if (popping && alive) {
                VR_WC(18);
                VrUnionFrom from = UOPS::root(v);
                if (UOPS::pop(U, stack, x, suspect, from)) {
                    popping = false;
                    if (suspect)
                        UOPS::resolve(U, v, fetch, done, sink, depth_k, cell_before, (int)UOPS::depth(U, stack) - 1, U.L.t, UOPS::cell(U), true, from);
                    UOPS::after_pop(U, fetch);
                    if (!suspect) vr_union_credit_esc(done, U, sink, U.prev_mn, U.prev_mx, x.esc);
                    if (U.alive_lo > U.alive_hi) {
                        alive = union_front_done<UOPS>(U, stack, done, v, sink, length_out, n_rounds);
                        if (!alive) VR_MEASURE_DO(measure.end());
                    }
                }
            }
        }
        /* ---- a run that has left a lane's front goes to an idle lane of the wave, if there is one: a front of its own,
         * walked at the same time instead of after the ray's front (which is often the longest of the wave) ---- */
        for (;;) {
            const unsigned long long givers = __ballot(alive && sink.pend), takers = __ballot(!alive);
            if (givers == 0ull || takers == 0ull) break;
            VR_WC(14);
            const uint32_t giver = (uint32_t)__ffsll((long long)givers) - 1u, taker = (uint32_t)__ffsll((long long)takers) - 1u;
            const float g_t = lane_f(sink.pend_t, giver);
            const uint32_t g_cell = (uint32_t)__builtin_amdgcn_readlane((int)sink.pend_cell, giver);
            VrUnionRuns g_runs;
            g_runs.n = (uint32_t)__builtin_amdgcn_readlane((int)sink.pend_runs.n, giver);
            g_runs.w01 = (uint32_t)__builtin_amdgcn_readlane((int)sink.pend_runs.w01, giver);
            g_runs.w23 = (uint32_t)__builtin_amdgcn_readlane((int)sink.pend_runs.w23, giver);
            const uint32_t g_pix = (uint32_t)__builtin_amdgcn_readlane((int)sink.pix, giver);
            const uint32_t g_ray = (uint32_t)__builtin_amdgcn_readlane((int)sink.ray, giver);
            VR_MEASURE_DO(measure.hand_over(giver, lane == taker));
            UOPS::Ray::adopt(U.L, giver, lane == taker);
            if (lane == giver) sink.pend_shift();
            if (lane == taker) {
                sink.pix = g_pix; sink.ray = g_ray;
                UOPS::restart(U, stack, done, v, g_t, g_cell, g_runs);
                alive = true; popping = false; n_states = 0; n_rounds = 0;
            }
        }
    }

    {
        const uint32_t w_walks = wave_sum(n_walks);
        if (lane == 0) {
            atomicAdd(&s_cnt[kSweepCounters * n_k], w_walks);
            if ((threadIdx.x >> 6) == 0u) {
                atomicMin(&info[2], tick_begin);
                if (tick_empty) atomicMin(&info[3], tick_empty);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&info[0], (unsigned long long)s_cnt[kSweepCounters * n_k]);
        if (blockIdx.x == 0 && tile_cost) *(tile_cost - kLptHeaderWords) = (n_active + 63u) >> 6;
        if (blockIdx.x == 0) { /* items of the launch: a ray of a fine tile has n_fine_blocks of them */
            atomicAdd(&info[1], (unsigned long long)n_active * n_blocks + (unsigned long long)fine_tiles * 64u * (n_fine_blocks - n_blocks));
            info[5] = launch_seq; /* (the pre-pass has left info[1] at zero; the correction below is an atomic as well) */
        }
        atomicMax(&info[4], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    if (blockIdx.x == 0 && (n_active & 63u) != 0u && n_fine_blocks != n_blocks) {
        /* the last tile of the list is short: if it is one of the fine ones, the rays it lacks have no items */
        for (uint32_t i = threadIdx.x; i < fine_tiles; i += kTraceThreads)
            if (tile_order[i] == n_tiles_all - 1u)
                atomicAdd(&info[1], 0ull - (unsigned long long)(64u - (n_active & 63u)) * (n_fine_blocks - n_blocks));
    }
    if (threadIdx.x < kSweepCounters) {
        uint32_t *c = s_cnt + threadIdx.x * n_k;
        uint32_t run = 0;
        for (uint32_t k = 0; k < iso_count; k++) { run += c[k]; c[k] = run; }
    } else if (threadIdx.x == kSweepCounters) {
        unsigned long long run = 0;
        for (uint32_t k = 0; k < iso_count; k++) { run += s_checks[k]; s_checks[k] = run; }
    }
    __syncthreads();
    if (stats) {
        for (uint32_t k = threadIdx.x; k < iso_count; k += kTraceThreads) {
            vr_render_stats *s = stats + k;
            add_stat(&s->nodes_checked, s_checks[k] & ((1ull << kParkMoveShift) - 1ull));
            add_stat(&s->hit_count, s_cnt[k]);
            add_stat(&s->miss_count, s_cnt[n_k + k]);
            add_stat(&s->moves, (s_checks[k] >> kParkMoveShift) + s_cnt[2u * n_k + k]);
            add_stat(&s->escape_help_events, s_cnt[3u * n_k + k]);
            add_stat(&s->errors, s_cnt[4u * n_k + k]);
        }
    }
}


/* The union kernel asked for the call to be rendered again (kUnionRedoWord): every block of the fix kernel's grid clears its
 * share of the frames, and the first one puts the per-iso counters and the queue heads back to where the pre-pass left them. */
__device__ void union_redo_clear(const uint32_t *__restrict__ n_active_ptr, uint32_t *__restrict__ heads, uint32_t iso_count,
                                 uint64_t frame_pixels, uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats)
{
    const uint64_t bytes = frame_pixels * 3u * iso_count;
    const uint64_t head = (16u - ((uintptr_t)rgb & 15u)) & 15u, lead = head < bytes ? head : bytes;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_threads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < lead; i += n_threads) rgb[i] = 0u;
    const uint64_t n16 = (bytes - lead) / 16u;
    uint4 *mid = (uint4 *)(rgb + lead);
    for (uint64_t i = tid; i < n16; i += n_threads) mid[i] = make_uint4(0u, 0u, 0u, 0u);
    for (uint64_t i = lead + n16 * 16u + tid; i < bytes; i += n_threads) rgb[i] = 0u;
    if (blockIdx.x == 0)
        for (uint32_t k = threadIdx.x; k < iso_count; k += blockDim.x) {
            heads[k] = 0u;
            if (stats) {
                vr_render_stats z = { 0, 0, 0, frame_pixels - *n_active_ptr, 0, 0, 0 };
                stats[k] = z;
            }
        }
}

/* The iso values that left the union walk: each is traced on its own from its own state (the checks and the MOVE of
 * the round that parked it are counted already, its escape helps of that MOVE come with the record). */
__global__ __launch_bounds__(64) void vr_k_union_fix(VrViewK v, const uint32_t *__restrict__ nodes, const float4 *__restrict__ ray_dt,
                                                     const uint32_t *__restrict__ ray_pix, const uint32_t *__restrict__ n_active_ptr,
                                                     const VrUnionRecord *__restrict__ list, uint32_t cap, uint32_t iso_begin,
                                                     uint64_t frame_pixels, uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats,
                                                     uint32_t stack_rows, uint32_t n_top, uint32_t *__restrict__ heads, uint32_t iso_count)
{
    if (n_active_ptr[kUnionRedoWord]) { union_redo_clear(n_active_ptr, heads, iso_count, frame_pixels, rgb, stats); return; }
    uint32_t count = n_active_ptr[32];
    count = count < cap ? count : cap;
    if (blockIdx.x >= count) return; /* nearly every block of nearly every launch */
    LdsStack<64u> st; /* dynamic LDS: [levels + 1][64], then the top levels of the tree (these walks start at the root) */
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_top = s_dyn + stack_rows * 64u;
    for (uint32_t i = threadIdx.x; i < n_top; i += 64u) s_top[i] = nodes[i];
    __syncthreads();
    const TopFetch fetch(nodes, s_top, n_top, v.n_slots);
    /* one wave per record, a lane per iso value of its run */
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const VrUnionRecord rec = list[i];
        const float4 r = ray_dt[rec.ray];
        const uint32_t pix = ray_pix[rec.ray];
        const uint32_t r0 = rec.run & 0xffu, r1 = (rec.run >> 8) & 0xffu;
        const bool in_root = (rec.run >> 16) & 1u;
        VrLane A;
        vr_lane_begin(A, st, v, r.x, r.y, r.z, r.w);
        for (uint32_t iso = r0 + threadIdx.x; iso <= r1; iso += 64u) {
            uint32_t moves = 0, esc = rec.esc, err = 0;
            bool hit = false;
            A.checked = 0;
            if (in_root) {
                vr_lane_restart(A, st, v, __builtin_bit_cast(float, rec.t_bits), rec.cell);
                for (;;) {
                    const uint32_t go = vr_lane_descend(A, iso, fetch, st, 64u);
                    if (go == VR_GO_MORE) continue;
                    if (go == VR_GO_HIT) { hit = true; break; }
                    moves++;
                    const bool in_root_now = vr_lane_move(A, v, esc);
                    if (A.checked > VR_MAX_CHECKED) { err = 1; break; } /* given up behind the MOVE, as in vr_k_trace */
                    if (!in_root_now) break;
                    while (!vr_lane_pop(A, st, 64u)) {}
                }
            }
            if (stats) {
                vr_render_stats *s = stats + (iso - iso_begin);
                add_stat(&s->nodes_checked, A.checked);
                add_stat(hit ? &s->hit_count : &s->miss_count, 1ull);
                add_stat(&s->moves, moves);
                add_stat(&s->escape_help_events, esc);
                add_stat(&s->errors, err);
            }
            if (hit) {
                const uint8_t g = vr_shade(A.t);
                uint8_t *p = rgb + ((uint64_t)(iso - iso_begin) * frame_pixels + pix) * 3u;
                p[0] = g; p[1] = g; p[2] = g;
            }
        }
    }
    /* the sweep's report ends where its last piece of work ends: the cut of the iso range is judged by that */
    if (threadIdx.x == 0)
        atomicMax((unsigned long long *)(n_active_ptr + 16) + 4, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

/* vr_k_union_fix for wide trees */
__global__ __launch_bounds__(64) void vr_k_union_fix_wide(VrViewK v, const uint32_t *__restrict__ nodes, const float4 *__restrict__ ray_dt,
                                                          const uint32_t *__restrict__ ray_pix, const uint32_t *__restrict__ n_active_ptr,
                                                          const VrUnionRecord *__restrict__ list, uint32_t cap, uint32_t iso_begin,
                                                          uint64_t frame_pixels, uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats,
                                                          uint32_t *__restrict__ heads, uint32_t iso_count)
{
    __shared__ uint32_t s_path[VR_WIDE_MAX_DEPTH * 64u];
    if (n_active_ptr[kUnionRedoWord]) { union_redo_clear(n_active_ptr, heads, iso_count, frame_pixels, rgb, stats); return; }
    uint32_t count = n_active_ptr[32];
    count = count < cap ? count : cap;
    LdsPath<64u> path{ s_path + threadIdx.x };
    const WideFetch fetch{ nodes, 0u };
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const VrUnionRecord rec = list[i];
        const float4 r = ray_dt[rec.ray];
        const uint32_t pix = ray_pix[rec.ray];
        const uint32_t r0 = rec.run & 0xffu, r1 = (rec.run >> 8) & 0xffu;
        const bool in_root = (rec.run >> 16) & 1u;
        VrWideLane A;
        vr_wide_begin(A, v, r.x, r.y, r.z, r.w);
        for (uint32_t iso = r0 + threadIdx.x; iso <= r1; iso += 64u) {
            uint32_t moves = 0, esc = rec.esc, err = 0;
            bool hit = false;
            A.checked = 0;
            if (in_root) {
                vr_wide_restart(A, v, __builtin_bit_cast(float, rec.t_bits), rec.cell);
                for (;;) {
                    const uint32_t go = vr_wide_descend(A, v, iso, fetch, path, VR_WIDE_MAX_DEPTH + 1u);
                    if (go == VR_GO_HIT) { hit = true; break; }
                    moves++;
                    const bool in_root_now = vr_wide_move(A, v, esc);
                    if (A.checked > VR_MAX_CHECKED) { err = 1; break; } /* given up behind the MOVE, as in every kernel */
                    if (!in_root_now) break;
                    while (!vr_wide_pop(A, path, 64u)) {}
                }
            }
            if (stats) {
                vr_render_stats *s = stats + (iso - iso_begin);
                add_stat(&s->nodes_checked, A.checked);
                add_stat(hit ? &s->hit_count : &s->miss_count, 1ull);
                add_stat(&s->moves, moves);
                add_stat(&s->escape_help_events, esc);
                add_stat(&s->errors, err);
            }
            if (hit) {
                const uint8_t g = vr_shade(A.t);
                uint8_t *p = rgb + ((uint64_t)(iso - iso_begin) * frame_pixels + pix) * 3u;
                p[0] = g; p[1] = g; p[2] = g;
            }
        }
    }
    /* the sweep's report ends where its last piece of work ends: the cut of the iso range is judged by that */
    if (blockIdx.x < count && threadIdx.x == 0)
        atomicMax((unsigned long long *)(n_active_ptr + 16) + 4, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

/* ------------------------------------------------------------------ wide volumes ---- */

/* Volumes beyond the reference's u8 coordinates (up to 1024 voxels per axis; SPACESIZE is u8 in the reference,
 * src/typedef.zig:1): the kernels below walk the wide node array as built, or as uploaded and validated (lane
 * code in vr_trace_wide.h), with the lane's path -- one word per level -- in LDS as [level][thread] words.  A wide tree
 * is up to 2^31 four-byte words (8 GiB); node fetches use 64-bit addresses. */
/* one thread per pixel and iso value: a 16 x 16 pixel block, each wave an 8 x 8 tile */
__global__ __launch_bounds__(256) void vr_k_render_wide(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t iso_begin,
                                                        uint32_t row_begin, uint32_t n_rows, uint8_t *__restrict__ rgb,
                                                        vr_render_stats *__restrict__ stats, VrTapsK taps)
{
    __shared__ uint32_t s_path[VR_WIDE_MAX_DEPTH * 256u];

    const uint32_t tiles_x = (v.width + kBlockW - 1) / kBlockW;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t xi = bx * kBlockW + (wave & 1u) * kTileW + (lane & 7u);
    const uint32_t yr = by * kBlockH + (wave >> 1) * kTileH + (lane >> 3); /* row relative to row_begin */
    const uint32_t iso = iso_begin + blockIdx.y;
    const bool valid = xi < v.width && yr < n_rows;

    VrHit h;
    h.t = 0.0f; h.flag = VR_FLAG_OOB; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (valid) {
        const VrRay ray = vr_make_ray(v, xi, row_begin + vr_band_row(v, yr));
        LdsPath<256u> path{ s_path + threadIdx.x };
        h = vr_trace_wide(v, ray, iso, WideFetch{ nodes, 0u }, path);
        const uint64_t pix = ((uint64_t)blockIdx.y * n_rows + yr) * v.width + xi;
        const uint8_t g = h.flag == VR_FLAG_HIT ? vr_shade(h.t) : (uint8_t)0;
        rgb[pix * 3 + 0] = g;
        rgb[pix * 3 + 1] = g;
        rgb[pix * 3 + 2] = g;
        if (taps.cell_flag) taps.cell_flag[pix] = h.cell | (h.flag << VR_WIDE_FLAG_SHIFT);
        if (taps.t) taps.t[pix] = h.flag == VR_FLAG_OOB ? 0.0f : h.t;
        if (taps.checked) taps.checked[pix] = h.checked;
        if (taps.node) taps.node[pix] = h.node;
    }

    if (stats) {
        const uint32_t c_checked = wave_sum(valid ? h.checked : 0u);
        const uint32_t c_hit = wave_sum(valid && h.flag == VR_FLAG_HIT ? 1u : 0u);
        const uint32_t c_miss = wave_sum(valid && h.flag == VR_FLAG_MISS ? 1u : 0u);
        const uint32_t c_oob = wave_sum(valid && h.flag == VR_FLAG_OOB ? 1u : 0u);
        const uint32_t c_esc = wave_sum(valid ? h.escapes : 0u);
        const uint32_t c_moves = wave_sum(valid ? h.moves : 0u);
        const uint32_t c_err = wave_sum(valid ? h.error : 0u);
        if (lane == 0) {
            vr_render_stats *s = stats + blockIdx.y;
            add_stat(&s->nodes_checked, c_checked);
            add_stat(&s->hit_count, c_hit);
            add_stat(&s->miss_count, c_miss);
            add_stat(&s->oob_count, c_oob);
            add_stat(&s->escape_help_events, c_esc);
            add_stat(&s->moves, c_moves);
            add_stat(&s->errors, c_err);
        }
    }
}

#ifndef VR_WIDE_DESCEND_TRIPS
#define VR_WIDE_DESCEND_TRIPS 6
#endif
#ifndef VR_WIDE_POP_TRIPS
#define VR_WIDE_POP_TRIPS 64 /* levels a lane may pop per round; bounds of 3..8 measured no faster than none */
#endif
#ifndef VR_WIDE_THREADS
#define VR_WIDE_THREADS 1024
#endif
#ifndef VR_WIDE_BLOCKS_PER_CU
#define VR_WIDE_BLOCKS_PER_CU 1
#endif
/* block shape of the wide persistent kernel: the path store costs 120 B of LDS per lane, so a CU holds at most
 * 1365 lanes (21 waves) of it */
constexpr uint32_t kWideThreads = VR_WIDE_THREADS, kWideBlocksPerCu = VR_WIDE_BLOCKS_PER_CU;
/* The persistent kernel of the pipeline for wide trees: the compacted ray list, one queue head per iso value,
 * waves that refill idle lanes (as vr_k_trace), the wide lane code.  A round is a bounded descend, the MOVE for the
 * lanes whose node excluded the iso value, and a bounded pop (LDS only); a lane that runs out of a budget carries
 * on in the next round, so the wave does not run at the length of its slowest lane.  Dynamic LDS: the path store. */
template <bool TAPS>
__global__ __launch_bounds__(kWideThreads) void vr_k_trace_wide(
    VrViewK v, const uint32_t *__restrict__ nodes, const float4 *__restrict__ ray_dt, const uint32_t *__restrict__ ray_pix,
    const uint32_t *__restrict__ n_active_ptr, uint32_t *__restrict__ heads, uint32_t iso_begin, uint32_t iso_count,
    uint64_t frame_pixels, uint8_t *__restrict__ rgb, vr_render_stats *__restrict__ stats, uint32_t block_counters,
    VrTapsK taps)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ unsigned long long s_acc_mem[kSlotIsoMax * 6u]; /* the block's counter sums of a short iso range (block_counters) */
    unsigned long long *const s_acc = block_counters ? s_acc_mem : nullptr;
    if (threadIdx.x < kSlotIsoMax * 6u) s_acc_mem[threadIdx.x] = 0ull;
    __syncthreads();
    LdsPath<kWideThreads> path{ s_dyn + threadIdx.x };
    const WideFetch fetch{ nodes, (uint32_t)(v.n_slots - 2u) }; /* a wide tree has far more than two nodes */

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_active = *n_active_ptr;
    const uint32_t waves_per_block = kWideThreads / 64u;
    const uint32_t n_waves = gridDim.x * waves_per_block, gw = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const uint32_t k0 = (uint32_t)(((uint64_t)gw * iso_count) / n_waves);
    const uint32_t chunk_rays = trace_chunk(n_active, n_waves);

    for (uint32_t visit = 0; visit < iso_count; visit++) {
        uint32_t k = k0 + visit;
        if (k >= iso_count) k -= iso_count;
        const uint32_t iso = iso_begin + k;
        uint8_t *frame = rgb + (uint64_t)k * frame_pixels * 3u;
        const uint64_t tap_base = (uint64_t)k * frame_pixels;

        VrWideLane L;
        WaveChunk chunk;
        trace_seed(chunk, visit, k, iso_count, gw, n_waves, chunk_rays, n_active);
        uint32_t pix = 0;
        bool alive = false, exhausted = false, popping = false;
        unsigned long long a_checked = 0;
        uint32_t a_hit = 0, a_miss = 0, a_moves = 0, a_esc = 0, a_err = 0;

        for (;;) {
            const unsigned long long idle = __ballot(!alive);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (!exhausted && n_idle >= kRefillMin) {
                uint32_t mine;
                const bool got = chunk.take(n_idle, (uint32_t)__popcll(idle & ((1ull << lane) - 1ull)), &heads[k], n_active, chunk_rays,
                                            lane, mine);
                if (!alive && got) {
                    const float4 r = ray_dt[mine];
                    pix = ray_pix[mine];
                    vr_wide_begin(L, v, r.x, r.y, r.z, r.w);
                    alive = true;
                    popping = false;
                }
                exhausted = chunk.exhausted();
            }
            if (__ballot(alive) == 0ull) break; /* only possible once the queue is exhausted */

            if (alive) {
                uint32_t go = VR_GO_MORE;
#ifdef VR_WIDE_NO_AHEAD
                if (!popping) go = vr_wide_descend(L, v, iso, fetch, path, VR_WIDE_DESCEND_TRIPS);
#else
                if (!popping) { VrNoIsoTrack none; go = vr_wide_descend_ahead(L, v, iso, fetch, path, VR_WIDE_DESCEND_TRIPS, none); }
#endif
                if (go == VR_GO_HIT) { /* shade and store (src/raytrace.zig:56-61) */
                    const uint8_t g = vr_shade(L.t);
                    frame[(uint64_t)pix * 3u + 0u] = g;
                    frame[(uint64_t)pix * 3u + 1u] = g;
                    frame[(uint64_t)pix * 3u + 2u] = g;
                    a_hit++;
                    alive = false;
                    if (TAPS) {
                        if (taps.cell_flag) taps.cell_flag[tap_base + pix] = L.c | (VR_FLAG_HIT << VR_WIDE_FLAG_SHIFT);
                        if (taps.node) taps.node[tap_base + pix] = L.idx;
                    }
                } else if (go == VR_GO_EXCLUDED) {
                    a_moves++;
                    bool on = vr_wide_move(L, v, a_esc);
                    if (L.checked > VR_MAX_CHECKED) { on = false; a_err++; } /* the bound on a ray's checks: given up behind the MOVE, as in vr_k_trace */
                    popping = on;
                    if (!on) { /* left the volume: a miss, the pixel stays black (src/raytrace.zig:66-68) */
                        a_miss++;
                        alive = false;
                        if (TAPS) {
                            if (taps.cell_flag) taps.cell_flag[tap_base + pix] = VR_FLAG_MISS << VR_WIDE_FLAG_SHIFT;
                            if (taps.node) taps.node[tap_base + pix] = 0u;
                        }
                    }
                }
                /* pop to the lowest ancestor that holds the new point (bounded: the rest next round) */
                if (alive && popping) popping = !vr_wide_pop(L, path, VR_WIDE_POP_TRIPS);
                if (!alive) {
                    a_checked += L.checked;
                    if (TAPS) {
                        if (taps.t) taps.t[tap_base + pix] = L.t;
                        if (taps.checked) taps.checked[tap_base + pix] = L.checked;
                    }
                }
            }
        }

        if (stats) wave_stats_out(stats, s_acc, k, lane, a_checked, a_hit, a_miss, a_esc, a_moves, a_err);
    }
    if (stats && s_acc) block_stats_out(stats, s_acc, iso_count);
}

#ifdef VR_MEASURE
/* Measuring build: ONE ray of one iso value walked by ONE lane of one wave with the lane code of the pipeline kernel (resume
 * traversal, top tree levels in LDS, the same flat node fetch), every step bracketed by cycle stamps with all memory operations
 * waited for at each stamp -- the latency chain of a walk with nothing else on the chip.  out[]: per kind of step (0 node fetch
 * from the LDS copy, 1 node fetch from global memory, 2 rest of a descend level: range test, child choice, box update, push,
 * 3 MOVE with its escape loop, 4 one popped level, 5 entering the sibling at the end of a pop, 6 loop bookkeeping between the
 * steps) the number of steps and the sum of their cycles; out[20] the whole walk, out[21] checks, out[22] MOVEs, out[23] flag;
 * lat[]: the first 4096 node fetches as cycles | from_global << 20 | level << 24. */
__global__ __launch_bounds__(64) void vr_k_ray_timeline(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t n_top, uint32_t stack_rows,
                                                        uint32_t xi, uint32_t yi, uint32_t iso, unsigned long long *__restrict__ out,
                                                        uint32_t *__restrict__ lat)
{
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_top = s_dyn + stack_rows * 64u;
    uint32_t *s_lat = s_top + n_top; /* 4096 words: the table is kept in LDS during the walk and copied out behind it */
    for (uint32_t i = threadIdx.x; i < n_top; i += 64u) s_top[i] = nodes[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    LdsStack<64u> st;
    const TopFetch fetch(nodes, s_top, n_top, v.n_slots);
    const VrRay ray = vr_make_ray(v, xi, yi);
    /* scalars, not an array: an array indexed at run time would live in scratch memory and its traffic in the stamps */
    unsigned long long c_lds = 0, c_glb = 0, c_rest = 0, c_move = 0, c_pop = 0, c_sib = 0, c_book = 0;
    uint32_t n_lds = 0, n_glb = 0, n_rest = 0, n_move = 0, n_pop = 0, n_sib = 0;
    uint32_t n_lat = 0, moves = 0, esc = 0, flag = ray.oob ? 2u : 0u;
#define VR_STAMP(var) __builtin_amdgcn_s_waitcnt(0); const unsigned long long var = __builtin_amdgcn_s_memtime()
    VrLane L;
    VR_STAMP(t_begin);
    if (!ray.oob) {
        vr_lane_begin(L, st, v, ray.dx, ray.dy, ray.dz, ray.t);
        unsigned long long t_prev = __builtin_amdgcn_s_memtime();
        for (;;) {
            /* ---- descend, level by level (vr_lane_descend, one trip at a time) ---- */
            uint32_t go = VR_GO_MORE;
            while (go == VR_GO_MORE) {
                VR_STAMP(t0);
                c_book += t0 - t_prev;
                const uint32_t n = fetch(L.g);
                asm volatile("" :: "v"(n));
                VR_STAMP(t1);
                L.checked++;
                if (!vr_node_holds(n, iso)) go = VR_GO_EXCLUDED;
                else if (n & VR_N_LEAF) go = VR_GO_HIT;
                else {
                    const uint32_t sh = (n >> 24) & 31u, part = (n >> 16) & 0xffu;
                    const bool right = ((L.cell >> sh) & 0xffu) >= part;
                    uint32_t side = right ? L.lo : L.hi;
                    st.push(side);
                    side = (side & ~(0xffu << sh)) | (part << sh);
                    L.lo = right ? side : L.lo;
                    L.hi = right ? L.hi : side;
                    L.g = vr_twice_plus(L.g, right);
                }
                VR_STAMP(t2);
                c_rest += t2 - t1; n_rest++;
                /* bookkeeping of the measurement itself (counted as such): the fetch goes to the table */
                const bool from_global = (go == VR_GO_MORE ? (L.g >> 1) : L.g) >= n_top;
                if (from_global) { c_glb += t1 - t0; n_glb++; } else { c_lds += t1 - t0; n_lds++; }
                if (n_lat < 4096u) s_lat[n_lat++] = (uint32_t)(t1 - t0 < 0xfffffu ? t1 - t0 : 0xfffffu) | ((from_global ? 1u : 0u) << 20) | ((st.depth() - (go == VR_GO_MORE ? 1u : 0u)) << 24);
                t_prev = t2;
            }
            if (go == VR_GO_HIT) { flag = 1u; break; }
            moves++;
            VR_STAMP(t3);
            c_book += t3 - t_prev;
            const bool in_root = vr_lane_move(L, v, esc);
            VR_STAMP(t4);
            c_move += t4 - t3; n_move++;
            t_prev = t4;
            if (!in_root || moves > (1u << 20)) break;
            bool entered = false;
            while (!entered) {
                VR_STAMP(t5);
                c_book += t5 - t_prev;
                const uint32_t d0 = st.depth();
                entered = vr_lane_pop(L, st, 1u);
                VR_STAMP(t6);
                if (entered && st.depth() == d0) { c_sib += t6 - t5; n_sib++; } /* found at once: only the step into the sibling */
                else { c_pop += t6 - t5; n_pop++; }
                t_prev = t6;
            }
        }
    }
    VR_STAMP(t_end);
    const uint32_t stamped_checked = ray.oob ? 0u : L.checked;
    /* the same walk once more with nothing in between: the lane code as the kernels call it (bounded descend / pop phases) */
    unsigned long long bare = 0;
    uint32_t bare_checked = 0;
    if (!ray.oob) {
        uint32_t esc2 = 0, mv2 = 0;
        VR_STAMP(b0);
        vr_lane_begin(L, st, v, ray.dx, ray.dy, ray.dz, ray.t);
        for (;;) {
            const uint32_t go = vr_lane_descend(L, iso, fetch, st, kDescendTrips);
            if (go == VR_GO_MORE) continue;
            if (go == VR_GO_HIT) break;
            if (++mv2 > (1u << 20) || !vr_lane_move(L, v, esc2)) break;
            while (!vr_lane_pop(L, st, kPopTrips)) {}
        }
        VR_STAMP(b1);
        bare = b1 - b0; bare_checked = L.checked;
    }
    /* what a stamp itself costs: the distance of two with nothing in between */
    unsigned long long stamp_cost = 0;
    for (int i = 0; i < 16; i++) {
        VR_STAMP(s0);
        VR_STAMP(s1);
        stamp_cost += s1 - s0;
    }
#undef VR_STAMP
    out[14] = bare; out[15] = bare_checked; out[16] = stamp_cost;
    out[0] = c_lds; out[1] = n_lds; out[2] = c_glb; out[3] = n_glb; out[4] = c_rest; out[5] = n_rest; out[6] = c_move; out[7] = n_move;
    out[8] = c_pop; out[9] = n_pop; out[10] = c_sib; out[11] = n_sib; out[12] = c_book; out[13] = n_rest;
    out[20] = t_end - t_begin; out[21] = stamped_checked; out[22] = moves; out[23] = flag; out[24] = n_lat;
    for (uint32_t i = 0; i < n_lat; i++) lat[i] = s_lat[i];
}
#endif

/* The frames of a call set to zero with stores that do not stay in the L2 (16 bytes per lane and turn, the odd bytes at either
 * end one by one).  hipMemsetAsync writes through the L2 like any store and leaves it full of frame lines: the traversal kernel
 * behind it then finds none of the tree there -- Skull sweep 0.857 ms behind the memset, 0.800 behind this kernel, which also
 * takes 0.11 ms for the 805 MB instead of 0.14 (step 1.015 -> 0.94 ms; C60 0.797 -> 0.752). */
typedef uint32_t vr_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void vr_k_clear_frames(uint8_t *__restrict__ rgb, uint64_t bytes)
{
    const uint64_t head = (16u - ((uintptr_t)rgb & 15u)) & 15u, lead = head < bytes ? head : bytes;
    const uint64_t n16 = (bytes - lead) / 16u;
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x, n_threads = (uint64_t)gridDim.x * 256u;
    vr_u32x4 *mid = (vr_u32x4 *)(rgb + lead);
    const vr_u32x4 z = { 0u, 0u, 0u, 0u };
    for (uint64_t i = tid; i < n16; i += n_threads) __builtin_nontemporal_store(z, mid + i);
    if (tid < lead) rgb[tid] = 0u;
    const uint64_t tail = lead + n16 * 16u;
    if (tail + tid < bytes) rgb[tail + tid] = 0u; /* fewer than 16 bytes */
}

static hipError_t clear_frames(hipStream_t stream, uint8_t *d_rgb, uint64_t bytes)
{
    static const bool use_memset = getenv("VR_CLEAR_KERNEL") && atoi(getenv("VR_CLEAR_KERNEL")) == 0; /* for A/B runs */
    if (use_memset || bytes < 4096u) return hipMemsetAsync(d_rgb, 0, bytes, stream);
    /* a block per 16 KiB, at most 32 768 blocks (8 192: 0.965 ms per Skull sweep step, 16 384: 0.949, 32 768 ... 131 072: 0.941) */
    uint64_t blocks = bytes / 16384u;
    blocks = blocks < 64u ? 64u : (blocks > 32768u ? 32768u : blocks);
    hipLaunchKernelGGL(vr_k_clear_frames, dim3((uint32_t)blocks), dim3(256), 0, stream, d_rgb, bytes);
    return hipGetLastError();
}

/* workspace layout of the pipeline and its pre-pass (entry test once per call, dense ray list in tile order) */
struct Prepass {
    float4 *ray_dt;
    uint32_t *ray_pix, *n_active, *heads;
};

hipError_t launch_prepass(hipStream_t stream, const VrViewK &view, uint32_t iso_count, uint32_t row_begin, uint32_t n_rows,
                          uint32_t center_out, uint8_t *d_rgb, vr_render_stats *d_stats, const VrTapsK &taps,
                          uint32_t oob_word, void *workspace, Prepass &pp, uint32_t *tile_cost = nullptr, uint32_t *tile_order = nullptr,
                          uint32_t grid_lanes = 0, uint32_t *chord_order = nullptr, bool list_kept = false)
{
    const size_t n_pix = (size_t)view.width * n_rows;
    const uint32_t n_blocks = padded_tiles(view.width, n_rows) / (kSuper * kSuper);
    char *w = (char *)workspace;
    pp.ray_dt = (float4 *)w;
    w += (n_pix * sizeof(float4) + 255) & ~(size_t)255;
    pp.ray_pix = (uint32_t *)w;
    w += (n_pix * sizeof(uint32_t) + 255) & ~(size_t)255;
    uint32_t *block_cnt = (uint32_t *)w;
    w += ((size_t)n_blocks * sizeof(uint32_t) + 255) & ~(size_t)255;
    pp.n_active = (uint32_t *)w;
    w += 256;
    pp.heads = (uint32_t *)w;

    hipError_t e;
    const bool any_tap = taps.cell_flag || taps.t || taps.checked || taps.node;
    /* the frames start black: a memset for sweeps, the count kernel's own threads for the few frames of a short call */
    /* (a kept list: no count kernel, so a memset in every case; the compact kernel runs only for the chords of a first call -- it
     * writes the same list again) */
    const uint32_t clear_isos = !list_kept && iso_count < kCountClearsBelow && n_pix * iso_count <= kCountClearsPixels ? iso_count : 0u;
    if (!clear_isos && (e = clear_frames(stream, d_rgb, (uint64_t)n_pix * 3u * iso_count)) != hipSuccess) return e;
    if (!list_kept)
        hipLaunchKernelGGL(vr_k_count, dim3(n_blocks), dim3(kPreThreads), 0, stream, view, row_begin, n_rows, center_out, block_cnt, d_rgb, clear_isos);
    hipLaunchKernelGGL(vr_k_scan, dim3(tile_order ? 2 : 1), dim3(1024), 0, stream, block_cnt, n_blocks, pp.n_active, pp.heads, iso_count, d_stats,
                       (uint64_t)n_pix, (uint32_t)((n_pix + 63u) / 64u), tile_cost, tile_order, lpt_max_tiles(grid_lanes),
                       chord_order ? tile_cost - kLptHeaderWords : (uint32_t *)nullptr, list_kept ? 1u : 0u);
    if (!list_kept || chord_order)
        hipLaunchKernelGGL(vr_k_compact, dim3(n_blocks), dim3(kPreThreads), 0, stream, view, row_begin, n_rows, center_out, block_cnt,
                           pp.ray_dt, pp.ray_pix, chord_order ? tile_cost : (uint32_t *)nullptr, lpt_max_tiles(grid_lanes), pp.n_active);
    if (chord_order) /* a first call: the order from the chords the kernel above has left */
        hipLaunchKernelGGL(vr_k_tile_order, dim3(1), dim3(1024), 0, stream, (uint32_t)((n_pix + 63u) / 64u), tile_cost, chord_order,
                           lpt_max_tiles(grid_lanes));
    if (any_tap)
        hipLaunchKernelGGL(vr_k_tap_defaults, dim3(2048), dim3(256), 0, stream, taps, (uint64_t)n_pix * iso_count, oob_word);
    return hipGetLastError();
}

} // namespace

/* Scratch of the entry point that allocates nothing (vr_render_device_noalloc): the ray list, tile offsets, ray
 * count and queue heads of one call.  Every GPU the library runs on holds its own copy until the process ends,
 * and every call on a GPU shares that copy.  In host code this name is a stand-in, not a device address: the
 * address comes from hipGetSymbolAddress, once per device (vr_device_setup), and only that pointer is handed to
 * kernels and memsets. */
static __device__ __align__(256) unsigned char scratch[VR_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_setup_mutex;
bool g_setup_done[kMaxDevices];
void *g_scratch_ptr[kMaxDevices]; /* device address of `scratch` on each GPU */
} // namespace

/* Once per device, with that device current (called by vr_ctx_create): raises the dynamic-LDS ceiling of the
 * persistent kernels (a property of the function on the device) and looks up this GPU's copy of `scratch`. */
hipError_t vr_device_setup(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_setup_mutex);
    if (g_setup_done[device]) return hipSuccess;
    hipError_t e;
    const void *fns[] = { (const void *)vr_k_trace<false>,      (const void *)vr_k_trace<true>,
                          (const void *)vr_k_sweep<false, SweepNarrow>, (const void *)vr_k_sweep<true, SweepNarrow>,
                          (const void *)vr_k_sweep<false, SweepWide>,   (const void *)vr_k_sweep<true, SweepWide>,
                          (const void *)vr_k_trace_wide<false>, (const void *)vr_k_trace_wide<true>,
                          (const void *)vr_k_union<UnionNarrow>, (const void *)vr_k_union<UnionWide>,
                          (const void *)vr_k_union<UnionNarrow, false>, (const void *)vr_k_union<UnionWide, false> };
    for (const void *fn : fns) { /* what is left of the CU's 160 KiB beside the kernel's static LDS (the trace kernels' counter sums) */
        hipFuncAttributes attr;
        if ((e = hipFuncGetAttributes(&attr, fn)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u - attr.sharedSizeBytes))) != hipSuccess) return e;
    }
    void *p = nullptr;
    if ((e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))) != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = p;
    g_setup_done[device] = true;
    return hipSuccess;
}

void *vr_noalloc_scratch(int device)
{
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_setup_mutex);
    return g_setup_done[device] ? g_scratch_ptr[device] : nullptr;
}

size_t vr_pipeline_info_offset(uint32_t width, uint32_t n_rows)
{
    const size_t n_pix = (size_t)width * n_rows;
    const size_t n_blocks = padded_tiles(width, n_rows) / (kSuper * kSuper);
    return ((n_pix * sizeof(float4) + 255) & ~(size_t)255) + ((n_pix * sizeof(uint32_t) + 255) & ~(size_t)255) +
           ((n_blocks * sizeof(uint32_t) + 255) & ~(size_t)255) + 64u;
}

size_t vr_pipeline_workspace_bytes(uint32_t width, uint32_t n_rows, uint32_t iso_count)
{
    const size_t n_pix = (size_t)width * n_rows;
    const size_t n_blocks = padded_tiles(width, n_rows) / (kSuper * kSuper);
    size_t bytes = 0;
    bytes += (n_pix * sizeof(float4) + 255) & ~(size_t)255; /* ray_dt     */
    bytes += (n_pix * sizeof(uint32_t) + 255) & ~(size_t)255; /* ray_pix    */
    bytes += (n_blocks * sizeof(uint32_t) + 255) & ~(size_t)255; /* super-tile count / offset */
    bytes += 256;                                               /* n_active   */
    bytes += ((size_t)iso_count * sizeof(uint32_t) + 255) & ~(size_t)255; /* queue heads */
    bytes += (size_t)VR_UNION_EXCEPT_CAP * sizeof(VrUnionRecord);          /* iso runs that left a union walk */
    return bytes;
}

static size_t lpt_tiles(uint32_t width, uint32_t n_rows)
{
    const size_t n_tiles = ((size_t)width * n_rows + 63u) / 64u + 1u;
    return n_tiles < VR_LPT_MAX_TILES ? n_tiles : VR_LPT_MAX_TILES; /* longer lists are neither measured nor ordered */
}

size_t vr_pipeline_lpt_bytes(uint32_t width, uint32_t n_rows)
{
    const size_t n = lpt_tiles(width, n_rows);
    return 256u + n * VR_LPT_STRIDE_WORDS * sizeof(uint32_t) + ((n * sizeof(uint32_t) + 255) & ~(size_t)255); /* header, lengths, order */
}

/* the tile lengths and the tile order of the union kernel lie in a buffer of their own (vr_pipeline_lpt_bytes long), one per
 * call that the context remembers: [256-byte header][lengths, one 256-byte line per tile][order] */
static void lpt_arrays(void *lpt_mem, uint32_t width, uint32_t n_rows, uint32_t **cost, uint32_t **order)
{
    char *w = (char *)lpt_mem;
    *cost = (uint32_t *)(w + 256u);
    *order = (uint32_t *)(w + 256u + lpt_tiles(width, n_rows) * VR_LPT_STRIDE_WORDS * sizeof(uint32_t));
}

/* For a union launch, from the bits of lpt_mode (vr_internal.h): where the lengths and the order lie, whether the pre-pass
 * sorts (*sort_into), whether the kernel follows the order (*order) and keeps lengths (*keep); clears the lengths when asked. */
static hipError_t lpt_prepare(hipStream_t stream, int lpt_mode, void *lpt_mem, const VrViewK &view, uint32_t n_rows,
                              uint32_t **order, uint32_t **sort_into, uint32_t **chord_into, uint32_t **cost, uint32_t *keep)
{
    *order = nullptr; *sort_into = nullptr; *chord_into = nullptr; *cost = nullptr; *keep = 0u;
    if (lpt_mode == 0 || !lpt_mem) return hipSuccess;
    uint32_t *o;
    lpt_arrays(lpt_mem, view.width, n_rows, cost, &o);
    if (lpt_mode & VR_LPT_USE) *order = o;
    if (lpt_mode & VR_LPT_SORT) *sort_into = o;
    if (lpt_mode & VR_LPT_CHORD) *chord_into = o;
    *keep = (lpt_mode & VR_LPT_RECORD) ? 1u : 0u;
    if (lpt_mode & VR_LPT_CLEAR)
        return hipMemsetAsync(*cost - kLptHeaderWords, 0, 256u + lpt_tiles(view.width, n_rows) * VR_LPT_STRIDE_WORDS * sizeof(uint32_t), stream);
    return hipSuccess;
}

/* Runs of iso values that may leave the union walks of one call and be finished by vr_k_union_fix: what the list behind the
 * queue heads holds.  VR_UNION_EXCEPT_CAP in the environment lowers it (tests of the full list: the call is rendered again). */
static uint32_t union_except_cap()
{
    const char *env = getenv("VR_UNION_EXCEPT_CAP");
    if (!env) return VR_UNION_EXCEPT_CAP;
    const long v = atol(env);
    return v < 1 ? 1u : (v > (long)VR_UNION_EXCEPT_CAP ? VR_UNION_EXCEPT_CAP : (uint32_t)v);
}

/* VR_UNION_FORCE_REDO=1 in the environment (tests): the word that makes the fix kernel clear the call's outputs and the sweep
 * kernel render it again is set behind every union launch, as the union kernel sets it when its list is full -- which no
 * volume tried so far comes near (a lane keeps three states of runs that left its front, idle lanes take them over). */
static hipError_t union_force_redo(hipStream_t stream, uint32_t *n_active)
{
    const char *env = getenv("VR_UNION_FORCE_REDO");
    if (!env || !atoi(env)) return hipSuccess;
    return hipMemsetD32Async((hipDeviceptr_t)(n_active + kUnionRedoWord), 1, 1, stream);
}

/* levels of the tree a persistent block keeps in LDS: what is left of the CU's 160 KiB beside the path stacks (and
 * the sweep kernel's counters), at most 14 */
static uint32_t lds_top_levels(uint32_t max_depth, uint64_t n_slots, size_t stat_bytes)
{
    const uint32_t stack_levels = max_depth > 0 ? max_depth : 1u;
    const size_t lds_limit = 160u * 1024u / kTraceBlocksPerCu;
    const size_t trace_stack_bytes = ((size_t)stack_levels + 1u) * kTraceThreads * sizeof(uint32_t);
    uint32_t top_levels = 14;
    while (top_levels > 0 && ((((size_t)1 << top_levels) * sizeof(uint32_t)) + trace_stack_bytes + stat_bytes > lds_limit ||
                              ((uint64_t)1 << top_levels) > n_slots + 1u))
        top_levels--;
    return top_levels;
}

hipError_t vr_launch_render(int variant, hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes_raw,
                            const uint32_t *d_nodes_prepared, uint32_t max_depth, uint32_t iso_begin, uint32_t iso_count,
                            uint32_t row_begin, uint32_t row_end, uint8_t *d_rgb, vr_render_stats *d_stats,
                            const VrTapsK &taps, void *workspace, int n_cus, hipEvent_t ev_begin, hipEvent_t ev_end,
                            uint32_t sweep_items_per_lane, uint32_t launch_seq, bool ranges_nested, int lpt_mode, void *lpt_mem,
                            bool *list_state)
{
    const bool list_kept = list_state && *list_state; /* in: the workspace holds this call's ray list; out: it does now */
    if (list_state) *list_state = false;
    const uint32_t n_rows = row_end - row_begin;
    const uint32_t stack_levels = max_depth > 0 ? max_depth : 1u;
    const bool want_union = variant == VR_KERNEL_UNION;
    if (want_union) variant = VR_KERNEL_DEFAULT;
    {
        /* the persistent kernel adds 32-bit offsets to the low word of the node array's address */
        const uintptr_t p = (uintptr_t)d_nodes_prepared, last = p + (view.n_slots + 1u) * sizeof(uint32_t) - 1u;
        const bool family = variant == VR_KERNEL_DEFAULT || variant == VR_KERNEL_PIPELINE || variant == VR_KERNEL_SWEEP;
        if (family && (p >> 32) != (last >> 32)) variant = VR_KERNEL_RESUME;
    }
    const size_t stack_bytes = ((size_t)stack_levels + 1u) * 256u * sizeof(uint32_t); /* + the spare row above the root */
    hipError_t e;

    const bool use_default = variant == VR_KERNEL_DEFAULT;
    if (!use_default && variant != VR_KERNEL_PIPELINE && variant != VR_KERNEL_SWEEP) {
        const uint32_t tiles_x = (view.width + kBlockW - 1) / kBlockW;
        const uint32_t tiles_y = (n_rows + kBlockH - 1) / kBlockH;
        dim3 grid(tiles_x * tiles_y, iso_count, 1), block(256, 1, 1);
        if (d_stats && (e = hipMemsetAsync(d_stats, 0, sizeof(vr_render_stats) * iso_count, stream)) != hipSuccess) return e;
        if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
        switch (variant) {
        case VR_KERNEL_RESTART_GLOBAL:
            hipLaunchKernelGGL(vr_k_render<VR_KERNEL_RESTART_GLOBAL>, grid, block, 0, stream, view, d_nodes_raw, max_depth,
                               iso_begin, row_begin, n_rows, d_rgb, d_stats, taps);
            break;
        case VR_KERNEL_RESTART_LDS:
            hipLaunchKernelGGL(vr_k_render<VR_KERNEL_RESTART_LDS>, grid, block, 0, stream, view, d_nodes_raw, max_depth,
                               iso_begin, row_begin, n_rows, d_rgb, d_stats, taps);
            break;
        default:
            hipLaunchKernelGGL(vr_k_render<VR_KERNEL_RESUME>, grid, block, stack_bytes, stream, view, d_nodes_prepared,
                               max_depth, iso_begin, row_begin, n_rows, d_rgb, d_stats, taps);
            break;
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
        return hipSuccess;
    }

    /* ---- pipeline ---- */
    const size_t n_pix = (size_t)view.width * n_rows;
    const bool any_tap = taps.cell_flag || taps.t || taps.checked || taps.node;
    /* a block of iso values: the sweep kernel, which shares traversals between iso values (for fewer than 8 the
     * kernel with one traversal per iso value is as fast or a little faster) */
    const bool sweep = variant == VR_KERNEL_SWEEP || want_union || (use_default && iso_count >= VR_UNION_MIN_ISOS);
    /* one traversal per ray for the whole iso range (vr_k_union) where it applies: no taps (the per-pixel check counts
     * need one walk per iso value) and node ranges nested along every path; else traversals shared between iso values */
    const bool use_union = sweep && variant != VR_KERNEL_SWEEP && !any_tap && ranges_nested;
    Prepass pp;
    uint32_t *tile_order = nullptr, *sort_into = nullptr, *chord_into = nullptr, *tile_cost = nullptr, keep_lengths = 0;
    if (use_union && (e = lpt_prepare(stream, lpt_mode, lpt_mem, view, n_rows, &tile_order, &sort_into, &chord_into, &tile_cost, &keep_lengths)) != hipSuccess)
        return e;
    if ((e = launch_prepass(stream, view, iso_count, row_begin, n_rows, sweep ? 1u : 0u, d_rgb, d_stats, taps,
                            VR_FLAG_OOB << 24, workspace, pp, tile_cost, sort_into, (uint32_t)n_cus * kTraceBlocksPerCu * kTraceThreads, chord_into,
                            list_kept)) != hipSuccess)
        return e;
    if (list_state) *list_state = true;
    float4 *ray_dt = pp.ray_dt;
    uint32_t *ray_pix = pp.ray_pix, *n_active = pp.n_active, *heads = pp.heads;

    /* persistent grid: kTraceBlocksPerCu blocks per CU.  A block's LDS holds the path stacks and, in what is
     * left of its share of the CU's 160 KiB, the top K levels of the tree (two thirds of all node reads).
     * More than one iso value: the sweep kernel, which shares traversals between iso values. */
    const size_t trace_stack_bytes = ((size_t)stack_levels + 1u) * kTraceThreads * sizeof(uint32_t);
    const size_t stat_bytes = (sweep ? (size_t)sweep_stat_words(iso_count) * sizeof(uint32_t) : (size_t)kSlotIsoMax * 6u * sizeof(unsigned long long)) +
                              (use_union ? (size_t)8u * kTraceThreads * sizeof(uint32_t) : 0u); /* + the done bits of the union walk */
    const uint32_t top_levels = lds_top_levels(max_depth, view.n_slots, stat_bytes);
    const uint32_t n_top = 1u << top_levels; /* heap positions 0 .. 2^K - 1 = levels 0 .. K-1 (K = 0: the unused slot 0 only) */
    const size_t trace_lds = (size_t)n_top * sizeof(uint32_t) + trace_stack_bytes + stat_bytes;
    if (trace_lds > 160u * 1024u / kTraceBlocksPerCu) return hipErrorInvalidValue;
    const uint32_t blocks = (uint32_t)n_cus * kTraceBlocksPerCu;
    const uint64_t frame_pixels = n_pix;
    const uint32_t stack_rows = stack_levels + 1u;
    /* counters of a short iso range: summed per block in LDS before they go to the counters */
    const uint32_t block_counters = !sweep && d_stats && iso_count <= kSlotIsoMax ? 1u : 0u;
    const uint32_t *run_always = nullptr;
    void *args_sweep[] = { (void *)&view, (void *)&d_nodes_prepared, (void *)&n_top, (void *)&stack_rows, (void *)&ray_dt, (void *)&ray_pix,
                           (void *)&n_active, (void *)&heads, (void *)&iso_begin, (void *)&iso_count, (void *)&frame_pixels,
                           (void *)&d_rgb, (void *)&d_stats, (void *)&taps, (void *)&sweep_items_per_lane, (void *)&launch_seq,
                           (void *)&run_always };
    void *args_trace[] = { (void *)&view, (void *)&d_nodes_prepared, (void *)&n_top, (void *)&stack_rows, (void *)&ray_dt, (void *)&ray_pix,
                           (void *)&n_active, (void *)&heads, (void *)&iso_begin, (void *)&iso_count, (void *)&frame_pixels,
                           (void *)&d_rgb, (void *)&d_stats, (void *)&block_counters, (void *)&taps };
    VrUnionRecord *except_list = (VrUnionRecord *)((char *)heads + (((size_t)iso_count * sizeof(uint32_t) + 255) & ~(size_t)255));
    uint32_t except_cap = union_except_cap();
    void *args_union[] = { (void *)&view, (void *)&d_nodes_prepared, (void *)&n_top, (void *)&iso_begin, (void *)&iso_count, (void *)&stack_rows,
                           (void *)&d_rgb, (void *)&frame_pixels, (void *)&ray_dt, (void *)&ray_pix, (void *)&n_active, (void *)&heads,
                           (void *)&d_stats, (void *)&except_list, (void *)&except_cap, (void *)&sweep_items_per_lane,
                           (void *)&launch_seq, (void *)&keep_lengths, (void *)&tile_order, (void *)&tile_cost };
    void **args = use_union ? args_union : (sweep ? args_sweep : args_trace);
    const void *fn = use_union ? (d_stats ? (const void *)vr_k_union<UnionNarrow> : (const void *)vr_k_union<UnionNarrow, false>)
                   : sweep ? (any_tap ? (const void *)vr_k_sweep<true, SweepNarrow> : (const void *)vr_k_sweep<false, SweepNarrow>)
                           : (any_tap ? (const void *)vr_k_trace<true> : (const void *)vr_k_trace<false>);

#ifdef VR_MEASURE
    if (use_union) vr_measure_before_launch(stream);
#endif
    if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
    if ((e = hipLaunchKernel(fn, dim3(blocks), dim3(kTraceThreads), args, trace_lds, stream)) != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
#ifdef VR_MEASURE
    if (use_union) vr_measure_after_launch(stream, ray_pix);
#endif
    if (use_union) { /* the few iso values whose own MOVE parted from the front's */
        if ((e = union_force_redo(stream, n_active)) != hipSuccess) return e;
        const uint32_t fix_top = n_top < (1u << 13) ? n_top : (1u << 13); /* 32 KiB of top levels beside the wave's stack */
        hipLaunchKernelGGL(vr_k_union_fix, dim3(2048), dim3(64), ((size_t)stack_rows * 64u + fix_top) * sizeof(uint32_t), stream, view,
                           d_nodes_prepared, ray_dt, ray_pix, n_active, except_list, except_cap, iso_begin, frame_pixels, d_rgb, d_stats,
                           stack_rows, fix_top, heads, iso_count);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        /* ... and, should the union kernel have given up (its list full, a front past its bound), the whole call once more with
         * the kernel that shares traversals between iso values: it leaves at once unless kUnionRedoWord is set */
        const size_t redo_stat = (size_t)sweep_stat_words(iso_count) * sizeof(uint32_t);
        const uint32_t redo_top = 1u << lds_top_levels(max_depth, view.n_slots, redo_stat);
        const uint32_t *run_if = n_active + kUnionRedoWord;
        const uint32_t redo_ipl = VR_SWEEP_ITEMS_DEFAULT, redo_seq = 0u;
        const VrTapsK no_taps = { nullptr, nullptr, nullptr, nullptr };
        void *args_redo[] = { (void *)&view, (void *)&d_nodes_prepared, (void *)&redo_top, (void *)&stack_rows, (void *)&ray_dt, (void *)&ray_pix,
                              (void *)&n_active, (void *)&heads, (void *)&iso_begin, (void *)&iso_count, (void *)&frame_pixels,
                              (void *)&d_rgb, (void *)&d_stats, (void *)&no_taps, (void *)&redo_ipl, (void *)&redo_seq, (void *)&run_if };
        if ((e = hipLaunchKernel((const void *)vr_k_sweep<false, SweepNarrow>, dim3(blocks), dim3(kTraceThreads), args_redo,
                                 (size_t)redo_top * sizeof(uint32_t) + trace_stack_bytes + redo_stat, stream)) != hipSuccess) return e;
    }
    return hipSuccess;
}

/* Wide trees: the pipeline (pre-pass + vr_k_trace_wide) by default, the per-pixel kernel for the variants that
 * name a per-pixel kernel (a cross-check with no ray list, no queues and no lane refill). */
hipError_t vr_launch_render_wide(int variant, hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes,
                                 uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin, uint32_t row_end, uint8_t *d_rgb,
                                 vr_render_stats *d_stats, const VrTapsK &taps, void *workspace, int n_cus,
                                 hipEvent_t ev_begin, hipEvent_t ev_end, uint32_t sweep_items_per_lane, uint32_t launch_seq,
                                 bool ranges_nested, int lpt_mode, void *lpt_mem, bool *list_state)
{
    const bool list_kept = list_state && *list_state; /* as in vr_launch_render */
    if (list_state) *list_state = false;
    const uint32_t n_rows = row_end - row_begin;
    hipError_t e = hipSuccess;
    if (variant == VR_KERNEL_RESTART_GLOBAL || variant == VR_KERNEL_RESTART_LDS || variant == VR_KERNEL_RESUME) {
        const uint32_t tiles = ((view.width + kBlockW - 1) / kBlockW) * ((n_rows + kBlockH - 1) / kBlockH);
        if (d_stats && (e = hipMemsetAsync(d_stats, 0, sizeof(vr_render_stats) * iso_count, stream)) != hipSuccess) return e;
        if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(vr_k_render_wide, dim3(tiles, iso_count), dim3(256), 0, stream, view, d_nodes, iso_begin, row_begin,
                           n_rows, d_rgb, d_stats, taps);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (ev_end) e = hipEventRecord(ev_end, stream);
        return e;
    }
    /* a block of iso values: the sweep kernel with the wide lane code (traversals shared between iso values) */
    const bool sweep = variant == VR_KERNEL_SWEEP || variant == VR_KERNEL_UNION || (variant == VR_KERNEL_DEFAULT && iso_count >= 8u);
    Prepass pp;
    const bool any_tap = taps.cell_flag || taps.t || taps.checked || taps.node;
    const bool use_union = sweep && variant != VR_KERNEL_SWEEP && !any_tap && ranges_nested;
    uint32_t *tile_order = nullptr, *sort_into = nullptr, *chord_into = nullptr, *tile_cost = nullptr, keep_lengths = 0;
    if (use_union && (e = lpt_prepare(stream, lpt_mode, lpt_mem, view, n_rows, &tile_order, &sort_into, &chord_into, &tile_cost, &keep_lengths)) != hipSuccess)
        return e;
    if ((e = launch_prepass(stream, view, iso_count, row_begin, n_rows, sweep ? 1u : 0u, d_rgb, d_stats, taps,
                            VR_FLAG_OOB << VR_WIDE_FLAG_SHIFT, workspace, pp, tile_cost, sort_into,
                            (uint32_t)n_cus * kTraceBlocksPerCu * kTraceThreads, chord_into, list_kept)) != hipSuccess)
        return e;
    if (list_state) *list_state = true;
    /* one traversal per ray for the whole iso range, under the same conditions as on narrow trees (vr_launch_render) */
    if (use_union) {
        const uint32_t n_top = 0u, stack_rows = VR_WIDE_MAX_DEPTH;
        const size_t lds = ((size_t)(stack_rows + 8u) * kTraceThreads + sweep_stat_words(iso_count)) * sizeof(uint32_t);
        if (lds > 160u * 1024u) return hipErrorInvalidValue;
        const uint32_t blocks = (uint32_t)n_cus * kTraceBlocksPerCu;
        const uint64_t frame_pixels = (uint64_t)view.width * n_rows;
        VrUnionRecord *except_list = (VrUnionRecord *)((char *)pp.heads + (((size_t)iso_count * sizeof(uint32_t) + 255) & ~(size_t)255));
        uint32_t except_cap = union_except_cap();
        void *args[] = { (void *)&view, (void *)&d_nodes, (void *)&n_top, (void *)&iso_begin, (void *)&iso_count, (void *)&stack_rows,
                         (void *)&d_rgb, (void *)&frame_pixels, (void *)&pp.ray_dt, (void *)&pp.ray_pix, (void *)&pp.n_active, (void *)&pp.heads,
                         (void *)&d_stats, (void *)&except_list, (void *)&except_cap, (void *)&sweep_items_per_lane,
                         (void *)&launch_seq, (void *)&keep_lengths, (void *)&tile_order, (void *)&tile_cost };
        if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
        const void *fn_union = d_stats ? (const void *)vr_k_union<UnionWide> : (const void *)vr_k_union<UnionWide, false>;
        if ((e = hipLaunchKernel(fn_union, dim3(blocks), dim3(kTraceThreads), args, lds, stream)) != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
        if ((e = union_force_redo(stream, pp.n_active)) != hipSuccess) return e;
        hipLaunchKernelGGL(vr_k_union_fix_wide, dim3(2048), dim3(64), 0, stream, view, d_nodes, pp.ray_dt, pp.ray_pix, pp.n_active,
                           except_list, except_cap, iso_begin, frame_pixels, d_rgb, d_stats, pp.heads, iso_count);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        /* the whole call once more with the sweep kernel, should the union kernel have given up (see vr_launch_render) */
        const uint32_t *run_if = pp.n_active + kUnionRedoWord;
        const uint32_t redo_ipl = VR_SWEEP_ITEMS_DEFAULT, redo_seq = 0u;
        const VrTapsK no_taps = { nullptr, nullptr, nullptr, nullptr };
        void *args_redo[] = { (void *)&view, (void *)&d_nodes, (void *)&n_top, (void *)&stack_rows, (void *)&pp.ray_dt, (void *)&pp.ray_pix,
                              (void *)&pp.n_active, (void *)&pp.heads, (void *)&iso_begin, (void *)&iso_count, (void *)&frame_pixels,
                              (void *)&d_rgb, (void *)&d_stats, (void *)&no_taps, (void *)&redo_ipl, (void *)&redo_seq, (void *)&run_if };
        return hipLaunchKernel((const void *)vr_k_sweep<false, SweepWide>, dim3(blocks), dim3(kTraceThreads), args_redo,
                               ((size_t)stack_rows * kTraceThreads + sweep_stat_words(iso_count)) * sizeof(uint32_t), stream);
    }
    if (sweep) {
        const uint32_t n_top = 0u, stack_rows = VR_WIDE_MAX_DEPTH;
        const size_t lds = ((size_t)stack_rows * kTraceThreads + sweep_stat_words(iso_count)) * sizeof(uint32_t);
        const uint32_t blocks = (uint32_t)n_cus * kTraceBlocksPerCu;
        const uint64_t frame_pixels = (uint64_t)view.width * n_rows;
        const void *fn = any_tap ? (const void *)vr_k_sweep<true, SweepWide> : (const void *)vr_k_sweep<false, SweepWide>;
        const uint32_t *run_always = nullptr;
        void *args[] = { (void *)&view, (void *)&d_nodes, (void *)&n_top, (void *)&stack_rows, (void *)&pp.ray_dt, (void *)&pp.ray_pix,
                         (void *)&pp.n_active, (void *)&pp.heads, (void *)&iso_begin, (void *)&iso_count, (void *)&frame_pixels,
                         (void *)&d_rgb, (void *)&d_stats, (void *)&taps, (void *)&sweep_items_per_lane, (void *)&launch_seq,
                         (void *)&run_always };
        if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
        if ((e = hipLaunchKernel(fn, dim3(blocks), dim3(kTraceThreads), args, lds, stream)) != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (ev_end) e = hipEventRecord(ev_end, stream);
        return e;
    }
    const void *fn = any_tap ? (const void *)vr_k_trace_wide<true> : (const void *)vr_k_trace_wide<false>;
    const size_t lds = (size_t)VR_WIDE_MAX_DEPTH * kWideThreads * sizeof(uint32_t);
    const uint32_t blocks = (uint32_t)n_cus * kWideBlocksPerCu;
    const uint64_t frame_pixels = (uint64_t)view.width * n_rows;
    const uint32_t block_counters = d_stats && iso_count <= kSlotIsoMax ? 1u : 0u;
    void *args[] = { (void *)&view, (void *)&d_nodes, (void *)&pp.ray_dt, (void *)&pp.ray_pix, (void *)&pp.n_active,
                     (void *)&pp.heads, (void *)&iso_begin, (void *)&iso_count, (void *)&frame_pixels, (void *)&d_rgb,
                     (void *)&d_stats, (void *)&block_counters, (void *)&taps };
    if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
    if ((e = hipLaunchKernel(fn, dim3(blocks), dim3(kWideThreads), args, lds, stream)) != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_end && (e = hipEventRecord(ev_end, stream)) != hipSuccess) return e;
    return e;
}

#ifdef VR_MEASURE
hipError_t vr_launch_ray_timeline(hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes_prepared, uint32_t max_depth, uint32_t xi,
                                  uint32_t yi, uint32_t iso, unsigned long long *d_out, uint32_t *d_lat)
{
    const uint32_t stack_rows = (max_depth > 0 ? max_depth : 1u) + 1u;
    uint32_t top_levels = 13; /* 32 KiB of top levels beside the one wave's stack, as vr_k_union_fix has them */
    while (top_levels > 0 && ((uint64_t)1 << top_levels) > view.n_slots + 1u) top_levels--;
    const uint32_t n_top = 1u << top_levels;
    const size_t lds = ((size_t)stack_rows * 64u + n_top + 4096u) * sizeof(uint32_t);
    hipLaunchKernelGGL(vr_k_ray_timeline, dim3(1), dim3(64), lds, stream, view, d_nodes_prepared, n_top, stack_rows, xi, yi, iso, d_out, d_lat);
    return hipGetLastError();
}
#endif
