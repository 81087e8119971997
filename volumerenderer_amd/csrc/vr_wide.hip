/*
 * vr_wide.hip -- render kernel for volumes beyond the reference's u8 coordinates (up to 1024 voxels per
 * axis; the reference's SPACESIZE is u8, src/typedef.zig:1).  One thread per pixel and iso value walks
 * the wide node array as uploaded or built (vr_trace_wide.h: resume after a MOVE instead of restart), with
 * the lane's path (one word per level) in LDS as [level][thread] words.
 *
 * A wide tree is up to 2^31 four-byte words (8 GiB); node fetches use 64-bit addresses.
 */
#include "vr_internal.h"
#include "vr_trace_wide.h"

namespace {

constexpr uint32_t kWideThreads = 256; /* a 16 x 16 pixel block; each wave an 8 x 8 tile */

struct WideFetch {
    const uint32_t *nodes;
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const { return nodes[idx]; }
};

struct LdsPath {
    uint32_t *col; /* this thread's column of the [VR_WIDE_MAX_DEPTH][kWideThreads] path store */
    __device__ __forceinline__ uint32_t get(uint32_t d) const { return col[d * kWideThreads]; }
    __device__ __forceinline__ void set(uint32_t d, uint32_t w) { col[d * kWideThreads] = w; }
};

__device__ __forceinline__ uint32_t wide_wave_sum(uint32_t x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ void wide_add_stat(uint64_t *p, unsigned long long v)
{
    if (v) atomicAdd((unsigned long long *)p, v);
}

__global__ __launch_bounds__(kWideThreads) void vr_k_render_wide(VrViewK v, const uint32_t *__restrict__ nodes,
                                                                 uint32_t iso_begin, uint32_t row_begin, uint32_t n_rows,
                                                                 uint8_t *__restrict__ rgb,
                                                                 vr_render_stats *__restrict__ stats, VrTapsK taps)
{
    __shared__ uint32_t s_path[VR_WIDE_MAX_DEPTH * kWideThreads];

    const uint32_t tiles_x = (v.width + 15u) / 16u;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t xi = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t yr = by * 16u + (wave >> 1) * 8u + (lane >> 3); /* row relative to row_begin */
    const uint32_t iso = iso_begin + blockIdx.y;
    const bool valid = xi < v.width && yr < n_rows;

    VrHit h;
    h.t = 0.0f; h.flag = VR_FLAG_OOB; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (valid) {
        const VrRay ray = vr_make_ray(v, xi, yr + row_begin);
        LdsPath path{ s_path + threadIdx.x };
        h = vr_trace_wide(v, ray, iso, WideFetch{ nodes }, path);
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
        const uint32_t c_checked = wide_wave_sum(valid ? h.checked : 0u);
        const uint32_t c_hit = wide_wave_sum(valid && h.flag == VR_FLAG_HIT ? 1u : 0u);
        const uint32_t c_miss = wide_wave_sum(valid && h.flag == VR_FLAG_MISS ? 1u : 0u);
        const uint32_t c_oob = wide_wave_sum(valid && h.flag == VR_FLAG_OOB ? 1u : 0u);
        const uint32_t c_esc = wide_wave_sum(valid ? h.escapes : 0u);
        const uint32_t c_moves = wide_wave_sum(valid ? h.moves : 0u);
        const uint32_t c_err = wide_wave_sum(valid ? h.error : 0u);
        if (lane == 0) {
            vr_render_stats *s = stats + blockIdx.y;
            wide_add_stat(&s->nodes_checked, c_checked);
            wide_add_stat(&s->hit_count, c_hit);
            wide_add_stat(&s->miss_count, c_miss);
            wide_add_stat(&s->oob_count, c_oob);
            wide_add_stat(&s->escape_help_events, c_esc);
            wide_add_stat(&s->moves, c_moves);
            wide_add_stat(&s->errors, c_err);
        }
    }
}

} // namespace

hipError_t vr_launch_render_wide(hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes, uint32_t iso_begin,
                                 uint32_t iso_count, uint32_t row_begin, uint32_t row_end, uint8_t *d_rgb,
                                 vr_render_stats *d_stats, const VrTapsK &taps, hipEvent_t ev_begin, hipEvent_t ev_end)
{
    const uint32_t n_rows = row_end - row_begin;
    const uint32_t tiles = ((view.width + 15u) / 16u) * ((n_rows + 15u) / 16u);
    hipError_t e = hipSuccess;
    if (ev_begin && (e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(vr_k_render_wide, dim3(tiles, iso_count), dim3(kWideThreads), 0, stream, view, d_nodes, iso_begin,
                       row_begin, n_rows, d_rgb, d_stats, taps);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_end) e = hipEventRecord(ev_end, stream);
    return e;
}
