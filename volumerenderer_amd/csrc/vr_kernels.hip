/*
 * vr_kernels.hip -- gfx950 kernels of the isosurface ray caster: one primary ray per lane,
 * 8x8-pixel tiles per wave64 so the rays of a wave stay coherent.
 *
 * Follows renderImage / traceRay (src/raytrace.zig:11-106, 117-255); the per-ray arithmetic lives
 * in vr_trace_core.h.  The path is latency/VALU bound pointer chasing over 4-byte nodes: there is
 * no matrix work, so no MFMA.
 */
#include "vr_internal.h"

namespace {

constexpr uint32_t kTileW = 8, kTileH = 8;   /* pixels per wave                       */
constexpr uint32_t kBlockW = 16, kBlockH = 16; /* pixels per 256-thread block (2x2 waves) */
constexpr uint32_t kTopNodes = 4095;         /* heap levels 0..11, 16 KiB of LDS      */

struct GlobalFetch {
    const uint32_t *nodes;
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const { return nodes[idx]; }
};

struct TopLdsFetch {
    const uint32_t *nodes;
    const uint32_t *top;
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const
    {
        return idx < kTopNodes ? top[idx] : nodes[idx];
    }
};

/* one 16-bit record per level, one column per lane: [level][lane] */
struct LdsStack {
    uint16_t *col;
    __device__ __forceinline__ void push(uint32_t level, uint32_t rec) { col[level * 64u] = (uint16_t)rec; }
    __device__ __forceinline__ uint32_t pop(uint32_t level) const { return col[level * 64u]; }
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void vr_k_render(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t iso_begin,
                                                   uint32_t row_begin, uint32_t n_rows, uint8_t *__restrict__ rgb,
                                                   vr_render_stats *__restrict__ stats, VrTapsK taps)
{
    __shared__ uint32_t s_top[VARIANT == VR_KERNEL_RESTART_LDS ? kTopNodes : 1];
    __shared__ uint16_t s_stack[VARIANT == VR_KERNEL_RESUME ? 4 * (VR_MAX_DEPTH + 1) * 64 : 1];

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
        VrRay ray = vr_make_ray(v, xi, yr + row_begin);
        if (VARIANT == VR_KERNEL_RESTART_GLOBAL) {
            h = vr_trace_restart(v, ray, iso, GlobalFetch{ nodes });
        } else if (VARIANT == VR_KERNEL_RESTART_LDS) {
            h = vr_trace_restart(v, ray, iso, TopLdsFetch{ nodes, s_top });
        } else {
            LdsStack st{ s_stack + wave * ((VR_MAX_DEPTH + 1) * 64) + lane };
            h = vr_trace_resume(v, ray, iso, GlobalFetch{ nodes }, st);
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
            if (c_checked) atomicAdd((unsigned long long *)&s->nodes_checked, (unsigned long long)c_checked);
            if (c_hit) atomicAdd((unsigned long long *)&s->hit_count, (unsigned long long)c_hit);
            if (c_miss) atomicAdd((unsigned long long *)&s->miss_count, (unsigned long long)c_miss);
            if (c_oob) atomicAdd((unsigned long long *)&s->oob_count, (unsigned long long)c_oob);
            if (c_esc) atomicAdd((unsigned long long *)&s->escape_help_events, (unsigned long long)c_esc);
            if (c_moves) atomicAdd((unsigned long long *)&s->moves, (unsigned long long)c_moves);
            if (c_err) atomicAdd((unsigned long long *)&s->errors, (unsigned long long)c_err);
        }
    }
}

} // namespace

hipError_t vr_launch_render(int variant, hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes,
                            uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin, uint32_t row_end,
                            uint8_t *d_rgb, vr_render_stats *d_stats, const VrTapsK &taps)
{
    const uint32_t n_rows = row_end - row_begin;
    const uint32_t tiles_x = (view.width + kBlockW - 1) / kBlockW;
    const uint32_t tiles_y = (n_rows + kBlockH - 1) / kBlockH;
    dim3 grid(tiles_x * tiles_y, iso_count, 1), block(256, 1, 1);
    switch (variant) {
    case VR_KERNEL_RESTART_GLOBAL:
        hipLaunchKernelGGL(vr_k_render<VR_KERNEL_RESTART_GLOBAL>, grid, block, 0, stream, view, d_nodes, iso_begin,
                           row_begin, n_rows, d_rgb, d_stats, taps);
        break;
    case VR_KERNEL_RESTART_LDS:
        hipLaunchKernelGGL(vr_k_render<VR_KERNEL_RESTART_LDS>, grid, block, 0, stream, view, d_nodes, iso_begin,
                           row_begin, n_rows, d_rgb, d_stats, taps);
        break;
    default:
        hipLaunchKernelGGL(vr_k_render<VR_KERNEL_RESUME>, grid, block, 0, stream, view, d_nodes, iso_begin, row_begin,
                           n_rows, d_rgb, d_stats, taps);
        break;
    }
    return hipGetLastError();
}
