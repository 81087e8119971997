/*
 * vr_refine.hip -- the trilinear refinement step behind the ray caster: trilerp (src/raytrace.zig:257-291) on the
 * GPU, and the post-pass that evaluates it at every hit point from the hit cell's 8 corner voxels
 * (Data.getCornerDens, src/Data.zig:102-114) -- the only place on the path where voxels are read at render time.
 * HBM-bound streaming over the taps (8 B read + 4 B written per pixel) plus 8 scattered bytes per hit.
 */
#include "vr_internal.h"
#include "vr_trilerp.h"
#include "vr_trace_wide.h"

namespace {

__global__ __launch_bounds__(256) void vr_k_trilerp(const float *__restrict__ in, uint64_t n, float *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *p = in + i * 17u;
    float vals[8];
#pragma unroll
    for (int k = 0; k < 8; k++) vals[k] = p[9 + k];
    out[i] = vr_trilerp(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], vals);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void vr_k_refine(VrViewK v, const uint8_t *__restrict__ vox, uint32_t ry, uint32_t rz,
                                                   uint32_t row_begin, uint32_t n_rows, const uint32_t *__restrict__ cell_flag,
                                                   const float *__restrict__ t, float *__restrict__ density)
{
    const uint64_t frame = (uint64_t)v.width * n_rows;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; /* pixel inside the frame */
    if (i >= frame) return;
    const uint64_t at = (uint64_t)blockIdx.y * frame + i;
    const uint32_t w = cell_flag[at];
    const uint32_t flag = WIDE ? w >> VR_WIDE_FLAG_SHIFT : w >> 24;
    float d = 0.0f;
    if (flag == VR_FLAG_HIT) {
        const uint32_t cx = WIDE ? w & 0x3ffu : w & 0xffu, cy = WIDE ? (w >> 10) & 0x3ffu : (w >> 8) & 0xffu,
                       cz = WIDE ? (w >> 20) & 0x3ffu : (w >> 16) & 0xffu;
        const uint32_t yr = (uint32_t)(i / v.width), xi = (uint32_t)(i - (uint64_t)yr * v.width);
        const VrRay ray = vr_make_ray(v, xi, row_begin + vr_band_row(v, yr));
        const float tt = t[at];
        const float px = v.ox + ray.dx * tt, py = v.oy + ray.dy * tt, pz = v.oz + ray.dz * tt; /* Ray.point, src/geo.zig:37-39 */
        float c[8];
        vr_corner_dens(vox, ry, rz, cx, cy, cz, c);
        d = vr_trilerp(px, py, pz, (float)cx, (float)(cx + 1u), (float)cy, (float)(cy + 1u), (float)cz, (float)(cz + 1u), c);
    }
    density[at] = d;
}

} // namespace

hipError_t vr_launch_trilerp(hipStream_t stream, const float *d_in, uint64_t n, float *d_out)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(vr_k_trilerp, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, d_in, n, d_out);
    return hipGetLastError();
}

hipError_t vr_launch_refine(hipStream_t stream, const VrViewK &view, bool wide, const uint8_t *d_vox, uint32_t ry, uint32_t rz,
                            uint32_t row_begin, uint32_t n_rows, uint32_t n_frames, const uint32_t *d_cell_flag, const float *d_t,
                            float *d_density)
{
    const uint64_t frame = (uint64_t)view.width * n_rows;
    const dim3 grid((uint32_t)((frame + 255u) / 256u), n_frames);
    if (wide)
        hipLaunchKernelGGL(vr_k_refine<true>, grid, dim3(256), 0, stream, view, d_vox, ry, rz, row_begin, n_rows, d_cell_flag, d_t, d_density);
    else
        hipLaunchKernelGGL(vr_k_refine<false>, grid, dim3(256), 0, stream, view, d_vox, ry, rz, row_begin, n_rows, d_cell_flag, d_t, d_density);
    return hipGetLastError();
}
