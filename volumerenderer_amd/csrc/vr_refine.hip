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

/* ---- one ray, step by step: the reference's debug print as records (src/raytrace.zig:133-198) ---- */
struct RawFetch {
    const uint32_t *nodes;
    __device__ uint32_t operator()(uint32_t idx) const { return nodes[idx]; }
};
struct StepStore {
    vr_ray_step *out;
    uint32_t cap, n, iso;
    __device__ void step(float t, float px, float py, float pz, uint32_t node, uint32_t size, uint32_t word, uint32_t action)
    {
        if (n < cap) {
            vr_ray_step s;
            s.t = t; s.p[0] = px; s.p[1] = py; s.p[2] = pz;
            s.node = node; s.size = size;
            s.min = (uint8_t)(word & 0xffu); s.max = (uint8_t)((word >> 8) & 0xffu);
            s.isoval = (uint8_t)iso; s.action = (uint8_t)action;
            out[n] = s;
        }
        n++;
    }
};

__global__ void vr_k_trace_ray(VrViewK v, const uint32_t *__restrict__ nodes, uint32_t iso, uint32_t xi, uint32_t yi,
                               vr_ray_step *__restrict__ steps, uint32_t cap, vr_ray_report *__restrict__ report)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const VrRay ray = vr_make_ray(v, xi, yi);
    StepStore store{ steps, cap, 0u, iso };
    const VrHit h = vr_trace_restart(v, ray, iso, RawFetch{ nodes }, store);
    vr_ray_report r;
    r.origin[0] = v.ox; r.origin[1] = v.oy; r.origin[2] = v.oz;
    r.dir[0] = ray.dx; r.dir[1] = ray.dy; r.dir[2] = ray.dz;
    r.t_entry = ray.t;
    r.t_final = h.flag == VR_FLAG_OOB ? 0.0f : h.t;
    r.flag = h.flag;
    r.n_steps = store.n;
    r.n_stored = store.n < cap ? store.n : cap;
    *report = r;
}

/* ---- self-test of the lane code's GPU-only fast paths against their portable definitions ---- */
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void vr_k_selftest(uint64_t n, unsigned long long *__restrict__ bad)
{
    uint32_t wrong = 0, w_floor = 0, w_carry = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t a = mix32((uint32_t)i * 2u + 1u), b = mix32((uint32_t)i * 2u + 2u) ^ (uint32_t)(i >> 31);
        /* floor to a saturating i32: every bit pattern, NaN and infinities included */
        const float x = __builtin_bit_cast(float, a);
        int32_t want;
        if (x != x) want = (a >> 31) ? INT32_MIN : INT32_MAX; /* a NaN saturates by its sign: outside every box */
        else if (x >= 2147483648.0f) want = INT32_MAX;
        else if (x < -2147483648.0f) want = INT32_MIN;
        else want = (int32_t)floorf(x);
        w_floor += vr_floor_i32(x) != want;
        /* 2 g + bit through the carry of the lane mask */
        const bool bit = (b >> 7) & 1u;
        w_carry += vr_twice_plus(a >> 1, bit) != 2u * (a >> 1) + (bit ? 1u : 0u);
        /* the reciprocal division of the MOVE against the correctly rounded divide, on operands of the shape the path
         * produces: numerator = plane offset (0..1023) minus an origin coordinate, denominator = a direction component */
        const float off = (float)(b & 1023u), o = ((float)(int32_t)(a >> 8) - 8388608.0f) * (1.0f / 16384.0f);
        float d = ((float)(b >> 9) + 0.5f) * (1.0f / 8388608.0f) * ((a & 1u) ? 1.0f : -1.0f);
        if (fabsf(d) < 1e-8f) d = 1e-8f;
        const float num = off - o, q = num / d, r = 1.0f / d;
        wrong += __builtin_bit_cast(uint32_t, vr_div(num, d, r)) != __builtin_bit_cast(uint32_t, q);
    }
    if (w_floor) atomicAdd(bad, (unsigned long long)w_floor);
    if (w_carry) atomicAdd(bad + 1, (unsigned long long)w_carry);
    if (wrong) atomicAdd(bad + 2, (unsigned long long)wrong);
}

} // namespace

hipError_t vr_launch_selftest(hipStream_t stream, uint64_t n, unsigned long long *d_bad)
{
    hipLaunchKernelGGL(vr_k_selftest, dim3(2048), dim3(256), 0, stream, n, d_bad);
    return hipGetLastError();
}

hipError_t vr_launch_trace_ray(hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes_raw, uint32_t iso, uint32_t xi,
                               uint32_t yi, vr_ray_step *d_steps, uint32_t cap, vr_ray_report *d_report)
{
    hipLaunchKernelGGL(vr_k_trace_ray, dim3(1), dim3(64), 0, stream, view, d_nodes_raw, iso, xi, yi, d_steps, cap, d_report);
    return hipGetLastError();
}

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
