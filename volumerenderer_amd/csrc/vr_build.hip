/*
 * vr_build.hip -- the centre-split SlimKDT built on the GPU from the uploaded u8 volume
 * (SlimKDT.newHeursiticPartition with AlwaysCenterH, src/SlimKDT.zig:222-322, :359-367).
 *
 * The reference recurses depth first and, per node, scans every corner of every cell of the node's
 * box (:275-281).  Here the same array falls out of two data-parallel passes:
 *   1. one thread per heap slot replays the slot's root-to-node path -- with centre splits the axis
 *      (largestDim, src/geo.zig:319-327) and partition ((min+max)/2, src/SlimKDT.zig:282-284) of every
 *      ancestor follow from the box alone -- and so learns whether the slot holds a node, its box, axis,
 *      partition and whether it is a leaf cell; leaves take their range from their 8 voxel corners
 *      (src/Data.zig:102-114);
 *   2. level by level from the bottom, every inner node takes the union of its children's ranges (the
 *      children's boxes share the voxel plane at the partition, so together they cover exactly the
 *      parent's voxels).
 * Both the reference-format array and the flattened array the render kernels walk are written.
 * HBM-bound integer work: 8 bytes written per slot, 8 voxel bytes read per leaf.
 *
 * WIDE = true builds the tree of a volume beyond the reference's u8 coordinates (up to 1024 voxels per
 * axis) in the wide node word of vr_trace_wide.h (12-bit partition, axis at bit 28); only that one array
 * is written.  No node word of either format is 0 (an inner node's partition is at least 1, a leaf's
 * axis is 2), so 0 marks an empty slot, and a node whose first child slot is empty is a leaf.
 */
#include "vr_internal.h"

namespace {

struct Box3 {
    uint32_t lo[3], hi[3];
    __device__ __forceinline__ uint32_t len(int a) const { return hi[a] - lo[a]; }
    __device__ __forceinline__ bool unit() const { return len(0) * len(1) * len(2) == 1u; }
    __device__ __forceinline__ int largest() const
    {
        if (len(0) > len(1) && len(0) > len(2)) return 0;
        if (len(1) > len(2)) return 1;
        return 2;
    }
};

template <bool WIDE>
__global__ __launch_bounds__(256) void vr_k_build_structure(const uint8_t *__restrict__ vox, uint32_t rx, uint32_t ry,
                                                           uint32_t rz, uint64_t n_slots, uint32_t *__restrict__ raw,
                                                           uint32_t *__restrict__ prepared)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_slots) return;
    const uint32_t g = (uint32_t)idx + 1u;
    const int level = 31 - __clz((int)g);
    Box3 b;
    b.lo[0] = b.lo[1] = b.lo[2] = 0;
    b.hi[0] = rx - 1u; b.hi[1] = ry - 1u; b.hi[2] = rz - 1u;
    bool exists = true;
    for (int s = level - 1; s >= 0; s--) {
        if (b.unit()) { exists = false; break; } /* an ancestor is a leaf cell: nothing below it */
        const int a = b.largest();
        const uint32_t part = (b.lo[a] + b.hi[a]) >> 1;
        if ((g >> s) & 1u) {
            if (a == 0) b.lo[0] = part; else if (a == 1) b.lo[1] = part; else b.lo[2] = part;
        } else {
            if (a == 0) b.hi[0] = part; else if (a == 1) b.hi[1] = part; else b.hi[2] = part;
        }
    }
    uint32_t w_raw = 0, w_prep = 0;
    if (exists) {
        const int a = b.largest();
        const uint32_t part = (b.lo[a] + b.hi[a]) >> 1;
        uint32_t mn = 0, mx = 0;
        const bool leaf = b.unit();
        if (leaf) {
            mn = 255u;
            const uint64_t sx = (uint64_t)ry * rz, sy = rz;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t v = vox[(b.lo[0] + (k & 1u)) * sx + (b.lo[1] + ((k >> 1) & 1u)) * sy + (b.lo[2] + (k >> 2))];
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
        }
        w_raw = mn | (mx << 8) | (part << 16) | ((uint32_t)a << (WIDE ? 28 : 24));
        w_prep = mn | (mx << 8) | (part << 16) | (leaf ? 0x80000000u : ((uint32_t)a * 10u) << 24);
    }
    raw[idx] = w_raw;
    if (!WIDE) {
        prepared[idx + 1] = w_prep;
        if (idx == 0) prepared[0] = 0;
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void vr_k_build_ranges(uint64_t first, uint64_t count, uint32_t *__restrict__ raw,
                                                        uint32_t *__restrict__ prepared)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t idx = first + i;
    uint32_t p = 0;
    if (WIDE) {
        if (raw[idx] == 0u || raw[2 * idx + 1] == 0u) return; /* empty slot or leaf */
    } else {
        p = prepared[idx + 1];
        if (p == 0u || (p & 0x80000000u)) return; /* empty slot or leaf */
    }
    const uint32_t c1 = raw[2 * idx + 1], c2 = raw[2 * idx + 2];
    const uint32_t mn1 = c1 & 0xffu, mn2 = c2 & 0xffu, mx1 = (c1 >> 8) & 0xffu, mx2 = (c2 >> 8) & 0xffu;
    const uint32_t r = (mn1 < mn2 ? mn1 : mn2) | ((mx1 > mx2 ? mx1 : mx2) << 8);
    raw[idx] = (raw[idx] & 0xffff0000u) | r;
    if (!WIDE) prepared[idx + 1] = (p & 0xffff0000u) | r;
}

uint32_t ceil_log2(uint32_t v)
{
    uint32_t l = 0;
    while ((1u << l) < v) l++;
    return l;
}

} // namespace

uint32_t vr_center_tree_depth(uint32_t rx, uint32_t ry, uint32_t rz)
{
    /* every split halves one axis (floor / ceil); the deepest leaf takes the larger half every time */
    return ceil_log2(rx - 1u) + ceil_log2(ry - 1u) + ceil_log2(rz - 1u);
}

hipError_t vr_launch_build_center(hipStream_t stream, const uint8_t *d_vox, uint32_t rx, uint32_t ry, uint32_t rz,
                                  uint32_t depth, uint32_t *d_raw, uint32_t *d_prepared)
{
    const uint64_t n_slots = (uint64_t)1 << (depth + 1);
    if (!d_prepared) { /* wide node words, one array */
        hipLaunchKernelGGL(vr_k_build_structure<true>, dim3((uint32_t)((n_slots + 255u) / 256u)), dim3(256), 0, stream, d_vox,
                           rx, ry, rz, n_slots, d_raw, (uint32_t *)nullptr);
        for (int level = (int)depth - 1; level >= 0; level--) {
            const uint64_t first = ((uint64_t)1 << level) - 1u, count = (uint64_t)1 << level;
            hipLaunchKernelGGL(vr_k_build_ranges<true>, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, first,
                               count, d_raw, (uint32_t *)nullptr);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(vr_k_build_structure<false>, dim3((uint32_t)((n_slots + 255u) / 256u)), dim3(256), 0, stream, d_vox, rx, ry,
                       rz, n_slots, d_raw, d_prepared);
    for (int level = (int)depth - 1; level >= 0; level--) {
        const uint64_t first = ((uint64_t)1 << level) - 1u, count = (uint64_t)1 << level;
        hipLaunchKernelGGL(vr_k_build_ranges<false>, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, first,
                           count, d_raw, d_prepared);
    }
    return hipGetLastError();
}
