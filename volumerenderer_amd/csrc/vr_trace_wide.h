/*
 * vr_trace_wide.h -- traceRay (src/raytrace.zig:117-255) for volumes beyond the reference's coordinate
 * range.
 *
 * The reference stores coordinates in u8 (SPACESIZE, src/typedef.zig:1) and partitions in one byte of
 * the SlimNode word (src/SlimKDT.zig:9-13), so it ends at 256 voxels per axis.  The same algorithm on
 * up to 1024 voxels per axis needs a node word with a wider partition field:
 *
 *     wide node word   bits 0-7 min | 8-15 max | 16-27 partition | 28-29 axis
 *
 * in the same heap layout (children of i at 2i+1 / 2i+2, src/SlimKDT.zig:24-30).  The arithmetic on
 * the ray is unchanged: same f32 operations in the same order, on boxes whose bounds go up to 1023.
 *
 * Traversal: resume instead of restart, as in the u8 kernels.  After a MOVE the reference resets to the
 * root and walks down again (src/SlimKDT.zig:105-114).  Every node on the way down to the node just
 * skipped passed its iso test, and the child taken at each level depends on the point alone, so the new
 * walk follows the old path down to the lowest ancestor whose box still holds the new point.  The lane
 * keeps one word per level of its path -- partition, axis, side taken, and the bound the split replaced
 * -- undoes splits from the bottom until the box holds the new point, credits the node checks the
 * reference would have spent from the root down to that ancestor, and goes on from its other child.
 * Hit cell, t, flags and the reference-defined check count are those of the literal loop; only nodes
 * below the ancestor are read.  Boxes are two words of three 10-bit fields (x | y<<10 | z<<20).
 *
 * Unlike the flattened array of the u8 kernels, wide trees are walked as uploaded, so the loop keeps
 * index, depth, axis and partition checks (a malformed tree ends the ray with error = 1 instead of
 * reading outside the array).  Heap indices stay below 2^31: levels 0..30.
 *
 * `Path` is the lane's per-level store: get(d) / set(d, word) for levels 0..VR_WIDE_MAX_DEPTH-1.
 */
#ifndef VR_TRACE_WIDE_H
#define VR_TRACE_WIDE_H

#include "vr_trace_core.h"

#define VR_WIDE_MAX_RES 1024u /* voxels per axis: coordinates fit the 10-bit fields of the cell tap */
#define VR_WIDE_PART(n) (((n) >> 16) & 0xfffu)
#define VR_WIDE_AXIS(n) (((n) >> 28) & 3u)
/* cell / flag tap word of wide contexts: x | y<<10 | z<<20 | VR_FLAG_*<<30 */
#define VR_WIDE_CELL(x, y, z) ((x) | ((y) << 10) | ((z) << 20))
#define VR_WIDE_FLAG_SHIFT 30

#define VR_WIDE_MAX_DEPTH 30u /* deepest node level: heap indices below 2^31 */
#define VR_WIDE_MAX_SLOTS ((uint64_t)1 << 31)

/* path word of one level: bits 0-9 partition | 10-11 axis | 12-16 axis * 10 | 17 side taken | 18-27 the bound
 * (hi on side 0, lo on side 1) the split replaced */
VR_HD uint32_t vr_wide_field(uint32_t packed, uint32_t sh) { return (packed >> sh) & 0x3ffu; }

template <class NodeFetch, class Path>
VR_HD VrHit vr_trace_wide(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Path &path)
{
    VrHit h;
    h.t = 0.0f; h.flag = 2u; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (ray.oob) return h;
    float t = ray.t;
    uint32_t lo = 0, hi = vr_pack10(v.rdx, v.rdy, v.rdz); /* box of the node at (idx, d) */
    uint32_t idx = 0, d = 0;
    float px = v.ox + ray.dx * t, py = v.oy + ray.dy * t, pz = v.oz + ray.dz * t; /* inside the root box: not oob */
    uint32_t c = vr_pack10((uint32_t)px, (uint32_t)py, (uint32_t)pz);
    for (;;) {
        /* SpaceTracker.zoom (src/SlimKDT.zig:90-103) until a node excludes the iso value or a leaf holds it */
        for (;;) {
            if (idx >= v.n_slots) { h.error = 1; break; }
            const uint32_t n = fetch(idx);
            h.checked++;
            const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
            if (!(mn <= iso && iso <= mx)) break;
            if (hi - lo == VR_F_ONES) { /* box size 1, src/raytrace.zig:178-187 */
                h.t = t; h.flag = 1u; h.node = idx; h.cell = c;
                return h;
            }
            const uint32_t axis = VR_WIDE_AXIS(n), part = VR_WIDE_PART(n);
            if (axis > 2u || part >= VR_WIDE_MAX_RES || d >= VR_WIDE_MAX_DEPTH) { h.error = 1; break; }
            const uint32_t sh = axis * 10u;
            const uint32_t side = vr_wide_field(c, sh) < part ? 0u : 1u;
            const uint32_t bound = side ? lo : hi; /* Volume.splitGlobal, src/geo.zig:256-275 */
            const uint32_t old = vr_wide_field(bound, sh);
            const uint32_t nb = side ? (old > part ? old : part) : (old < part ? old : part);
            const uint32_t moved = bound ^ ((old ^ nb) << sh);
            lo = side ? moved : lo;
            hi = side ? hi : moved;
            path.set(d, part | (axis << 10) | (sh << 12) | (side << 17) | (old << 18));
            idx = 2u * idx + 1u + side;
            d++;
        }
        if (h.error) break;
        /* MOVE: leave the node box through the nearest exit plane, :199-205 */
        const float lx = (float)(lo & 0x3ffu), ly = (float)((lo >> 10) & 0x3ffu), lz = (float)(lo >> 20);
        const float hx = (float)(hi & 0x3ffu), hy = (float)((hi >> 10) & 0x3ffu), hz = (float)(hi >> 20);
        t = INFINITY;
        t = fminf(t, vr_plane_t(ray.dx >= 0.0f ? hx : lx, v.ox, ray.dx));
        t = fminf(t, vr_plane_t(ray.dy >= 0.0f ? hy : ly, v.oy, ray.dy));
        t = fminf(t, vr_plane_t(ray.dz >= 0.0f ? hz : lz, v.oz, ray.dz));
        t += 0.0001f;
        if (++h.moves > VR_MAX_MOVES) { h.error = 1; break; }
        float i = 0.0f;
        for (;;) { /* escape help, :233-242 */
            px = v.ox + ray.dx * t; py = v.oy + ray.dy * t; pz = v.oz + ray.dz * t;
            if (!(lx <= px && px < hx && ly <= py && py < hy && lz <= pz && pz < hz)) break;
            t += 0.001f * i;
            i += 1.0f;
            h.escapes++;
        }
        /* the reference resets to the root here; its loop condition (:161) is the root box test */
        if (!vr_root_contains(v, px, py, pz)) break;
        c = vr_pack10((uint32_t)px, (uint32_t)py, (uint32_t)pz);
        /* undo splits until the box holds the new cell (for a point inside the root box, lo <= cell < hi on the
         * integer cell is Range.containsFloat on the point) */
        uint32_t in3 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t a = 0; a < 3u; a++) {
            const uint32_t ca = vr_wide_field(c, a * 10u);
            in3 |= (vr_wide_field(lo, a * 10u) <= ca && ca < vr_wide_field(hi, a * 10u) ? 1u : 0u) << a;
        }
        uint32_t k = d, e = 0;
        while (in3 != 7u && k > 0u) {
            k--;
            e = path.get(k);
            const uint32_t axis = (e >> 10) & 3u, sh = (e >> 12) & 31u, old = (e >> 18) & 0x3ffu;
            const uint32_t keep = ~(0x3ffu << sh), put = old << sh;
            if (e & (1u << 17)) lo = (lo & keep) | put;
            else hi = (hi & keep) | put;
            const uint32_t ca = vr_wide_field(c, sh);
            const uint32_t ok = vr_wide_field(lo, sh) <= ca && ca < vr_wide_field(hi, sh) ? 1u : 0u;
            in3 = (in3 & ~(1u << axis)) | (ok << axis);
        }
        if (k < d) {
            /* the node at level k passed its test before and holds the new point: the reference's walk from the
             * root checks levels 0..k again and takes the child on the point's side */
            h.checked += k + 1u;
            const uint32_t part = e & 0x3ffu, axis = (e >> 10) & 3u, sh = (e >> 12) & 31u;
            const uint32_t side = vr_wide_field(c, sh) < part ? 0u : 1u;
            const uint32_t bound = side ? lo : hi;
            const uint32_t old = vr_wide_field(bound, sh);
            const uint32_t nb = side ? (old > part ? old : part) : (old < part ? old : part);
            const uint32_t moved = bound ^ ((old ^ nb) << sh);
            lo = side ? moved : lo;
            hi = side ? hi : moved;
            path.set(k, part | (axis << 10) | (sh << 12) | (side << 17) | (old << 18));
            idx = 2u * (((idx + 1u) >> (d - k)) - 1u) + 1u + side;
            d = k + 1u;
        }
    }
    h.t = t;
    h.flag = 0u;
    return h;
}

#endif /* VR_TRACE_WIDE_H */
