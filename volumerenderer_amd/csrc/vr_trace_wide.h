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
 * Traversal: restart with a path cache.  After a MOVE the reference resets to the root and walks down
 * again (src/SlimKDT.zig:105-114).  Every node on the way down to the node just skipped passed its iso
 * test, so until the new point leaves the old path the walk meets the same nodes with the same test
 * outcome.  The lane keeps the node words of its current path (one per level) and the side taken at
 * each level; the walk after a MOVE replays those levels from the cache -- counting each as one node
 * check, as the reference does -- and goes back to memory from the level where the new point takes the
 * other child.  Hit cell, t, flags and the reference-defined check count are those of the literal loop;
 * only the nodes past the point of divergence are read.
 *
 * Unlike the flattened array of the u8 kernels, wide trees are walked as uploaded, so the loop keeps
 * the index, depth and axis checks of the literal variant (a malformed tree ends the ray with
 * error = 1 instead of reading outside the array).
 *
 * `Path` is the lane's cache: get(d) / set(d, word) for levels 0..VR_MAX_DEPTH-1.
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

template <class NodeFetch, class Path>
VR_HD VrHit vr_trace_wide(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Path &path)
{
    VrHit h;
    h.t = 0.0f; h.flag = 2u; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (ray.oob) return h;
    float t = ray.t;
    uint32_t n_cached = 0; /* levels 0..n_cached-1 of the cache hold the inner nodes of the current path */
    uint32_t sides = 0;    /* bit d: the walk took the upper child at level d */
    for (;;) {
        /* the walk starts at the root: loop condition of :161 with the root box */
        float px = v.ox + ray.dx * t, py = v.oy + ray.dy * t, pz = v.oz + ray.dz * t;
        if (!vr_root_contains(v, px, py, pz)) break;
        const uint32_t cell[3] = { (uint32_t)px, (uint32_t)py, (uint32_t)pz };
        uint32_t lo[3] = { 0, 0, 0 }, hi[3] = { v.rdx, v.rdy, v.rdz };
        uint64_t idx = 0;
        uint32_t d = 0;
        /* levels the previous walk already read: same node, same (passed) iso test */
        while (d < n_cached) {
            const uint32_t n = path.get(d);
            h.checked++;
            const uint32_t axis = VR_WIDE_AXIS(n), part = VR_WIDE_PART(n);
            const uint32_t side = cell[axis] < part ? 0u : 1u;
            if (side) lo[axis] = lo[axis] > part ? lo[axis] : part;
            else hi[axis] = hi[axis] < part ? hi[axis] : part;
            idx = 2u * idx + 1u + side;
            const bool same = ((sides >> d) & 1u) == side;
            sides = (sides & ~(1u << d)) | (side << d);
            d++;
            if (!same) break; /* the new point leaves the old path below this level */
        }
        /* from here on every node comes from memory: SpaceTracker.zoom until a test fails or a leaf holds */
        bool excluded = false;
        while (!excluded) {
            if (idx >= v.n_slots || d > VR_MAX_DEPTH) { h.error = 1; break; }
            const uint32_t n = fetch((uint32_t)idx);
            h.checked++;
            const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
            if (mn <= iso && iso <= mx) {
                if ((hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]) == 1u) { /* :178-187 */
                    h.t = t; h.flag = 1u; h.node = (uint32_t)idx;
                    h.cell = VR_WIDE_CELL(cell[0], cell[1], cell[2]);
                    return h;
                }
                const uint32_t axis = VR_WIDE_AXIS(n), part = VR_WIDE_PART(n);
                if (axis > 2u || d >= VR_MAX_DEPTH) { h.error = 1; break; }
                const uint32_t side = cell[axis] < part ? 0u : 1u;
                if (side) lo[axis] = lo[axis] > part ? lo[axis] : part;
                else hi[axis] = hi[axis] < part ? hi[axis] : part;
                path.set(d, n);
                sides = (sides & ~(1u << d)) | (side << d);
                idx = 2u * idx + 1u + side;
                d++;
            } else {
                excluded = true;
            }
        }
        if (h.error) break;
        n_cached = d;
        /* MOVE: leave the node box through the nearest exit plane, :199-205 */
        t = INFINITY;
        t = fminf(t, vr_plane_t((float)(ray.dx >= 0.0f ? hi[0] : lo[0]), v.ox, ray.dx));
        t = fminf(t, vr_plane_t((float)(ray.dy >= 0.0f ? hi[1] : lo[1]), v.oy, ray.dy));
        t = fminf(t, vr_plane_t((float)(ray.dz >= 0.0f ? hi[2] : lo[2]), v.oz, ray.dz));
        t += 0.0001f;
        if (++h.moves > VR_MAX_MOVES) { h.error = 1; break; }
        float i = 0.0f;
        for (;;) { /* escape help, :233-242 */
            px = v.ox + ray.dx * t; py = v.oy + ray.dy * t; pz = v.oz + ray.dz * t;
            const bool inside = (float)lo[0] <= px && px < (float)hi[0] && (float)lo[1] <= py && py < (float)hi[1] &&
                                (float)lo[2] <= pz && pz < (float)hi[2];
            if (!inside) break;
            t += 0.001f * i;
            i += 1.0f;
            h.escapes++;
        }
    }
    h.t = t;
    h.flag = 0u;
    return h;
}

#endif /* VR_TRACE_WIDE_H */
