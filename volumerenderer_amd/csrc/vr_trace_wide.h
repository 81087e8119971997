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
 * Wide trees are walked as uploaded or built, without a flattened copy and without per-node checks: the
 * GPU builder produces valid trees by construction and uploaded ones pass the validating walk
 * vr_host_check_kdt_wide first (every reachable node inside the array, every inner node splitting strictly
 * inside its box -- so a box of size 1 is exactly a leaf and a split replaces a bound by the partition).
 * Heap indices stay below 2^31: levels 0..30.
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

/* One ray in flight.  (idx, d) is the node the walk stands on, [lo, hi) its box, c the cell of the point at t. */
struct VrWideLane {
    float dx, dy, dz, t;
    float rx, ry, rz; /* 1 / d per axis, correctly rounded; NaN where |d| < 1e-8 (see vr_plane_t_r) */
    uint32_t lo, hi, c, idx, d;
    uint32_t checked; /* reference-defined node checks so far */
    uint32_t k, in3;  /* while popping: level of the box in [lo, hi), and per axis whether it holds the cell */
};

/* the ray enters at t (inside the root box): SpaceTracker.new, src/SlimKDT.zig:55-64 */
VR_HD void vr_wide_begin(VrWideLane &L, const VrViewK &v, float dx, float dy, float dz, float t)
{
    L.dx = dx; L.dy = dy; L.dz = dz; L.t = t;
    L.rx = fabsf(dx) < 1e-8f ? NAN : 1.0f / dx;
    L.ry = fabsf(dy) < 1e-8f ? NAN : 1.0f / dy;
    L.rz = fabsf(dz) < 1e-8f ? NAN : 1.0f / dz;
    L.lo = 0; L.hi = vr_pack10(v.rdx, v.rdy, v.rdz);
    L.idx = 0; L.d = 0; L.checked = 0;
    const float px = v.ox + dx * t, py = v.oy + dy * t, pz = v.oz + dz * t;
    L.c = vr_pack10((uint32_t)px, (uint32_t)py, (uint32_t)pz);
}

/* Volume.splitGlobal (src/geo.zig:256-275) on the side of the cell; returns the path word of the level.  The
 * partition lies strictly inside the box (validated tree), so the min / max of the reference is the partition. */
VR_HD uint32_t vr_wide_split(VrWideLane &L, uint32_t part, uint32_t axis, uint32_t sh)
{
    const uint32_t side = vr_wide_field(L.c, sh) < part ? 0u : 1u;
    const uint32_t bound = side ? L.lo : L.hi;
    const uint32_t old = vr_wide_field(bound, sh);
    const uint32_t moved = bound ^ ((old ^ part) << sh);
    L.lo = side ? moved : L.lo;
    L.hi = side ? L.hi : moved;
    L.idx = 2u * L.idx + 1u + side;
    return part | (axis << 10) | (sh << 12) | (side << 17) | (old << 18);
}

/* SpaceTracker.zoom (src/SlimKDT.zig:90-103) for at most max_trips levels: until a node excludes the iso value
 * (VR_GO_EXCLUDED, MOVE next), a leaf holds it (VR_GO_HIT, src/raytrace.zig:178-187) or the budget is used up
 * (VR_GO_MORE) */
template <class NodeFetch, class Path, class Track>
VR_HD uint32_t vr_wide_descend(VrWideLane &L, const VrViewK &v, uint32_t iso, const NodeFetch &fetch, Path &path,
                               uint32_t max_trips, Track &track)
{
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        const uint32_t n = fetch(L.idx);
        L.checked++;
        const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        if (!(mn <= iso && iso <= mx)) { track.excluded(n, iso); return VR_GO_EXCLUDED; }
        track.passed(n); /* the density range sits in bits 0-15 of both node words */
        if (L.hi - L.lo == VR_F_ONES) return VR_GO_HIT;
        const uint32_t axis = VR_WIDE_AXIS(n), part = VR_WIDE_PART(n);
        path.set(L.d, vr_wide_split(L, part, axis, axis * 10u));
        L.d++;
    }
    return VR_GO_MORE;
}

template <class NodeFetch, class Path>
VR_HD uint32_t vr_wide_descend(VrWideLane &L, const VrViewK &v, uint32_t iso, const NodeFetch &fetch, Path &path,
                               uint32_t max_trips)
{
    VrNoIsoTrack none;
    return vr_wide_descend(L, v, iso, fetch, path, max_trips, none);
}

/* vr_wide_descend with the node reads taken off the critical path: both children of a node (heap neighbours
 * 2i + 1, 2i + 2) are requested with one 8-byte read as soon as the node is known (see vr_lane_descend_ahead).
 * fetch.children(i) returns those two words (any two when the node has no children).  Same results. */
template <class NodeFetch, class Path, class Track>
VR_HD uint32_t vr_wide_descend_ahead(VrWideLane &L, const VrViewK &v, uint32_t iso, const NodeFetch &fetch, Path &path,
                                     uint32_t max_trips, Track &track)
{
    uint32_t n = fetch(L.idx);
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        VrNodePair kids = { 0u, 0u };
        if (trip + 1u < max_trips) kids = fetch.children(L.idx);
        L.checked++;
        const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        if (!(mn <= iso && iso <= mx)) { track.excluded(n, iso); return VR_GO_EXCLUDED; }
        track.passed(n);
        if (L.hi - L.lo == VR_F_ONES) return VR_GO_HIT;
        const uint32_t axis = VR_WIDE_AXIS(n), part = VR_WIDE_PART(n);
        const uint32_t word = vr_wide_split(L, part, axis, axis * 10u);
        path.set(L.d, word);
        L.d++;
        n = (word & (1u << 17)) ? kids.right : kids.left;
    }
    return VR_GO_MORE;
}

/* Puts a lane back at the entry point of its ray (for the next iso value of a sweep). */
VR_HD void vr_wide_restart(VrWideLane &L, const VrViewK &v, float t_entry, uint32_t cell_entry)
{
    L.t = t_entry;
    L.c = cell_entry;
    L.lo = 0; L.hi = vr_pack10(v.rdx, v.rdy, v.rdz);
    L.idx = 0; L.d = 0; L.checked = 0;
}

/* MOVE (src/raytrace.zig:199-246) out of the box of the excluded node.  Returns false when the point has left
 * the root box (the ray is a miss); otherwise vr_wide_pop comes next. */
VR_HD bool vr_wide_move(VrWideLane &L, const VrViewK &v, uint32_t &escapes)
{
    const uint32_t lo = L.lo, hi = L.hi;
    const float lx = (float)(lo & 0x3ffu), ly = (float)((lo >> 10) & 0x3ffu), lz = (float)(lo >> 20);
    const float hx = (float)(hi & 0x3ffu), hy = (float)((hi >> 10) & 0x3ffu), hz = (float)(hi >> 20);
    /* the three divisions by a per-ray reciprocal with FMA corrections (vr_div: bit-identical to the IEEE
     * divide); an axis the ray is parallel to gives NaN, which drops out of the minimum as INFINITY would */
    float t = vr_plane_t_r(L.dx >= 0.0f ? hx : lx, v.ox, L.dx, L.rx);
    t = fminf(t, vr_plane_t_r(L.dy >= 0.0f ? hy : ly, v.oy, L.dy, L.ry));
    t = fminf(t, vr_plane_t_r(L.dz >= 0.0f ? hz : lz, v.oz, L.dz, L.rz));
    t += 0.0001f;
    float px, py, pz, i = 0.0f;
    for (;;) { /* escape help, :233-242 */
        px = v.ox + L.dx * t; py = v.oy + L.dy * t; pz = v.oz + L.dz * t;
        if (!(lx <= px && px < hx && ly <= py && py < hy && lz <= pz && pz < hz)) break;
        t += 0.001f * i;
        i += 1.0f;
        escapes++;
    }
    L.t = t;
    /* the reference resets to the root here; its loop condition (:161) is the root box test */
    if (!vr_root_contains(v, px, py, pz)) return false;
    const uint32_t c = vr_pack10((uint32_t)px, (uint32_t)py, (uint32_t)pz);
    L.c = c;
    uint32_t in3 = 0;
    for (uint32_t a = 0; a < 3u; a++) {
        const uint32_t ca = vr_wide_field(c, a * 10u);
        in3 |= (vr_wide_field(lo, a * 10u) <= ca && ca < vr_wide_field(hi, a * 10u) ? 1u : 0u) << a;
    }
    L.in3 = in3;
    L.k = L.d;
    return true;
}

/* The reference's restart at the root after a MOVE, folded into a step to the other child of the lowest ancestor
 * that holds the new point: undo splits from the bottom (at most max_trips of them per call) until the box holds
 * the new cell -- for a point inside the root box, lo <= cell < hi on the integer cell is Range.containsFloat on
 * the point -- then credit the checks of levels 0..k and take the child on the point's side.  Returns false when
 * the budget ran out first (call again). */
template <class Path>
VR_HD bool vr_wide_pop(VrWideLane &L, Path &path, uint32_t max_trips)
{
    uint32_t lo = L.lo, hi = L.hi, in3 = L.in3, k = L.k, e = 0;
    const uint32_t c = L.c;
    for (uint32_t trip = 0; in3 != 7u && k > 0u; trip++) {
        if (trip == max_trips) {
            L.lo = lo; L.hi = hi; L.in3 = in3; L.k = k;
            return false;
        }
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
    L.lo = lo; L.hi = hi;
    if (k < L.d) {
        /* the node at level k passed its test before and holds the new point: the reference's walk from the
         * root checks levels 0..k again and takes the child on the point's side */
        L.checked += k + 1u;
        L.idx = ((L.idx + 1u) >> (L.d - k)) - 1u;
        path.set(k, vr_wide_split(L, e & 0x3ffu, (e >> 10) & 3u, (e >> 12) & 31u));
        L.d = k + 1u;
    }
    return true;
}

/* one whole ray */
template <bool AHEAD, class NodeFetch, class Path>
VR_HD VrHit vr_trace_wide_impl(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Path &path,
                               uint32_t pop_trips, uint32_t ahead_trips)
{
    VrHit h;
    h.t = 0.0f; h.flag = 2u; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (ray.oob) return h;
    VrWideLane L;
    vr_wide_begin(L, v, ray.dx, ray.dy, ray.dz, ray.t);
    h.flag = 0u;
    for (;;) {
        uint32_t go;
        if constexpr (AHEAD) { /* the bounded, reading-ahead descents of the persistent kernels */
            VrNoIsoTrack none;
            do go = vr_wide_descend_ahead(L, v, iso, fetch, path, ahead_trips, none); while (go == VR_GO_MORE);
        } else {
            go = vr_wide_descend(L, v, iso, fetch, path, VR_WIDE_MAX_DEPTH + 1u);
        }
        if (go == VR_GO_HIT) { h.flag = 1u; h.node = L.idx; h.cell = L.c; break; }
        h.moves++;
        const bool in_root = vr_wide_move(L, v, h.escapes);
        if (L.checked > VR_MAX_CHECKED) { h.error = 1; break; } /* the bound on a ray's checks: given up behind the MOVE, as in the persistent kernels */
        if (!in_root) break;
        while (!vr_wide_pop(L, path, pop_trips)) {}
    }
    h.t = L.t;
    h.checked = L.checked;
    return h;
}

template <class NodeFetch, class Path>
VR_HD VrHit vr_trace_wide(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Path &path,
                          uint32_t pop_trips = 64u)
{
    return vr_trace_wide_impl<false>(v, ray, iso, fetch, path, pop_trips, 0u);
}

/* the same ray with descents of at most ahead_trips levels that read both children ahead; fetch needs children() */
template <class NodeFetch, class Path>
VR_HD VrHit vr_trace_wide_ahead(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Path &path,
                                uint32_t pop_trips, uint32_t ahead_trips)
{
    return vr_trace_wide_impl<true>(v, ray, iso, fetch, path, pop_trips, ahead_trips);
}

#endif /* VR_TRACE_WIDE_H */
