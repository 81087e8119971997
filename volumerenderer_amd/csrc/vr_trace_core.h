/*
 * vr_trace_core.h -- per-ray arithmetic of the isosurface ray caster, shared by every kernel.
 *
 * The functions here are written once and compiled for the GPU (hipcc, gfx950) by the product
 * library.  They are also compilable as plain host C++ so that tests can single-step the very
 * same lane code on the CPU (tests/emu); the product never takes that route.
 *
 * Float discipline: strict IEEE f32, evaluation order exactly as the reference writes it, no
 * contraction (build with -ffp-contract=off), correctly rounded division and square root.
 *
 * Reference lines (paths relative to the reference root):
 *   ray generation   src/raytrace.zig:24,38-45      entry / OOB   src/raytrace.zig:122-156
 *   traversal loop   src/raytrace.zig:161-255       zoom / reset  src/SlimKDT.zig:90-114
 *   plane hit        src/geo.zig:406-426            containment   src/geo.zig:173-176,221-226
 *   box split        src/geo.zig:256-275            leaf test     src/geo.zig:315-317
 *   node layout      src/SlimKDT.zig:9-13           shading       src/raytrace.zig:56-61
 */
#ifndef VR_TRACE_CORE_H
#define VR_TRACE_CORE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VR_HD __host__ __device__ __forceinline__
#else
#define VR_HD inline
#endif

#define VR_MAX_DEPTH 31 /* deepest node level a heap index below 2^32 can address */
#define VR_MAX_MOVES (1u << 20) /* safety valve: no ray through a <=256^3 volume skips this many nodes */

/* Boxes and cells are kept as three 8-bit values in 10-bit fields (bits 0-7, 10-17, 20-27) so
 * that one 32-bit subtract compares all three axes at once: bit 8 of every field is a guard. */
#define VR_F_ONES ((1u << 0) | (1u << 10) | (1u << 20))  /* 1 in every field     */
#define VR_F_GUARD ((1u << 8) | (1u << 18) | (1u << 28)) /* bit 8 of every field */

VR_HD uint32_t vr_pack10(uint32_t x, uint32_t y, uint32_t z) { return x | (y << 10) | (z << 20); }
VR_HD uint32_t vr_field(uint32_t packed, uint32_t axis) { return (packed >> (axis * 10u)) & 0xffu; }

/* lo <= c < hi on all three axes (all operands packed, every value in 0..255) */
VR_HD bool vr_box_contains_cell(uint32_t lo, uint32_t hi, uint32_t c)
{
    uint32_t a = (c | VR_F_GUARD) - lo;              /* guard survives where c >= lo */
    uint32_t b = (hi | VR_F_GUARD) - c - VR_F_ONES;  /* guard survives where c <  hi */
    return ((a & b) & VR_F_GUARD) == VR_F_GUARD;
}

/* uniform per-launch constants */
struct VrViewK {
    float ox, oy, oz;   /* src                                   */
    float fx, fy, fz;   /* forw                                  */
    float ux, uy, uz;   /* up                                    */
    float rx, ry, rz;   /* right = forw x up, src/raytrace.zig:38 */
    float pixel_size;   /* viewport_width / f32(width), :24      */
    float half_w;       /* f32(width / 2), :43                   */
    float half_h;       /* f32(height / 2), :41                  */
    uint32_t width, height;
    uint32_t root_hi;   /* packed root_dim (src/SlimKDT.zig:55-64: root box = [0, root_dim)) */
    uint32_t rdx, rdy, rdz;
    uint64_t n_slots;
};

struct VrRay {
    float dx, dy, dz;
    float t;        /* entry t (after the +0.001) */
    bool oob;
};

struct VrHit {
    float t;
    uint32_t flag;     /* VR_FLAG_* */
    uint32_t cell;     /* x | y<<8 | z<<16, 0 unless hit */
    uint32_t node;     /* heap index of the hit leaf, 0 unless hit */
    uint32_t checked;  /* reference-defined node checks */
    uint32_t moves;
    uint32_t escapes;
    uint32_t error;    /* 1: traversal aborted on a malformed tree (depth or index out of range) */
};

/* Plane.rayIntersect for an axis plane, src/geo.zig:406-426.  With a unit axis normal the two dot
 * products reduce to d_a and (off - o_a): the other terms are products with 0 and add +-0. */
VR_HD float vr_plane_t(float off, float o, float d)
{
    if (fabsf(d) < 1e-8f) return INFINITY;
    float t = (off - o) / d;
    return t > 0.0f ? t : -1.0f;
}

/* Range.containsFloat on the three axes of the root box, src/geo.zig:173-176,221-226 */
VR_HD bool vr_root_contains(const VrViewK &v, float px, float py, float pz)
{
    return 0.0f <= px && px < (float)v.rdx && 0.0f <= py && py < (float)v.rdy && 0.0f <= pz && pz < (float)v.rdz;
}

/* src/raytrace.zig:41-45 (direction) and :122-156 (entry into the root box, OOB test) */
VR_HD VrRay vr_make_ray(const VrViewK &v, uint32_t xi, uint32_t yi)
{
    VrRay r;
    float yf = ((float)yi - v.half_h) * v.pixel_size;
    float xf = ((float)xi - v.half_w) * v.pixel_size;
    float sx = v.ux * yf + v.rx * xf; /* up.scale(yf).add(right.scale(xf)) */
    float sy = v.uy * yf + v.ry * xf;
    float sz = v.uz * yf + v.rz * xf;
    float dx = v.fx + sx, dy = v.fy + sy, dz = v.fz + sz;
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float inv = 1.0f / len;
    r.dx = dx * inv;
    r.dy = dy * inv;
    r.dz = dz * inv;
    /* entry plane: the min plane when the ray travels in +axis (dir >= 0), else the max plane */
    float t = 0.0f;
    t = fmaxf(t, vr_plane_t(r.dx >= 0.0f ? 0.0f : (float)v.rdx, v.ox, r.dx));
    t = fmaxf(t, vr_plane_t(r.dy >= 0.0f ? 0.0f : (float)v.rdy, v.oy, r.dy));
    t = fmaxf(t, vr_plane_t(r.dz >= 0.0f ? 0.0f : (float)v.rdz, v.oz, r.dz));
    t += 0.001f;
    r.t = t;
    r.oob = !vr_root_contains(v, v.ox + r.dx * t, v.oy + r.dy * t, v.oz + r.dz * t);
    return r;
}

/* src/raytrace.zig:56-61: 255 - u8(trunc(min(200, t - 300))).  For t - 300 <= -1 the reference's
 * float-to-u8 conversion is undefined; here it is 0. */
VR_HD uint8_t vr_shade(float t)
{
    float s = t - 300.0f;
    if (s > 200.0f) s = 200.0f;
    if (!(s > 0.0f)) s = 0.0f;
    return (uint8_t)(255u - (uint32_t)s);
}

/* ------------------------------------------------------------------------------------------
 * Traversal A: literal restart-from-root loop (src/raytrace.zig:161-247).  One node read per
 * iteration, every MOVE goes back to heap index 0.  This is the anchor the faster variant is
 * checked against on the GPU.
 * ---------------------------------------------------------------------------------------- */
template <class NodeFetch>
VR_HD VrHit vr_trace_restart(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch)
{
    VrHit h;
    h.t = 0.0f; h.flag = 2u; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (ray.oob) return h;
    float t = ray.t;
    uint32_t lo[3] = { 0, 0, 0 }, hi[3] = { v.rdx, v.rdy, v.rdz };
    uint32_t idx = 0, depth = 0;
    for (;;) {
        float px = v.ox + ray.dx * t, py = v.oy + ray.dy * t, pz = v.oz + ray.dz * t;
        bool inside = (float)lo[0] <= px && px < (float)hi[0] && (float)lo[1] <= py && py < (float)hi[1] &&
                      (float)lo[2] <= pz && pz < (float)hi[2];
        if (!inside) break; /* :161 loop condition */
        h.checked++;
        uint32_t cell[3] = { (uint32_t)px, (uint32_t)py, (uint32_t)pz };
        if (idx >= v.n_slots || depth > VR_MAX_DEPTH) { h.error = 1; break; }
        uint32_t n = fetch(idx);
        uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        if (mn <= iso && iso <= mx) {
            if ((hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]) == 1u) {
                h.t = t; h.flag = 1u; h.node = idx;
                h.cell = cell[0] | (cell[1] << 8) | (cell[2] << 16);
                return h;
            }
            uint32_t axis = (n >> 24) & 3u, part = (n >> 16) & 0xffu;
            if (axis > 2u) { h.error = 1; break; }
            if (cell[axis] < part) {
                hi[axis] = hi[axis] < part ? hi[axis] : part;
                idx = 2u * idx + 1u;
            } else {
                lo[axis] = lo[axis] > part ? lo[axis] : part;
                idx = 2u * idx + 2u;
            }
            depth++;
            continue;
        }
        /* MOVE: leave the node box through the nearest exit plane */
        t = INFINITY;
        t = fminf(t, vr_plane_t((float)(ray.dx >= 0.0f ? hi[0] : lo[0]), v.ox, ray.dx));
        t = fminf(t, vr_plane_t((float)(ray.dy >= 0.0f ? hi[1] : lo[1]), v.oy, ray.dy));
        t = fminf(t, vr_plane_t((float)(ray.dz >= 0.0f ? hi[2] : lo[2]), v.oz, ray.dz));
        t += 0.0001f;
        if (++h.moves > VR_MAX_MOVES) { h.error = 1; break; }
        float i = 0.0f;
        for (;;) { /* escape help, :233-242 */
            px = v.ox + ray.dx * t; py = v.oy + ray.dy * t; pz = v.oz + ray.dz * t;
            inside = (float)lo[0] <= px && px < (float)hi[0] && (float)lo[1] <= py && py < (float)hi[1] &&
                     (float)lo[2] <= pz && pz < (float)hi[2];
            if (!inside) break;
            t += 0.001f * i;
            i += 1.0f;
            h.escapes++;
        }
        lo[0] = lo[1] = lo[2] = 0; /* reset to the root, src/SlimKDT.zig:105-114 */
        hi[0] = v.rdx; hi[1] = v.rdy; hi[2] = v.rdz;
        idx = 0; depth = 0;
    }
    h.t = t;
    h.flag = 0u;
    return h;
}

/* ------------------------------------------------------------------------------------------
 * Traversal B: resume instead of restart.
 *
 * After a MOVE the reference walks down again from the root.  Every node on the path to the node
 * just skipped passed its iso test, and the child choice at each level is a pure function of the
 * point, so the new walk follows the old path down to the lowest ancestor whose box still holds
 * the new point -- node for node.  This variant therefore keeps the path (one small record per
 * level), pops ancestors until the box holds the new point, credits the `depth` node checks the
 * reference would have spent getting there, and carries on from that ancestor.  Hit cell, t,
 * flags and the reference-defined checked count are identical; far fewer nodes are read.
 *
 * Stack record per level: axis | old_bound << 2 (the bound of the parent box that the zoom
 * replaced).  `Stack` provides push(level, rec) / pop(level).
 * ---------------------------------------------------------------------------------------- */
template <class NodeFetch, class Stack>
VR_HD VrHit vr_trace_resume(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Stack &stack)
{
    VrHit h;
    h.t = 0.0f; h.flag = 2u; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (ray.oob) return h;
    float t = ray.t;
    uint32_t lo = 0, hi = v.root_hi;
    uint32_t idx = 0, depth = 0;
    /* inside the root box every coordinate is >= 0, so trunc == floor */
    uint32_t cell = vr_pack10((uint32_t)(v.ox + ray.dx * t), (uint32_t)(v.oy + ray.dy * t), (uint32_t)(v.oz + ray.dz * t));
    const bool xpos = ray.dx >= 0.0f, ypos = ray.dy >= 0.0f, zpos = ray.dz >= 0.0f;
    for (;;) {
        /* ---- descend while the node's density range holds the iso value ---- */
        bool done = false;
        for (;;) {
            h.checked++;
            if (idx >= v.n_slots) { h.error = 1; done = true; break; }
            uint32_t n = fetch(idx);
            uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
            if (!(mn <= iso && iso <= mx)) break; /* -> MOVE */
            if (hi - lo == VR_F_ONES) {           /* all three extents are 1: a leaf cell */
                h.t = t; h.flag = 1u; h.node = idx;
                h.cell = (cell & 0xffu) | (((cell >> 10) & 0xffu) << 8) | (((cell >> 20) & 0xffu) << 16);
                return h;
            }
            uint32_t axis = (n >> 24) & 3u, part = (n >> 16) & 0xffu;
            if (axis > 2u || depth >= VR_MAX_DEPTH) { h.error = 1; done = true; break; }
            uint32_t sh = axis * 10u;
            bool left = ((cell >> sh) & 0xffu) < part;
            uint32_t side = left ? hi : lo;
            uint32_t old = (side >> sh) & 0xffu;
            uint32_t nb = left ? (old < part ? old : part) : (old > part ? old : part);
            side = (side & ~(0xffu << sh)) | (nb << sh);
            if (left) hi = side; else lo = side;
            stack.push(depth, axis | (old << 2));
            idx = 2u * idx + (left ? 1u : 2u);
            depth++;
        }
        if (done) break;
        /* ---- MOVE: leave the node box through the nearest exit plane (:199-205) ---- */
        t = INFINITY;
        t = fminf(t, vr_plane_t((float)vr_field(xpos ? hi : lo, 0), v.ox, ray.dx));
        t = fminf(t, vr_plane_t((float)vr_field(ypos ? hi : lo, 1), v.oy, ray.dy));
        t = fminf(t, vr_plane_t((float)vr_field(zpos ? hi : lo, 2), v.oz, ray.dz));
        t += 0.0001f;
        if (++h.moves > VR_MAX_MOVES) { h.error = 1; break; }
        float px, py, pz;
        bool in_root;
        float i = 0.0f;
        for (;;) { /* escape help (:233-242): nudge until the point is outside the skipped box */
            px = v.ox + ray.dx * t; py = v.oy + ray.dy * t; pz = v.oz + ray.dz * t;
            in_root = vr_root_contains(v, px, py, pz);
            if (!in_root) break; /* outside the root is outside every node box */
            cell = vr_pack10((uint32_t)px, (uint32_t)py, (uint32_t)pz);
            if (!vr_box_contains_cell(lo, hi, cell)) break;
            t += 0.001f * i;
            i += 1.0f;
            h.escapes++;
        }
        if (!in_root) break; /* the restarted loop would fail its condition at the root: a miss */
        /* ---- pop ancestors until the box holds the new point ---- */
        while (depth > 0) {
            uint32_t rec = stack.pop(depth - 1);
            uint32_t sh = (rec & 3u) * 10u, old = rec >> 2;
            if (idx & 1u) hi = (hi & ~(0xffu << sh)) | (old << sh); /* left child: the zoom had cut hi */
            else lo = (lo & ~(0xffu << sh)) | (old << sh);
            idx = (idx - 1u) >> 1;
            depth--;
            if (vr_box_contains_cell(lo, hi, cell)) break;
        }
        /* the reference re-checks root .. parent(this node) on its way back down */
        h.checked += depth;
    }
    h.t = t;
    h.flag = 0u;
    return h;
}

#endif /* VR_TRACE_CORE_H */
