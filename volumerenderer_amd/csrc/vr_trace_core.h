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

#if defined(__HIP__) /* compiled as HIP source: usable from kernels */
#define VR_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define VR_HD inline
#endif

#define VR_MAX_DEPTH 31 /* deepest node level a heap index below 2^32 can address */
#define VR_MAX_MOVES (1u << 20) /* safety valve of the literal loop: no ray through a <=256^3 volume skips this many nodes */

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

/* 2 * g + bit: on the GPU one add-with-carry fed by the compare's lane mask */
VR_HD uint32_t vr_twice_plus(uint32_t g, bool bit)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long carry_in = __builtin_amdgcn_ballot_w64(bit);
    unsigned long long carry_out;
    uint32_t r;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(g), "s"(carry_in));
    return r;
#else
    return 2u * g + (bit ? 1u : 0u);
#endif
}

/* floor(x) as a saturating i32: one conversion instruction on the GPU.  A NaN saturates by its sign bit like an
 * infinity, so that as an unsigned number it lies outside every box -- Range.containsFloat (src/geo.zig:173-176)
 * is false for a NaN too. */
VR_HD int32_t vr_floor_i32(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    if (x != x) return signbit(x) ? INT32_MIN : INT32_MAX;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x < -2147483648.0f) return INT32_MIN;
    return (int32_t)floorf(x);
#endif
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
    /* a call may render the rows of the image in bands: output row p (counted from the first row of the call)
     * is image row  first + (p / band_rows) * band_pitch + p % band_rows;  band_rows = 0: one contiguous block */
    uint32_t band_rows = 0, band_pitch = 0;
};

/* image row, relative to the first row of the call, of output row p */
VR_HD uint32_t vr_band_row(const VrViewK &v, uint32_t p)
{
    if (v.band_rows == 0u) return p;
    const uint32_t band = p / v.band_rows;
    return band * v.band_pitch + (p - band * v.band_rows);
}

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
/* Hook of the literal loop: step() sees every iteration the way the reference's debug print does
 * (src/raytrace.zig:164-198): t, the point, the node and its box size, its density range, and what happens next. */
#define VR_STEP_ZOOM 0u
#define VR_STEP_HIT 1u
#define VR_STEP_MOVE 2u
struct VrNoSteps {
    VR_HD void step(float, float, float, float, uint32_t, uint32_t, uint32_t, uint32_t) {}
};

template <class NodeFetch, class Steps>
VR_HD VrHit vr_trace_restart(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch, Steps &steps)
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
        const uint32_t size = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
        steps.step(t, px, py, pz, idx, size, n,
                   mn <= iso && iso <= mx ? (size == 1u ? VR_STEP_HIT : VR_STEP_ZOOM) : VR_STEP_MOVE);
        if (mn <= iso && iso <= mx) {
            if (size == 1u) {
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

template <class NodeFetch>
VR_HD VrHit vr_trace_restart(const VrViewK &v, const VrRay &ray, uint32_t iso, NodeFetch fetch)
{
    VrNoSteps none;
    return vr_trace_restart(v, ray, iso, fetch, none);
}

/* ------------------------------------------------------------------------------------------
 * Traversal B: resume instead of restart.
 *
 * After a MOVE the reference walks down again from the root.  Every node on the path to the node
 * just skipped passed its iso test, and the child choice at each level is a pure function of the
 * point, so the new walk follows the old path down to the lowest ancestor whose box still holds
 * the new point -- node for node.  This variant therefore keeps the path (one word per level),
 * pops ancestors until the box holds the new point, credits the node checks the reference would
 * have spent getting there, and carries on from there.  Hit cell, t, flags and the
 * reference-defined checked count are identical; far fewer nodes are read.
 *
 * It walks the PREPARED node array (vr_host_prepare_kdt): produced by a validating walk that
 * guarantees every reachable inner node splits strictly inside its box, which is what lets the
 * loops below run without index, axis or depth checks.  Layout: the reference's heap, 1-based
 * (the node at heap index i sits at position g = i + 1, children at 2g and 2g + 1, parent g >> 1,
 * sibling g ^ 1).  Word format:
 *
 *   bits 0-7 min | 8-15 max | 16-23 partition | 24-28 shift (axis * 10) | 31 leaf
 *
 * The kernels are bound by VALU issue, so the steps below are written to cost as few vector
 * instructions as possible: the path stack is addressed by a running pointer instead of a level
 * index, the reference-defined check count is credited per phase instead of per node, and the exit
 * plane divisions use a per-ray reciprocal with an FMA correction that reproduces IEEE division
 * bit for bit (vr_div).
 *
 * `Stack` is a cursor over the lane's path stack: push(word) stores at the current level and moves
 * one level down, up() moves one level up, peek_up(k) returns the word k levels above the cursor
 * without moving, depth() is the current level.
 * ---------------------------------------------------------------------------------------- */
#define VR_N_LEAF 0x80000000u
/* The bound on a ray's checks.  The reference's loop does not end on every ray (one that runs along a lattice plane can alternate
 * between two states for ever: about 1 random far-away view in 100 has such a ray, the reference's own views have none); here such a
 * ray is given up as a miss and counted in vr_render_stats.errors.  The longest walks of valid rays are a few thousand checks (Skull
 * 1024^2: 1 847; a ray through all of a 1024^3 volume: below 10^5), so 2^23 leaves a factor of a hundred and costs a launch with a
 * stuck ray about a second (2^27: 17 s on a 256^3 volume, minutes on the wide node array). */
#define VR_MAX_CHECKED (1u << 23)

/* n / d, correctly rounded, from r = RN(1 / d): two FMA residual corrections of n * r.  With r the
 * correctly rounded reciprocal, the first correction makes the quotient faithful and the second one
 * lands on the correctly rounded result (Markstein); tests/test_division.py checks 10^8 operand
 * pairs of the range this path produces against the hardware divide. */
VR_HD float vr_div(float n, float d, float r)
{
    const float q0 = n * r;
    const float e0 = fmaf(-d, q0, n);
    const float q1 = fmaf(e0, r, q0);
    const float e1 = fmaf(-d, q1, n);
    return fmaf(e1, r, q1);
}

/* Plane.rayIntersect (src/geo.zig:406-426) with the division done by vr_div.  For an axis the ray is
 * parallel to (|d| < 1e-8: "no intersection", INFINITY in the reference's min) the lane holds r = NaN:
 * the quotient is NaN, passes the `t > 0` select untouched and drops out of the minimum over the three
 * axes (fminf returns the other operand), which is what INFINITY does there. */
VR_HD float vr_plane_t_r(float off, float o, float d, float r)
{
    const float t = vr_div(off - o, d, r);
    return t <= 0.0f ? -1.0f : t;
}

struct VrLane {
    float dx, dy, dz, t;
    float rx, ry, rz; /* 1 / d per axis, correctly rounded; NaN where |d| < 1e-8 (see vr_plane_t_r) */
    uint32_t cell;    /* packed floor(point), valid while the point is inside the root box */
    uint32_t lo, hi;  /* packed box of the current node */
    uint32_t g;       /* 1-based heap position of the current node */
    uint32_t posmask; /* 0x3ff in the field of every axis the ray travels in the + direction */
    uint32_t checked; /* reference-defined node checks of this ray */
};

template <class Stack>
VR_HD void vr_lane_begin(VrLane &L, Stack &stack, const VrViewK &v, float dx, float dy, float dz, float t)
{
    L.dx = dx; L.dy = dy; L.dz = dz; L.t = t;
    L.rx = fabsf(dx) < 1e-8f ? NAN : 1.0f / dx;
    L.ry = fabsf(dy) < 1e-8f ? NAN : 1.0f / dy;
    L.rz = fabsf(dz) < 1e-8f ? NAN : 1.0f / dz;
    /* inside the root box every coordinate is >= 0, so trunc == floor */
    L.cell = vr_pack10((uint32_t)(v.ox + dx * t), (uint32_t)(v.oy + dy * t), (uint32_t)(v.oz + dz * t));
    L.lo = 0; L.hi = v.root_hi;
    L.g = 1; L.checked = 0;
    stack.reset();
    L.posmask = (dx >= 0.0f ? 0x3ffu : 0u) | (dy >= 0.0f ? (0x3ffu << 10) : 0u) | (dz >= 0.0f ? (0x3ffu << 20) : 0u);
}

/* Range.containsInclusive on the node's density range, src/geo.zig:196-198 */
VR_HD bool vr_node_holds(uint32_t n, uint32_t iso) { return (n & 0xffu) <= iso && iso <= ((n >> 8) & 0xffu); }

/* Outcome of a bounded phase of a lane */
#define VR_GO_HIT 0u      /* reached a leaf whose range holds the iso value                     */
#define VR_GO_EXCLUDED 1u /* the node at L.g excludes the iso value: MOVE next                   */
#define VR_GO_MORE 2u     /* trip budget used up: the same phase continues in the next round     */

/* Descends from the current node while the node's density range holds the iso value
 * (src/raytrace.zig:161-194, zoom: src/SlimKDT.zig:90-103), at most `max_trips` levels per call.
 * A wave runs a loop as long as its slowest lane; bounding the trips and letting the few long
 * descents continue in the next round keeps the lanes of a wave busy. */
/* What a traversal learns about OTHER iso values.  The walk of a ray depends on the iso value only
 * through the outcomes of the density-range tests at the nodes it visits; every iso value that gives
 * the same outcome at every one of them gives the same walk: same MOVE points, same `t` bits, same
 * hit cell, same reference-defined check count.  A sweep that goes through the iso values in rising
 * order therefore only needs the largest such value: `upper` starts at 255 and shrinks to
 *   max      at a node whose range [min, max] holds the iso value,
 *   min - 1  at a node excluded because the iso value is below its range
 * (a node excluded from above stays excluded for every larger iso value).
 * VrNoIsoTrack is the no-op used where one iso value is rendered at a time. */
struct VrNoIsoTrack {
    VR_HD void passed(uint32_t) {}
    VR_HD void excluded(uint32_t, uint32_t) {}
};
struct VrIsoUpper {
    uint32_t upper;
    VR_HD void passed(uint32_t n)
    {
        const uint32_t mx = (n >> 8) & 0xffu;
        upper = mx < upper ? mx : upper;
    }
    VR_HD void excluded(uint32_t n, uint32_t iso)
    {
        const uint32_t mn = n & 0xffu;
        const uint32_t lim = iso < mn ? mn - 1u : 255u;
        upper = lim < upper ? lim : upper;
    }
};

template <class Fetch, class Stack, class Track>
VR_HD uint32_t vr_lane_descend(VrLane &L, uint32_t iso, const Fetch &fetch, Stack &stack, uint32_t max_trips, Track &track)
{
    const uint32_t d0 = stack.depth();
    uint32_t out = VR_GO_MORE;
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        const uint32_t n = fetch(L.g);
        const bool holds = vr_node_holds(n, iso);
        if (!holds) { track.excluded(n, iso); out = VR_GO_EXCLUDED; break; }
        track.passed(n);
        if (n & VR_N_LEAF) { out = VR_GO_HIT; break; }
        const uint32_t sh = (n >> 24) & 31u, part = (n >> 16) & 0xffu;
        const bool right = ((L.cell >> sh) & 0xffu) >= part;
        uint32_t side = right ? L.lo : L.hi;
        stack.push(side); /* the bound word the zoom replaces */
        side = (side & ~(0xffu << sh)) | (part << sh); /* partition is strictly inside the box: min/max of geo.zig:256-275 pick it */
        L.lo = right ? side : L.lo;
        L.hi = right ? L.hi : side;
        L.g = vr_twice_plus(L.g, right);
    }
    /* one check per node tested (src/raytrace.zig:162): every level passed, plus the node that ended the descent */
    L.checked += stack.depth() - d0 + (out != VR_GO_MORE ? 1u : 0u);
    return out;
}

template <class Fetch, class Stack>
VR_HD uint32_t vr_lane_descend(VrLane &L, uint32_t iso, const Fetch &fetch, Stack &stack, uint32_t max_trips)
{
    VrNoIsoTrack none;
    return vr_lane_descend(L, iso, fetch, stack, max_trips, none);
}

/* vr_lane_descend with the node reads taken off the lane's critical path: the two children of a node are neighbours
 * in the heap (positions 2g and 2g + 1), so both words are requested with one 8-byte read as soon as the node itself is
 * known, and the read is under way while the node's range test, the child choice and the box update are worked out.
 * fetch.children(g) returns the words at 2g and 2g + 1 (any two words when the node has no children: they are not
 * used then).  Same results as vr_lane_descend. */
struct VrNodePair { uint32_t left, right; };

template <class Fetch, class Stack, class Track>
VR_HD uint32_t vr_lane_descend_ahead(VrLane &L, uint32_t iso, const Fetch &fetch, Stack &stack, uint32_t max_trips, Track &track)
{
    const uint32_t d0 = stack.depth();
    uint32_t out = VR_GO_MORE;
    uint32_t n = fetch(L.g);
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        VrNodePair kids = { 0u, 0u };
        if (trip + 1u < max_trips) kids = fetch.children(L.g); /* the last trip's child is read by the next call */
        const bool holds = vr_node_holds(n, iso);
        if (!holds) { track.excluded(n, iso); out = VR_GO_EXCLUDED; break; }
        track.passed(n);
        if (n & VR_N_LEAF) { out = VR_GO_HIT; break; }
        const uint32_t sh = (n >> 24) & 31u, part = (n >> 16) & 0xffu;
        const bool right = ((L.cell >> sh) & 0xffu) >= part;
        uint32_t side = right ? L.lo : L.hi;
        stack.push(side);
        side = (side & ~(0xffu << sh)) | (part << sh);
        L.lo = right ? side : L.lo;
        L.hi = right ? L.hi : side;
        L.g = vr_twice_plus(L.g, right);
        n = right ? kids.right : kids.left;
    }
    L.checked += stack.depth() - d0 + (out != VR_GO_MORE ? 1u : 0u);
    return out;
}

/* Puts a lane back at the entry point of its ray (for the next iso value of a sweep): everything
 * vr_lane_begin derives from the direction alone is kept. */
template <class Stack>
VR_HD void vr_lane_restart(VrLane &L, Stack &stack, const VrViewK &v, float t_entry, uint32_t cell_entry)
{
    L.t = t_entry;
    L.cell = cell_entry;
    L.lo = 0; L.hi = v.root_hi;
    L.g = 1; L.checked = 0;
    stack.reset();
}

/* MOVE (src/raytrace.zig:199-205) + escape help (:233-242).  Returns false when the new point is
 * outside the root box (the restarted loop would fail its condition there: a miss). */
VR_HD bool vr_lane_move(VrLane &L, const VrViewK &v, uint32_t &escapes)
{
    const uint32_t ex = (L.hi & L.posmask) | (L.lo & ~L.posmask); /* exit plane per axis: max if d >= 0 else min */
    float t = fminf(fminf(vr_plane_t_r((float)(ex & 0xffu), v.ox, L.dx, L.rx),
                          vr_plane_t_r((float)((ex >> 10) & 0xffu), v.oy, L.dy, L.ry)),
                    vr_plane_t_r((float)((ex >> 20) & 0xffu), v.oz, L.dz, L.rz));
    t += 0.0001f;
    float i = 0.0f;
    bool in_root;
    for (;;) {
        const float px = v.ox + L.dx * t, py = v.oy + L.dy * t, pz = v.oz + L.dz * t;
        /* Range.containsFloat per axis (src/geo.zig:173-176): 0 <= p < dim  <=>  floor(p), as unsigned, < dim */
        const uint32_t cx = (uint32_t)vr_floor_i32(px), cy = (uint32_t)vr_floor_i32(py), cz = (uint32_t)vr_floor_i32(pz);
        in_root = (cx < v.rdx) & (cy < v.rdy) & (cz < v.rdz);
        if (!in_root) break; /* outside the root is outside every node box */
        L.cell = vr_pack10(cx, cy, cz);
        if (!vr_box_contains_cell(L.lo, L.hi, L.cell)) break;
        t += 0.001f * i;
        i += 1.0f;
        escapes++;
    }
    L.t = t;
    return in_root;
}

/* After a MOVE: pops ancestors until one holds the new point, then steps straight into that
 * ancestor's OTHER child.  (The ancestor A holds the point, its child C on the old path does not, and
 * A's two children tile A, so the point is in C's sibling; A itself passed its iso test on the way
 * down, so the reference's restarted walk goes root .. A .. sibling.  A is not read again: a stack
 * word is the bound word the zoom into C replaced, the split field is the one where it differs from
 * C's bound, and the partition is C's value there.)
 * Credits the depth(A)+1 node checks the reference spends on root .. A.  The stack word of the next
 * level up is requested before the current one is examined, so its latency hides behind the test.
 * At most `max_trips` levels per call; returns true when the sibling was entered, false when the
 * budget ran out first (call again).
 * Precondition: the point is inside the root box and outside the current box (so depth >= 1). */
template <class Stack>
VR_HD bool vr_lane_pop(VrLane &L, Stack &stack, uint32_t max_trips)
{
    const uint32_t cg = L.cell | VR_F_GUARD, cp = L.cell + VR_F_ONES;
    uint32_t next = stack.peek_up(1);
    uint32_t word = 0;
    bool left = false, found = false;
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        word = next;                 /* A's bound word that the zoom into C replaced */
        stack.up();                  /* cursor at the level of the parent A */
        next = stack.peek_up(1);     /* at the root this reads the spare row: never used */
        left = (L.g & 1u) == 0u;     /* even position = left child: the zoom had cut hi */
        const uint32_t a_lo = left ? L.lo : word, a_hi = left ? word : L.hi;
        const uint32_t a = cg - a_lo, b = (a_hi | VR_F_GUARD) - cp; /* vr_box_contains_cell, shared parts hoisted */
        found = ((a & b) & VR_F_GUARD) == VR_F_GUARD;
        if (found) break;            /* L.lo / L.hi are still C's */
        L.lo = a_lo; L.hi = a_hi;
        L.g >>= 1;
    }
    if (!found) return false;
    /* zoom from A into the sibling of C: the sibling keeps A's bound on C's cut side (`word`) and cuts the
     * side C shares with A at the partition, which is C's value in the one field where `word` and C's cut
     * bound differ.  Adding 0xff to every field of the difference carries into the guard bit exactly where
     * the field is not 0, which gives the field's mask. */
    const uint32_t cut = left ? L.hi : L.lo, keep = left ? L.lo : L.hi;
    const uint32_t carry = ((word ^ cut) + 0xffu * VR_F_ONES) & VR_F_GUARD;
    const uint32_t fm = carry - (carry >> 8);
    const uint32_t other = (keep & ~fm) | (cut & fm);
    stack.push(keep);
    L.lo = left ? other : word;
    L.hi = left ? word : other;
    L.g ^= 1u;
    L.checked += stack.depth();
    return true;
}

VR_HD uint32_t vr_unpack_cell(uint32_t c10) { return (c10 & 0xffu) | (((c10 >> 10) & 0xffu) << 8) | (((c10 >> 20) & 0xffu) << 16); }

/* one ray from entry to hit or miss; `fetch` reads the PREPARED node array */
template <class Fetch, class Stack>
VR_HD VrHit vr_trace_resume(const VrViewK &v, const VrRay &ray, uint32_t iso, const Fetch &fetch, Stack &stack,
                            uint32_t max_trips)
{
    VrHit h;
    h.t = 0.0f; h.flag = 2u; h.cell = 0; h.node = 0; h.checked = 0; h.moves = 0; h.escapes = 0; h.error = 0;
    if (ray.oob) return h;
    VrLane L;
    vr_lane_begin(L, stack, v, ray.dx, ray.dy, ray.dz, ray.t);
    h.flag = 0u;
    for (;;) {
        const uint32_t go = vr_lane_descend(L, iso, fetch, stack, max_trips);
        if (go == VR_GO_MORE) continue;
        if (go == VR_GO_HIT) {
            /* a leaf cell whose range holds the iso value: the hit (src/raytrace.zig:178-187) */
            h.flag = 1u; h.node = L.g - 1u; h.cell = vr_unpack_cell(L.cell);
            break;
        }
        h.moves++;
        const bool in_root = vr_lane_move(L, v, h.escapes);
        /* the bound on a ray's checks (the reference's loop does not end on every ray): the MOVE is made and counted, then the
         * ray is given up as a miss -- the same point at which the persistent kernels give it up, so that all counters agree */
        if (L.checked > VR_MAX_CHECKED) { h.error = 1; break; }
        if (!in_root) break;
        while (!vr_lane_pop(L, stack, max_trips)) {}
    }
    h.t = L.t;
    h.checked = L.checked;
    return h;
}

#endif /* VR_TRACE_CORE_H */
