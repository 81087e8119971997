/*
 * vr_trace_union_wide.h -- the union walk (vr_trace_union.h: ONE traversal per ray for a whole range of iso values) on
 * the wide trees of vr_trace_wide.h (volumes beyond the reference's u8 coordinates, up to 1023 cells per axis).
 *
 * Same front, same rules; what differs is the geometry: nodes are addressed by 0-based heap index, a lane keeps one
 * path word per level (partition, axis, side taken, the bound the split replaced) instead of a stack of bound words,
 * and boxes are two words of three 10-bit fields.  The iso-set side (runs, done bits, kept runs) is shared with the
 * narrow lane code.
 */
#ifndef VR_TRACE_UNION_WIDE_H
#define VR_TRACE_UNION_WIDE_H

#include "vr_trace_union.h"
#include "vr_trace_wide.h"

struct VrUnionWideLane {
    VrWideLane L;                /* ray, point, current node (idx, d) and its box */
    uint32_t prev_mn, prev_mx;   /* range of the parent of the current node ([0, 255] above the root) */
    uint32_t alive_lo, alive_hi; /* smallest / largest iso value not done; lo > hi: none left */
    bool holes;                  /* see VrUnionLane */
};

template <class Done>
VR_HD void vr_uw_begin(VrUnionWideLane &U, Done &done, const VrViewK &v, float dx, float dy, float dz, float t, uint32_t iso_begin,
                       uint32_t iso_count)
{
    vr_wide_begin(U.L, v, dx, dy, dz, t);
    U.prev_mn = 0u; U.prev_mx = 255u;
    U.alive_lo = iso_begin; U.alive_hi = iso_begin + iso_count - 1u;
    U.holes = false;
    (void)done;
}

template <class Done>
VR_HD void vr_uw_restart(VrUnionWideLane &U, Done &done, const VrViewK &v, float t, uint32_t cell, const VrUnionRuns &runs)
{
    vr_wide_restart(U.L, v, t, cell);
    vr_union_set_runs(U, done, runs);
}

/* vr_union_descend on a wide tree (HOLES as there) */
template <bool HOLES, class Fetch, class Path, class Done, class Sink>
VR_HD uint32_t vr_uw_descend(VrUnionWideLane &U, const Fetch &fetch, Path &path, Done &done, Sink &sink, uint32_t max_trips)
{
    VrWideLane &L = U.L;
    const bool holes = HOLES ? U.holes : false;
    VR_ROLLED
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        const uint32_t n = fetch(L.idx);
        const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        const uint32_t checks = L.d + 1u; /* root .. this node (src/raytrace.zig:162) */
        if (mn > U.prev_mn) {
            const uint32_t p = U.prev_mn > U.alive_lo ? U.prev_mn : U.alive_lo, q = mn - 1u < U.alive_hi ? mn - 1u : U.alive_hi;
            if (p <= q) vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.park(r0, r1, checks); }, holes);
        }
        if (mx < U.prev_mx) {
            const uint32_t p = mx + 1u > U.alive_lo ? mx + 1u : U.alive_lo, q = U.prev_mx < U.alive_hi ? U.prev_mx : U.alive_hi;
            if (p <= q) vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.park(r0, r1, checks); }, holes);
        }
        if (!vr_union_any(done, U, mn, mx, holes)) return VR_GO_EXCLUDED;
        if (L.hi - L.lo == VR_F_ONES) { /* a box of size 1 is a leaf: every present iso value in its range hits here */
            const uint32_t p = mn > U.alive_lo ? mn : U.alive_lo, q = mx < U.alive_hi ? mx : U.alive_hi;
            vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.hit(r0, r1, checks, L.t, L.c, L.idx); }, holes);
            if (!holes && (p == U.alive_lo || q == U.alive_hi)) {
                if (p == U.alive_lo) U.alive_lo = q + 1u;
                else U.alive_hi = p - 1u;
            } else {
                vr_union_need_bits(done, U);
                vr_union_mark(done, p, q);
                vr_union_shrink(done, U);
            }
            return VR_GO_EXCLUDED;
        }
        const uint32_t axis = VR_WIDE_AXIS(n), part = VR_WIDE_PART(n);
        path.set(L.d, vr_wide_split(L, part, axis, axis * 10u));
        L.d++;
        U.prev_mn = mn; U.prev_mx = mx;
    }
    return VR_GO_MORE;
}

/* vr_wide_move (MOVE + escape help, src/raytrace.zig:199-246) that also says which planes gave t, as vr_union_move does:
 * `ex` = the exit planes of the box that was left (its upper bound where the ray travels in +, else its lower one),
 * `tied` = all ones in the field of every axis whose plane gave exactly the smallest t, `pm` = all ones in the field of
 * every axis the ray travels in the + direction.  in_root false: the point has left the volume (L.c, L.in3, L.k are
 * not set then). */
struct VrUnionWideExit { uint32_t ex, tied, pm, esc; bool in_root; };

VR_HD VrUnionWideExit vr_uw_move(VrWideLane &L, const VrViewK &v)
{
    VrUnionWideExit x;
    const uint32_t lo = L.lo, hi = L.hi;
    const float lx = (float)(lo & 0x3ffu), ly = (float)((lo >> 10) & 0x3ffu), lz = (float)(lo >> 20);
    const float hx = (float)(hi & 0x3ffu), hy = (float)((hi >> 10) & 0x3ffu), hz = (float)(hi >> 20);
    const float tx = vr_plane_t_r(L.dx >= 0.0f ? hx : lx, v.ox, L.dx, L.rx);
    const float ty = vr_plane_t_r(L.dy >= 0.0f ? hy : ly, v.oy, L.dy, L.ry);
    const float tz = vr_plane_t_r(L.dz >= 0.0f ? hz : lz, v.oz, L.dz, L.rz);
    float t = fminf(fminf(tx, ty), tz);
    x.pm = (L.dx >= 0.0f ? 0x3ffu : 0u) | (L.dy >= 0.0f ? 0x3ffu << 10 : 0u) | (L.dz >= 0.0f ? 0x3ffu << 20 : 0u);
    x.ex = (hi & x.pm) | (lo & ~x.pm);
    x.tied = (t == tx ? 0x3ffu : 0u) | (t == ty ? 0x3ffu << 10 : 0u) | (t == tz ? 0x3ffu << 20 : 0u);
    t += 0.0001f;
    float px, py, pz, i = 0.0f;
    x.esc = 0u;
    for (;;) { /* escape help, :233-242 */
        px = v.ox + L.dx * t; py = v.oy + L.dy * t; pz = v.oz + L.dz * t;
        if (!(lx <= px && px < hx && ly <= py && py < hy && lz <= pz && pz < hz)) break;
        t += 0.001f * i;
        i += 1.0f;
        x.esc++;
    }
    L.t = t;
    x.in_root = vr_root_contains(v, px, py, pz);
    if (!x.in_root) return x;
    const uint32_t c = vr_pack10((uint32_t)px, (uint32_t)py, (uint32_t)pz);
    L.c = c;
    uint32_t in3 = 0;
    for (uint32_t a = 0; a < 3u; a++) {
        const uint32_t ca = vr_wide_field(c, a * 10u);
        in3 |= (vr_wide_field(lo, a * 10u) <= ca && ca < vr_wide_field(hi, a * 10u) ? 1u : 0u) << a;
    }
    L.in3 = in3;
    L.k = L.d;
    return x;
}

/* vr_wide_pop for the union walk: undoes splits from the bottom until the box holds the new point, checking that
 * every box the point has left shares the front's exit plane (`suspect` otherwise), then takes the other child. */
template <class Path>
VR_HD bool vr_uw_pop(VrWideLane &L, Path &path, uint32_t max_trips, const VrUnionWideExit &x, bool &suspect, VrUnionFrom &from)
{
    uint32_t lo = L.lo, hi = L.hi, in3 = L.in3, k = L.k, e = 0;
    uint32_t left_lo = lo, left_hi = hi; /* the largest box left so far: it decides for all the boxes inside it (vr_union_pop) */
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
        /* the box of level k is left as well: the iso values parked there make their own MOVE out of it */
        if (in3 != 7u) { left_lo = lo; left_hi = hi; }
    }
    suspect = suspect || !vr_union_same_exit((left_hi & x.pm) | (left_lo & ~x.pm), x.ex, x.tied);
    L.lo = lo; L.hi = hi;
    if (k < L.d) {
        L.idx = ((L.idx + 1u) >> (L.d - k)) - 1u;
        from.node = L.idx; from.lo = lo; from.hi = hi; /* the ancestor that holds the new point */
        path.set(k, vr_wide_split(L, e & 0x3ffu, (e >> 10) & 3u, (e >> 12) & 31u));
        L.d = k + 1u;
    }
    return true;
}

/* the front has left the root box: every level of the path is left (vr_union_leave; root_hi = the root box's upper corner) */
template <class Path>
VR_HD void vr_uw_leave(VrWideLane &L, Path &path, const VrUnionWideExit &x, bool &suspect, uint32_t root_hi)
{
    (void)L; (void)path;
    suspect = suspect || !vr_union_same_exit(root_hi & x.pm, x.ex, x.tied);
}

template <class Fetch>
VR_HD void vr_uw_after_pop(VrUnionWideLane &U, const Fetch &fetch)
{
    const uint32_t n = fetch((U.L.idx - 1u) >> 1);
    U.prev_mn = n & 0xffu; U.prev_mx = (n >> 8) & 0xffu;
}

/* vr_union_resolve on a wide tree: the path to the node the front stood on (level depth_k, point cell `cell_before`) is
 * walked again from the root in registers; for every level above `a_level` the iso values parked there make their own
 * MOVE out of that level's box, which is compared with the front's state (t_new, cell_new, in_root_new). */
template <class Fetch, class Done, class Sink>
VR_HD void vr_uw_resolve(VrUnionWideLane &U, const VrViewK &v, const Fetch &fetch, Done &done, Sink &sink, uint32_t depth_k,
                         uint32_t cell_before, int a_level, float t_new, uint32_t cell_new, bool in_root_new, const VrUnionFrom &from)
{
    VrWideLane T = U.L; /* the ray; box, node and point are set below */
    const uint32_t m0 = a_level < 0 ? 0u : (uint32_t)a_level; /* the walk starts at the ancestor `from` (see vr_union_resolve) */
    T.idx = from.node; T.d = m0; T.lo = from.lo; T.hi = from.hi; T.c = cell_before;
    uint32_t prev_mn = 0u, prev_mx = 255u;
    bool marked = false;
    for (uint32_t m = m0; m <= depth_k; m++) {
        const uint32_t n = fetch(T.idx);
        const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        if ((int)m > a_level) {
            bool worked_out = false, same = false;
            VrUnionWideExit own;
            own.esc = 0u; own.in_root = false; own.ex = 0u; own.tied = 0u; own.pm = 0u;
            float own_t = 0.0f;
            uint32_t own_cell = 0u;
            auto each = [&](uint32_t r0, uint32_t r1) {
                if (!worked_out) {
                    VrWideLane O = T; /* the MOVE writes the point: on a copy, the walk down needs cell_before */
                    own = vr_uw_move(O, v);
                    own_t = O.t; own_cell = O.c;
                    same = own.in_root == in_root_new && (!own.in_root || (vr_f32_bits(own_t) == vr_f32_bits(t_new) && own_cell == cell_new));
                    worked_out = true;
                }
                if (!same) { sink.except(r0, r1, own_t, own_cell, own.in_root, own.esc); marked = true; }
                else if (own.esc) sink.esc(r0, r1, own.esc);
            };
            uint32_t lo0 = 1u, hi0 = 0u, lo1 = 1u, hi1 = 0u;
            if (mn > prev_mn) { lo0 = prev_mn > U.alive_lo ? prev_mn : U.alive_lo; hi0 = mn - 1u < U.alive_hi ? mn - 1u : U.alive_hi; }
            if (mx < prev_mx) { lo1 = mx + 1u > U.alive_lo ? mx + 1u : U.alive_lo; hi1 = prev_mx < U.alive_hi ? prev_mx : U.alive_hi; }
            if (lo0 <= hi0) vr_union_runs(done, lo0, hi0, each, U.holes);
            if (lo1 <= hi1) vr_union_runs(done, lo1, hi1, each, U.holes);
            if (worked_out && !same) {
                vr_union_need_bits(done, U);
                if (lo0 <= hi0) vr_union_mark(done, lo0, hi0);
                if (lo1 <= hi1) vr_union_mark(done, lo1, hi1);
            }
        }
        if (m == depth_k) break;
        (void)vr_wide_split(T, VR_WIDE_PART(n), VR_WIDE_AXIS(n), VR_WIDE_AXIS(n) * 10u);
        T.d++;
        prev_mn = mn; prev_mx = mx;
    }
    if (marked) vr_union_shrink(done, U);
}

#endif /* VR_TRACE_UNION_WIDE_H */
