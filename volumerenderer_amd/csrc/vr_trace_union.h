/*
 * vr_trace_union.h -- lane code of the union walk: ONE traversal per ray for a whole range of iso values.
 *
 * The reference traces a ray once per iso value (src/raytrace.zig:117-255, called per iso value by
 * src/main.zig:111-138).  The state of such a walk after a MOVE -- t and with it the point -- depends only on the
 * box it left (src/raytrace.zig:199-246): t = the nearest exit plane of that box + 0.0001, then escape help against
 * that box.  Walks of different iso values part at a node (one is excluded there and leaves the node's box at once,
 * the other goes inside and leaves smaller boxes first) and meet again where the ray leaves that node's box: the last
 * box left inside shares the exit plane, the division has the same operands, t has the same bits.  So ONE front can
 * visit every state that the walk of any iso value visits, each once:
 *
 *   - the front goes into a node while any iso value that is not done yet holds there (min <= iso <= max);
 *   - the iso values present at a node that its range excludes are "parked": the reference has checked depth + 1
 *     nodes for them in this round (src/raytrace.zig:162) and MOVEs them out of this node's box; they are present
 *     again where the front leaves that box;
 *   - at a leaf every present iso value in the leaf's range hits, at the front's t and cell (src/raytrace.zig:178-187);
 *   - present at a state whose lowest ancestor holding the point is A: every iso value in A's range that is not
 *     done (node ranges are nested along a path, so this is what the parked sets add up to);
 *   - when the front leaves a box B_k and with it the boxes of ancestors B_m, the iso values parked at level m make
 *     their own MOVE out of B_m.  It lands on the front's state if B_m and B_k share the exit plane on the axis that
 *     gave the front's t and the front needed no escape help (then the point is outside B_m as well, so neither do
 *     they).  Anything else is SUSPECT and resolved exactly (vr_union_resolve): the own MOVE of every parked set is
 *     worked out and compared; iso values whose state differs leave the union walk as an exception and are walked
 *     on their own from their own state.
 *
 * Per-iso counters are sums over the events an iso value was present at, so they go to the sink as (run of iso
 * values, amount) pairs; a run is a maximal stretch of iso values that are not done.
 *
 * Policies (template parameters): Fetch = node read by heap position (prepared array, vr_trace_core.h), Stack = the
 * path stack cursor of vr_trace_core.h, Done = 256 done bits as eight words (word, set, unset, clear, fill), Sink = where
 * events go: park(r0, r1, checks), hit(r0, r1, checks, t, cell, g), miss(r0, r1), esc(r0, r1, n),
 * except(r0, r1, t, cell, in_root, esc).
 */
#ifndef VR_TRACE_UNION_H
#define VR_TRACE_UNION_H

#include "vr_trace_core.h"

/* The lane code of a front is large (runs of iso values, both sides of a range, exact resolution): the level loop is
 * not unrolled, so that the kernel's loop stays inside the instruction cache.  (Nothing is called out of line: a call
 * makes the kernel use scratch memory, and with scratch the whole grid is no longer resident at once.) */
/* the measuring build counts how often a wave runs through a piece of the lane code (VR_WC(i), vr_measure.h); everywhere
 * else -- the product library, the CPU emulation of the tests -- it is nothing */
#ifndef VR_WC
#define VR_WC(i) do { } while (0)
#endif

#if defined(__clang__)
#define VR_ROLLED _Pragma("unroll 1")
#else
#define VR_ROLLED
#endif

struct VrUnionLane {
    VrLane L;                    /* ray, point, current node and its box (checked is not used) */
    uint32_t prev_mn, prev_mx;   /* range of the parent of the current node ([0, 255] above the root) */
    uint32_t alive_lo, alive_hi; /* smallest / largest iso value not done; lo > hi: none left */
    bool holes;                  /* a done iso value may lie between them (never while the hits have come off the ends);
                                  * the done bits are kept only from then on, and only between alive_lo and alive_hi */
};

VR_HD uint32_t vr_f32_bits(float x) { uint32_t b; __builtin_memcpy(&b, &x, 4); return b; }
VR_HD uint32_t vr_ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
VR_HD uint32_t vr_clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }

/* calls f(r0, r1) for every maximal run of iso values in [p, q] that are not done */
template <class Done, class F>
VR_HD void vr_union_runs(const Done &done, uint32_t p, uint32_t q, F f, bool holes)
{
    if (!holes) { f(p, q); return; } /* p, q are clipped to [alive_lo, alive_hi] by every caller */
    uint32_t i = p;
    while (i <= q) {
        VR_WC(3);
        /* next iso value >= i that is not done */
        uint32_t w = i >> 5;
        uint32_t bits = ~done.word(w) >> (i & 31u);
        if (bits == 0u) { i = (w + 1u) << 5; continue; }
        i += vr_ctz32(bits);
        if (i > q) break;
        /* the run goes on to the next done iso value */
        uint32_t j = i;
        for (;;) {
            w = j >> 5;
            bits = done.word(w) >> (j & 31u);
            if (bits != 0u) { j += vr_ctz32(bits); break; }
            j = (w + 1u) << 5;
            if (j > q) break;
        }
        const uint32_t r1 = j - 1u < q ? j - 1u : q;
        f(i, r1);
        i = r1 + 2u; /* r1 + 1 is done or beyond q */
    }
}

/* marks [r0, r1] done */
template <class Done>
VR_HD void vr_union_mark(Done &done, uint32_t r0, uint32_t r1)
{
    for (uint32_t w = r0 >> 5; w <= (r1 >> 5); w++) {
        const uint32_t a = w == (r0 >> 5) ? (r0 & 31u) : 0u, b = w == (r1 >> 5) ? (r1 & 31u) : 31u;
        const uint32_t m = (0xffffffffu >> (31u - b)) & (0xffffffffu << a);
        done.set(w, m);
    }
}

/* is any iso value in [mn, mx] not done? */
template <class Done, class ULane>
VR_HD bool vr_union_any(const Done &done, const ULane &U, uint32_t mn, uint32_t mx, bool holes)
{
    if (U.alive_lo > U.alive_hi || mx < U.alive_lo || mn > U.alive_hi) return false;
    if (!holes || (mn <= U.alive_lo && U.alive_lo <= mx) || (mn <= U.alive_hi && U.alive_hi <= mx)) return true;
    /* strictly inside (alive_lo, alive_hi) here: the bits are valid */
    for (uint32_t w = mn >> 5; w <= (mx >> 5); w++) {
        VR_WC(4);
        const uint32_t a = w == (mn >> 5) ? (mn & 31u) : 0u, b = w == (mx >> 5) ? (mx & 31u) : 31u;
        const uint32_t m = (0xffffffffu >> (31u - b)) & (0xffffffffu << a);
        if (~done.word(w) & m) return true;
    }
    return false;
}

/* The first done iso value strictly inside the alive range is about to be marked: from here on the bits count. */
template <class Done, class ULane>
VR_HD void vr_union_need_bits(Done &done, ULane &U)
{
    if (U.holes) return;
    for (uint32_t w = 0; w < 8u; w++) done.clear(w);
    U.holes = true;
}

/* after marking: pulls alive_lo / alive_hi in to the nearest iso values that are not done; when no done iso value is left
 * between them -- the hole has reached an end of the range -- the two bounds say it all again and the bits are dropped */
template <class Done, class ULane>
VR_HD void vr_union_shrink(const Done &done, ULane &U)
{
    while (U.alive_lo <= U.alive_hi) {
        const uint32_t w = U.alive_lo >> 5, bits = ~done.word(w) >> (U.alive_lo & 31u);
        if (bits) { U.alive_lo += vr_ctz32(bits); break; }
        U.alive_lo = (w + 1u) << 5;
    }
    while (U.alive_lo <= U.alive_hi) {
        const uint32_t w = U.alive_hi >> 5, bits = ~done.word(w) << (31u - (U.alive_hi & 31u));
        if (bits) { U.alive_hi -= vr_clz32(bits); break; }
        if (w == 0u) { U.alive_lo = 1u; U.alive_hi = 0u; break; }
        U.alive_hi = (w << 5) - 1u;
    }
#ifndef VR_UNION_KEEP_HOLES
    if (U.holes && U.alive_lo <= U.alive_hi) {
        uint32_t any = 0u;
        for (uint32_t w = U.alive_lo >> 5; w <= (U.alive_hi >> 5); w++) {
            const uint32_t a = w == (U.alive_lo >> 5) ? (U.alive_lo & 31u) : 0u, b = w == (U.alive_hi >> 5) ? (U.alive_hi & 31u) : 31u;
            any |= done.word(w) & (0xffffffffu >> (31u - b)) & (0xffffffffu << a);
        }
        if (any == 0u) U.holes = false;
    }
#endif
}

template <class Stack, class Done>
VR_HD void vr_union_begin(VrUnionLane &U, Stack &stack, Done &done, const VrViewK &v, float dx, float dy, float dz, float t,
                          uint32_t iso_begin, uint32_t iso_count)
{
    vr_lane_begin(U.L, stack, v, dx, dy, dz, t);
    U.prev_mn = 0u; U.prev_mx = 255u;
    U.alive_lo = iso_begin; U.alive_hi = iso_begin + iso_count - 1u;
    U.holes = false;
    (void)done;
}

/* Up to four runs of iso values, in any order, as byte pairs (lo, hi) in two words: what has left a front at one state
 * (an excluding node's range has a side below and one above; ancestors that share its exit plane add theirs). */
struct VrUnionRuns {
    uint32_t n, w01, w23;
    VR_HD void clear() { n = 0u; w01 = 0u; w23 = 0u; }
    VR_HD uint32_t lo(uint32_t i) const { return ((i < 2u ? w01 : w23) >> ((i & 1u) * 16u)) & 0xffu; }
    VR_HD uint32_t hi(uint32_t i) const { return ((i < 2u ? w01 : w23) >> ((i & 1u) * 16u + 8u)) & 0xffu; }
    VR_HD bool room() const { return n < 4u; }
    VR_HD void add(uint32_t r0, uint32_t r1)
    {
        const uint32_t pair = (r0 | (r1 << 8)) << ((n & 1u) * 16u);
        if (n < 2u) w01 |= pair; else w23 |= pair;
        n++;
    }
};

/* A new front for the runs `runs` of iso values of the same ray from the state (t, cell): what iso values that left the
 * union walk carry on with (their own MOVE has been made and counted; the reference returns to the root next). */
/* the iso values of a front that starts at the root: the runs `runs`, everything else done */
template <class Done, class ULane>
VR_HD void vr_union_set_runs(ULane &U, Done &done, const VrUnionRuns &runs)
{
    U.prev_mn = 0u; U.prev_mx = 255u;
    U.alive_lo = 255u; U.alive_hi = 0u;
    U.holes = runs.n > 1u;
    if (!U.holes) { U.alive_lo = runs.lo(0u); U.alive_hi = runs.hi(0u); return; }
    for (uint32_t w = 0; w < 8u; w++) done.fill(w); /* everything done but the runs */
    for (uint32_t i = 0; i < runs.n; i++) {
        const uint32_t r0 = runs.lo(i), r1 = runs.hi(i);
        U.alive_lo = r0 < U.alive_lo ? r0 : U.alive_lo;
        U.alive_hi = r1 > U.alive_hi ? r1 : U.alive_hi;
        for (uint32_t w = r0 >> 5; w <= (r1 >> 5); w++) {
            const uint32_t a = w == (r0 >> 5) ? (r0 & 31u) : 0u, b = w == (r1 >> 5) ? (r1 & 31u) : 31u;
            done.unset(w, (0xffffffffu >> (31u - b)) & (0xffffffffu << a));
        }
    }
}

template <class Stack, class Done>
VR_HD void vr_union_restart(VrUnionLane &U, Stack &stack, Done &done, const VrViewK &v, float t, uint32_t cell, const VrUnionRuns &runs)
{
    vr_lane_restart(U.L, stack, v, t, cell);
    vr_union_set_runs(U, done, runs);
}

/* Goes down from the current node while an iso value that is not done holds, at most max_trips levels.  Returns
 * VR_GO_MORE (budget used up) or VR_GO_EXCLUDED (nobody goes into the current node, or it was a leaf and its iso
 * values have hit: the front leaves its box next). */
/* HOLES = false: the caller knows that no lane that goes down here has a done iso value inside its alive range (U.holes is
 * false for all of them), which is the rule: the code that goes through the done bits is then left out of the loop. */
template <bool HOLES, class Fetch, class Stack, class Done, class Sink>
VR_HD uint32_t vr_union_descend(VrUnionLane &U, const Fetch &fetch, Stack &stack, Done &done, Sink &sink, uint32_t max_trips)
{
    VrLane &L = U.L;
    const bool holes = HOLES ? U.holes : false; /* stays as it is until a leaf ends the descent */
    VR_ROLLED
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        const uint32_t n = fetch(L.g);
        const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        const uint32_t checks = stack.depth() + 1u; /* root .. this node (src/raytrace.zig:162) */
        VR_WC(0);
        /* present: the parent's range; excluded here: what this node's range leaves out on either side */
        if (mn > U.prev_mn) {
            const uint32_t p = U.prev_mn > U.alive_lo ? U.prev_mn : U.alive_lo, q = mn - 1u < U.alive_hi ? mn - 1u : U.alive_hi;
            if (p <= q) { VR_WC(1); vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.park(r0, r1, checks); }, holes); }
        }
        if (mx < U.prev_mx) {
            const uint32_t p = mx + 1u > U.alive_lo ? mx + 1u : U.alive_lo, q = U.prev_mx < U.alive_hi ? U.prev_mx : U.alive_hi;
            if (p <= q) { VR_WC(2); vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.park(r0, r1, checks); }, holes); }
        }
        if (!vr_union_any(done, U, mn, mx, holes)) return VR_GO_EXCLUDED;
        if (n & VR_N_LEAF) {
            VR_WC(5);
            const uint32_t p = mn > U.alive_lo ? mn : U.alive_lo, q = mx < U.alive_hi ? mx : U.alive_hi;
            vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.hit(r0, r1, checks, L.t, L.cell, L.g); }, holes);
            if (!holes && (p == U.alive_lo || q == U.alive_hi)) { /* hits off either end leave no hole */
                if (p == U.alive_lo) U.alive_lo = q + 1u; /* past alive_hi when both ends went: none left */
                else U.alive_hi = p - 1u;
            } else {
                vr_union_need_bits(done, U);
                vr_union_mark(done, p, q);
                vr_union_shrink(done, U);
            }
            return VR_GO_EXCLUDED;
        }
        const uint32_t sh = (n >> 24) & 31u, part = (n >> 16) & 0xffu;
        const bool right = ((L.cell >> sh) & 0xffu) >= part;
        uint32_t side = right ? L.lo : L.hi;
        stack.push(side);
        side = (side & ~(0xffu << sh)) | (part << sh);
        L.lo = right ? side : L.lo;
        L.hi = right ? L.hi : side;
        L.g = vr_twice_plus(L.g, right);
        U.prev_mn = mn; U.prev_mx = mx;
    }
    return VR_GO_MORE;
}

/* After vr_union_pop has entered the other child of the ancestor that holds the point: the present iso values are the
 * ones in that ancestor's range. */
template <class Fetch>
VR_HD void vr_union_after_pop(VrUnionLane &U, const Fetch &fetch)
{
    const uint32_t n = fetch(U.L.g >> 1);
    U.prev_mn = n & 0xffu; U.prev_mx = (n >> 8) & 0xffu;
}

/* The escape helps of a MOVE whose popped boxes all share the front's exit plane.  Every iso value that made the MOVE --
 * the ones not done in [mn, mx], the range of the ancestor the front came to ([0, 255] when it left the volume) -- started
 * from the same t (same plane, same division) and ran the same escape loop: the loop of a popped box B_m goes on while
 * the point is inside B_m, the front's while it is inside its own box B_k, a part of B_m; the front's last point is
 * outside B_m (that is why B_m was popped), and a point outside B_m is outside B_k, so both stop at the same step. */
template <class Done, class ULane, class Sink>
VR_HD void vr_union_credit_esc(const Done &done, const ULane &U, Sink &sink, uint32_t mn, uint32_t mx, uint32_t esc)
{
    if (esc == 0u) return;
    const uint32_t p = mn > U.alive_lo ? mn : U.alive_lo, q = mx < U.alive_hi ? mx : U.alive_hi;
    if (p <= q) vr_union_runs(done, p, q, [&](uint32_t r0, uint32_t r1) { sink.esc(r0, r1, esc); }, U.holes);
}

/* vr_lane_move (MOVE + escape help, src/raytrace.zig:199-242) that also says which planes gave t: `ex` = the exit
 * planes of the box that was left (max where the ray travels in +, else min), `tied` = all ones in the field of every
 * axis whose plane gave exactly the smallest t (one axis as a rule; more where the ray goes through an edge of the cell
 * lattice, which rays from a camera at integer coordinates do all the time). */
struct VrUnionExit { uint32_t ex, tied, esc; bool in_root; };

/* Does the MOVE out of a larger box -- exit planes ex_a, each the same as the left box's or farther along the ray --
 * start from the same t?  Its t is the smallest over the planes; those it shares with the left box give the left box's
 * values, the others give larger ones.  So it does iff it shares the plane of an axis that gave the left box's t:
 * a zero field of (ex_a ^ ex) among the tied ones. */
VR_HD bool vr_union_same_exit(uint32_t ex_a, const uint32_t ex, const uint32_t tied)
{
    const uint32_t e = (ex_a ^ ex) | ~tied;
    return ((e - VR_F_ONES) & ~e & (VR_F_ONES << 9)) != 0u;
}

/* A node of the front's path and its box: where the exact resolution of a MOVE starts its walk down the old path (the
 * lowest ancestor whose box still holds the new point; the root when the front has left the volume). */
struct VrUnionFrom { uint32_t node, lo, hi; };

VR_HD VrUnionExit vr_union_move(VrLane &L, const VrViewK &v)
{
    VrUnionExit x;
    x.ex = (L.hi & L.posmask) | (L.lo & ~L.posmask);
    const float tx = vr_plane_t_r((float)(x.ex & 0xffu), v.ox, L.dx, L.rx);
    const float ty = vr_plane_t_r((float)((x.ex >> 10) & 0xffu), v.oy, L.dy, L.ry);
    const float tz = vr_plane_t_r((float)((x.ex >> 20) & 0xffu), v.oz, L.dz, L.rz);
    float t = fminf(fminf(tx, ty), tz);
    x.tied = (t == tx ? 0x3ffu : 0u) | (t == ty ? 0x3ffu << 10 : 0u) | (t == tz ? 0x3ffu << 20 : 0u);
    t += 0.0001f;
    float i = 0.0f;
    x.esc = 0u;
    VR_WC(7);
    for (;;) {
        VR_WC(8);
        const float px = v.ox + L.dx * t, py = v.oy + L.dy * t, pz = v.oz + L.dz * t;
        const uint32_t cx = (uint32_t)vr_floor_i32(px), cy = (uint32_t)vr_floor_i32(py), cz = (uint32_t)vr_floor_i32(pz);
        x.in_root = (cx < v.rdx) & (cy < v.rdy) & (cz < v.rdz);
        if (!x.in_root) break;
        L.cell = vr_pack10(cx, cy, cz);
        if (!vr_box_contains_cell(L.lo, L.hi, L.cell)) break;
        t += 0.001f * i;
        i += 1.0f;
        x.esc++;
    }
    L.t = t;
    return x;
}

/* vr_lane_pop for the union walk: pops ancestors until one holds the new point and steps into that ancestor's other
 * child (see vr_lane_pop), checking on the way that every box the point has left shares the front's exit plane
 * (`suspect` is set otherwise).  At most max_trips levels per call; true when the other child was entered. */
template <class Stack>
VR_HD bool vr_union_pop(VrLane &L, Stack &stack, uint32_t max_trips, const VrUnionExit &x, bool &suspect, VrUnionFrom &from)
{
    const uint32_t cg = L.cell | VR_F_GUARD, cp = L.cell + VR_F_ONES;
    uint32_t next = stack.peek_up(1);
    uint32_t word = 0, a_lo = 0, a_hi = 0;
    bool left = false, found = false;
    for (uint32_t trip = 0; trip < max_trips; trip++) {
        VR_WC(11);
        word = next;
        stack.up();
        next = stack.peek_up(1);
        left = (L.g & 1u) == 0u;
        a_lo = left ? L.lo : word; a_hi = left ? word : L.hi;
        const uint32_t a = cg - a_lo, b = (a_hi | VR_F_GUARD) - cp;
        found = ((a & b) & VR_F_GUARD) == VR_F_GUARD;
        if (found) break;
        /* the parent's box is left as well: the iso values parked there make their own MOVE out of it */
        L.lo = a_lo; L.hi = a_hi;
        L.g >>= 1;
    }
    if (!found) return false;
    /* Every box that was left must share the front's exit plane.  Along an axis the exit plane of a box lies between its
     * child's and its parent's, so the largest box left -- the one the lane stands on now, a child of the ancestor that holds
     * the point -- decides for all the boxes below it. */
    suspect = suspect || !vr_union_same_exit((L.hi & L.posmask) | (L.lo & ~L.posmask), x.ex, x.tied);
    from.node = L.g >> 1; from.lo = a_lo; from.hi = a_hi; /* the ancestor that holds the new point */
    const uint32_t cut = left ? L.hi : L.lo, keep = left ? L.lo : L.hi;
    const uint32_t carry = ((word ^ cut) + 0xffu * VR_F_ONES) & VR_F_GUARD;
    const uint32_t fm = carry - (carry >> 8);
    const uint32_t other = (keep & ~fm) | (cut & fm);
    stack.push(keep);
    L.lo = left ? other : word;
    L.hi = left ? word : other;
    L.g ^= 1u;
    return true;
}

/* The front has left the root box: every level of the path is left.  Sets `suspect` if any box on the way up does not
 * share the front's exit plane.  Along an axis the exit plane of a box lies between its child's and its parent's, so the
 * root box (root_hi: its upper corner) decides for every box below it, and nothing is walked (the path store keeps its
 * levels; the next front starts it afresh).  The rule is "not suspect": a ray leaves the volume through the face its last
 * box touches. */
template <class Stack>
VR_HD void vr_union_leave(VrLane &L, Stack &stack, const VrUnionExit &x, bool &suspect, uint32_t root_hi)
{
    (void)stack;
    suspect = suspect || !vr_union_same_exit(root_hi & L.posmask, x.ex, x.tied);
}

/* Exact resolution of a suspect MOVE.  The front stood on the node at heap position g_k (level depth_k) with point
 * cell `cell_before`, left its box and came to the state (t_new, cell_new, in_root_new) with `esc_k` escape helps; its
 * lowest ancestor that holds the new point is `from`, at level `a_level` (all levels left: a_level = -1, from = the
 * root).  For every level m in (a_level, depth_k] the iso values parked there (present at m - 1, excluded at m, not
 * done) make their own MOVE out of the box of level m: same state as the front's -> they get its escape helps, if any;
 * another state -> they leave the union walk (sink.except) and are marked done.  Walks the old path again from that
 * ancestor down, in registers (a chain of dependent node reads: from the root it was four times as long). */
template <class Fetch, class Done, class Sink>
VR_HD void vr_union_resolve(VrUnionLane &U, const VrViewK &v, const Fetch &fetch, Done &done, Sink &sink, uint32_t depth_k,
                            uint32_t cell_before, int a_level, float t_new, uint32_t cell_new, bool in_root_new,
                            const VrUnionFrom &from)
{
    VrLane T = U.L; /* the ray; box, node and point are set below */
    uint32_t g = from.node, lo = from.lo, hi = from.hi, prev_mn = 0u, prev_mx = 255u;
    bool marked = false;
    for (uint32_t m = a_level < 0 ? 0u : (uint32_t)a_level; m <= depth_k; m++) {
        VR_WC(10);
        const uint32_t n = fetch(g);
        const uint32_t mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        if ((int)m > a_level) {
            T.lo = lo; T.hi = hi; T.cell = cell_before;
            bool worked_out = false, same = false;
            VrUnionExit own;
            own.esc = 0u; own.in_root = false;
            float own_t = 0.0f;
            uint32_t own_cell = 0u;
            auto each = [&](uint32_t r0, uint32_t r1) {
                if (!worked_out) {
                    own = vr_union_move(T, v);
                    own_t = T.t; own_cell = T.cell;
                    same = own.in_root == in_root_new &&
                           (!own.in_root || (vr_f32_bits(own_t) == vr_f32_bits(t_new) && own_cell == cell_new));
                    worked_out = true;
                }
                if (!same) { sink.except(r0, r1, own_t, own_cell, own.in_root, own.esc); marked = true; }
                else if (own.esc) sink.esc(r0, r1, own.esc);
            };
            /* runs are taken before anything is marked at this level: marking happens after both sides (scalars, not
             * arrays: an indexed local array would live in scratch memory) */
            uint32_t lo0 = 1u, hi0 = 0u, lo1 = 1u, hi1 = 0u;
            if (mn > prev_mn) { lo0 = prev_mn > U.alive_lo ? prev_mn : U.alive_lo; hi0 = mn - 1u < U.alive_hi ? mn - 1u : U.alive_hi; }
            if (mx < prev_mx) { lo1 = mx + 1u > U.alive_lo ? mx + 1u : U.alive_lo; hi1 = prev_mx < U.alive_hi ? prev_mx : U.alive_hi; }
            if (lo0 <= hi0) vr_union_runs(done, lo0, hi0, each, U.holes);
            if (lo1 <= hi1) vr_union_runs(done, lo1, hi1, each, U.holes);
            if (worked_out && !same) {
                vr_union_need_bits(done, U);
                if (lo0 <= hi0) vr_union_mark(done, lo0, hi0); /* already done ones stay done */
                if (lo1 <= hi1) vr_union_mark(done, lo1, hi1);
            }
        }
        if (m == depth_k) break;
        const uint32_t sh = (n >> 24) & 31u, part = (n >> 16) & 0xffu;
        const bool right = ((cell_before >> sh) & 0xffu) >= part;
        uint32_t side = right ? lo : hi;
        side = (side & ~(0xffu << sh)) | (part << sh);
        lo = right ? side : lo;
        hi = right ? hi : side;
        g = vr_twice_plus(g, right);
        prev_mn = mn; prev_mx = mx;
    }
    if (marked) vr_union_shrink(done, U);
}

#endif /* VR_TRACE_UNION_H */
