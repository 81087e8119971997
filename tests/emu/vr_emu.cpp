/*
 * vr_emu.cpp -- TEST HARNESS: runs the lane code of volumerenderer_amd/csrc/vr_trace_core.h on the
 * CPU, one pixel at a time, so the traversal variants can be checked against the oracle without a
 * GPU.  Not part of the product library and never loaded by it.
 */
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../../volumerenderer_amd/csrc/vr_trace_core.h"

namespace {
struct HostFetch {
    const uint32_t *nodes;
    uint32_t operator()(uint32_t i) const { return nodes[i]; }
};
struct HostStack {
    uint32_t word[VR_MAX_DEPTH + 3];
    int cur = 1; /* one spare slot above the root, as on the device */
    void reset() { cur = 1; }
    void push(uint32_t w) { word[cur++] = w; }
    void up() { cur--; }
    uint32_t peek_up(uint32_t k) const { return word[cur - (int)k]; }
    uint32_t depth() const { return (uint32_t)(cur - 1); }
};
} // namespace

/* variant 1 walks the reference's SlimNode array with the literal restart loop; any other variant walks
 * the PREPARED array (vr_host_prepare_kdt of the product library) with the resume traversal; variant
 * 100 + k bounds every descend / pop phase to k levels, as the persistent kernel does */
extern "C" int vr_emu_render(int variant, const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3],
                             const float src[3], const float forw[3], const float up[3], uint32_t width,
                             uint32_t height, float viewport_width, uint32_t iso, uint8_t *rgb, uint32_t *cell_flag,
                             float *t_out, uint32_t *checked, uint32_t *node, uint64_t *sums /*7*/)
{
    VrViewK k;
    k.ox = src[0]; k.oy = src[1]; k.oz = src[2];
    k.fx = forw[0]; k.fy = forw[1]; k.fz = forw[2];
    k.ux = up[0]; k.uy = up[1]; k.uz = up[2];
    k.rx = k.fy * k.uz - k.fz * k.uy;
    k.ry = k.fz * k.ux - k.fx * k.uz;
    k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = viewport_width / (float)width;
    k.half_w = (float)(width / 2);
    k.half_h = (float)(height / 2);
    k.width = width; k.height = height;
    k.rdx = root_dim[0]; k.rdy = root_dim[1]; k.rdz = root_dim[2];
    k.root_hi = vr_pack10(k.rdx, k.rdy, k.rdz);
    k.n_slots = n_slots;
    std::memset(sums, 0, 7 * sizeof(uint64_t));
    for (uint32_t yi = 0; yi < height; yi++)
        for (uint32_t xi = 0; xi < width; xi++) {
            VrRay ray = vr_make_ray(k, xi, yi);
            VrHit h;
            if (variant == 1) {
                h = vr_trace_restart(k, ray, iso, HostFetch{ nodes });
            } else {
                HostStack st;
                h = vr_trace_resume(k, ray, iso, HostFetch{ nodes }, st, variant >= 100 ? (uint32_t)(variant - 100) : 64u);
            }
            uint64_t pix = (uint64_t)yi * width + xi;
            uint8_t g = h.flag == 1u ? vr_shade(h.t) : 0;
            rgb[pix * 3] = rgb[pix * 3 + 1] = rgb[pix * 3 + 2] = g;
            cell_flag[pix] = h.cell | (h.flag << 24);
            t_out[pix] = h.flag == 2u ? 0.0f : h.t;
            checked[pix] = h.checked;
            node[pix] = h.node;
            sums[0] += h.checked;
            sums[1] += h.flag == 1u;
            sums[2] += h.flag == 0u;
            sums[3] += h.flag == 2u;
            sums[4] += h.escapes;
            sums[5] += h.moves;
            sums[6] += h.error;
        }
    return 0;
}

/* vr_div (reciprocal + two FMA corrections) against the IEEE divide on the operand range of the MOVE step:
 * numerator = plane offset (0..255) - camera coordinate, denominator = a component of a unit direction with
 * |d| >= 1e-8.  Returns the number of operand pairs whose results differ in any bit. */
static uint64_t div_check(uint64_t samples, uint64_t seed, uint32_t n_offsets);
extern "C" uint64_t vr_emu_div_check(uint64_t samples, uint64_t seed) { return div_check(samples, seed, 256u); }
/* the same with plane offsets 0..1023: volumes of up to 1024 voxels per axis */
extern "C" uint64_t vr_emu_div_check_wide(uint64_t samples, uint64_t seed) { return div_check(samples, seed, 1024u); }
static uint64_t div_check(uint64_t samples, uint64_t seed, uint32_t n_offsets)
{
    uint64_t s = seed ? seed : 88172645463325252ull, bad = 0;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto urand = [&]() { return (float)((rnd() >> 40) * (1.0 / 16777216.0)); };
    for (uint64_t i = 0; i < samples; i++) {
        float d;
        const uint32_t mode = (uint32_t)(rnd() & 3u);
        if (mode == 0) d = urand() * 2.0f - 1.0f;
        else if (mode == 1) d = ldexpf(urand() + 0.5f, -(int)(rnd() % 27u)) * ((rnd() & 1u) ? 1.0f : -1.0f);
        else { uint32_t b = 0x3f800000u - (uint32_t)(rnd() % (27u << 23)); std::memcpy(&d, &b, 4); if (rnd() & 1u) d = -d; }
        if (!(fabsf(d) >= 1e-8f) || fabsf(d) > 1.0f) continue;
        float o;
        const uint32_t om = (uint32_t)(rnd() % 3u);
        if (om == 0) o = (urand() * 2.0f - 1.0f) * (n_offsets > 256u ? 3000.0f : 1000.0f);
        else if (om == 1) o = (float)((int)(rnd() % 2001u) - 1000) * 0.5f;
        else o = (urand() * 2.0f - 1.0f) * 300.0f;
        const float n = (float)(rnd() % n_offsets) - o;
        const float q = n / d, m = vr_div(n, d, 1.0f / d);
        if (std::memcmp(&q, &m, 4) != 0 && !(q == 0.0f && m == 0.0f)) bad++;
    }
    return bad;
}

/* The iso sweep with traversals shared between iso values, as the sweep kernel does it, one pixel at a
 * time: trace the lowest open iso value while tracking the largest iso value with the same walk, give
 * the result to the whole interval, go on behind it.  `chunk` cuts the iso range into independent
 * pieces as the kernel does for load balance (0 = one piece).  Outputs are [iso_count][height][width].
 *
 * `share` > 0 adds shared beginnings.  The walk for the next iso value, upper + 1, makes the same tests with
 * the same outcomes as this one up to the first node whose bound equals `upper`; the reference returns to the root
 * after every MOVE, so everything before the last such return ahead of that node -- ahead of the stretch of the walk
 * in which `upper` last went down -- is shared, and the next walk can start from the lane's state at that return: t,
 * cell, checks, moves and escape helps so far, and `upper` as it stood.  `share` such states are kept, newest on top:
 * the older ones lie in the shared part, where they are the next walk's own.
 * `ahead` != 0: descents read both children of a node ahead of its test (vr_lane_descend_ahead).
 * counts_out: [0] walks, [1] node reads, [2] the walks that carried on from a kept state, [3] returns to the root
 * walked (MOVEs that stayed inside the volume), [4] distinct ones among them per ray (by the bits of t), summed over the rays. */
namespace {
struct CountingFetch {
    const uint32_t *nodes;
    uint64_t *count;
    uint32_t last_pair;
    uint32_t operator()(uint32_t i) const { ++*count; return nodes[i]; }
    VrNodePair children(uint32_t g) const
    {
        uint32_t c = g << 1;
        c = c < last_pair ? c : last_pair;
        ++*count;
        return VrNodePair{ nodes[c], nodes[c + 1] };
    }
};
} // namespace

extern "C" int vr_emu_sweep(const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3], const float src[3],
                                   const float forw[3], const float up[3], uint32_t width, uint32_t height, float viewport_width,
                                   uint32_t iso_begin, uint32_t iso_count, uint32_t chunk, uint32_t max_trips, uint32_t share, uint32_t ahead,
                                   uint8_t *gray, uint32_t *cell_flag, float *t_out, uint32_t *checked, uint32_t *node,
                                   uint64_t *sums /*[iso_count][7]*/, uint64_t *counts_out /*[3]*/)
{
    VrViewK k;
    k.ox = src[0]; k.oy = src[1]; k.oz = src[2];
    k.fx = forw[0]; k.fy = forw[1]; k.fz = forw[2];
    k.ux = up[0]; k.uy = up[1]; k.uz = up[2];
    k.rx = k.fy * k.uz - k.fz * k.uy;
    k.ry = k.fz * k.ux - k.fx * k.uz;
    k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = viewport_width / (float)width;
    k.half_w = (float)(width / 2);
    k.half_h = (float)(height / 2);
    k.width = width; k.height = height;
    k.rdx = root_dim[0]; k.rdy = root_dim[1]; k.rdz = root_dim[2];
    k.root_hi = vr_pack10(k.rdx, k.rdy, k.rdz);
    k.n_slots = n_slots;
    std::memset(sums, 0, (size_t)iso_count * 7 * sizeof(uint64_t));
    const uint64_t frame = (uint64_t)width * height;
    if (chunk == 0) chunk = iso_count;
    uint64_t f_test = 0, n_resumed = 0, n_fresh = 0, n_segments = 0, n_states = 0;
    std::vector<uint32_t> seen; /* t bit patterns of the returns to the root of one ray, over all its walks */
    for (uint32_t yi = 0; yi < height; yi++)
        for (uint32_t xi = 0; xi < width; xi++) {
            const uint64_t pix = (uint64_t)yi * width + xi;
            VrRay ray = vr_make_ray(k, xi, yi);
            if (ray.oob) {
                for (uint32_t j = 0; j < iso_count; j++) {
                    gray[j * frame + pix] = 0; cell_flag[j * frame + pix] = 2u << 24; t_out[j * frame + pix] = 0.0f;
                    checked[j * frame + pix] = 0; node[j * frame + pix] = 0; sums[j * 7 + 3]++;
                }
                continue;
            }
            HostStack st;
            VrLane L;
            vr_lane_begin(L, st, k, ray.dx, ray.dy, ray.dz, ray.t);
            const uint32_t cell0 = L.cell;
            seen.clear();
            for (uint32_t c0 = 0; c0 < iso_count; c0 += chunk) {
                const uint32_t last = iso_begin + (c0 + chunk < iso_count ? c0 + chunk : iso_count) - 1u;
                struct Kept { uint32_t cell, checked, moves, esc, upper; float t; };
                Kept kept[64];
                uint32_t n_kept = 0;
                for (uint32_t iso = iso_begin + c0; iso <= last;) {
                    VrIsoUpper track;
                    uint32_t moves = 0, esc = 0, flag = 0;
                    vr_lane_restart(L, st, k, ray.t, cell0);
                    track.upper = 255u;
                    if (share && n_kept) {
                        const Kept s = kept[--n_kept];
                        L.t = s.t; L.cell = s.cell; L.checked = s.checked;
                        moves = s.moves; esc = s.esc;
                        track.upper = s.upper;
                        n_resumed++;
                    } else {
                        n_kept = 0;
                        n_fresh++;
                    }
                    /* the lane's state at the last return to the root */
                    Kept seg = { L.cell, L.checked, moves, esc, track.upper, L.t };
                    for (;;) {
                        const CountingFetch fetch{ nodes, &f_test, (uint32_t)(n_slots - 1u) & ~1u };
                        const uint32_t go = ahead ? vr_lane_descend_ahead(L, iso, fetch, st, max_trips, track)
                                                  : vr_lane_descend(L, iso, fetch, st, max_trips, track);
                        if (go == VR_GO_MORE) continue;
                        if (track.upper < seg.upper && share) { /* `upper` went down in the stretch that ends here */
                            if (n_kept == share) { for (uint32_t q = 1; q < n_kept; q++) kept[q - 1] = kept[q]; n_kept--; }
                            kept[n_kept++] = seg;
                        }
                        if (go == VR_GO_HIT) { flag = 1; break; }
                        moves++;
                        const uint32_t at_move = L.checked;
                        if (!vr_lane_move(L, k, esc)) break;
                        seg = Kept{ L.cell, at_move, moves, esc, track.upper, L.t };
                        n_segments++;
                        { uint32_t tb; std::memcpy(&tb, &L.t, 4); seen.push_back(tb); }
                        while (!vr_lane_pop(L, st, max_trips)) {}
                    }
                    const uint32_t e = track.upper < last ? track.upper : last;
                    for (uint32_t i = iso; i <= e; i++) {
                        const uint32_t j = i - iso_begin;
                        gray[j * frame + pix] = flag ? vr_shade(L.t) : 0;
                        cell_flag[j * frame + pix] = (flag ? vr_unpack_cell(L.cell) : 0u) | (flag << 24);
                        t_out[j * frame + pix] = L.t;
                        checked[j * frame + pix] = L.checked;
                        node[j * frame + pix] = flag ? L.g - 1u : 0u;
                        sums[j * 7 + 0] += L.checked; sums[j * 7 + 1] += flag; sums[j * 7 + 2] += 1u - flag;
                        sums[j * 7 + 4] += esc; sums[j * 7 + 5] += moves;
                    }
                    iso = e + 1u;
                }
            }
            std::sort(seen.begin(), seen.end());
            n_states += (uint64_t)(std::unique(seen.begin(), seen.end()) - seen.begin());
        }
    if (counts_out) { counts_out[0] = n_resumed + n_fresh; counts_out[1] = f_test; counts_out[2] = n_resumed; counts_out[3] = n_segments; counts_out[4] = n_states; }
    return 0;
}

/* ---- wide volumes: the path-cache traversal of vr_trace_wide.h on wide node words ---- */
#include "../../volumerenderer_amd/csrc/vr_trace_wide.h"

namespace {
struct HostWidePairs { /* children(i): the words at 2i + 1 and 2i + 2, clamped to the array as on the device */
    const uint32_t *nodes;
    uint32_t last_pair;
    uint32_t operator()(uint32_t i) const { return nodes[i]; }
    VrNodePair children(uint32_t i) const
    {
        uint32_t c = 2u * i + 1u;
        c = c < last_pair ? c : last_pair;
        return VrNodePair{ nodes[c], nodes[c + 1] };
    }
};
struct HostPath {
    uint32_t word[VR_WIDE_MAX_DEPTH];
    uint32_t get(uint32_t d) const { return word[d]; }
    void set(uint32_t d, uint32_t w) { word[d] = w; }
};
} // namespace

extern "C" int vr_emu_wide(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3], const float src[3],
                           const float forw[3], const float up[3], uint32_t width, uint32_t height,
                           float viewport_width, uint32_t iso, uint32_t pop_trips, uint32_t ahead_trips, uint8_t *rgb, uint32_t *cell_flag,
                           float *t_out, uint32_t *checked, uint32_t *node, uint64_t *sums /*7*/)
{
    VrViewK k;
    k.ox = src[0]; k.oy = src[1]; k.oz = src[2];
    k.fx = forw[0]; k.fy = forw[1]; k.fz = forw[2];
    k.ux = up[0]; k.uy = up[1]; k.uz = up[2];
    k.rx = k.fy * k.uz - k.fz * k.uy;
    k.ry = k.fz * k.ux - k.fx * k.uz;
    k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = viewport_width / (float)width;
    k.half_w = (float)(width / 2);
    k.half_h = (float)(height / 2);
    k.width = width; k.height = height;
    k.rdx = root_dim[0]; k.rdy = root_dim[1]; k.rdz = root_dim[2];
    k.root_hi = 0;
    k.n_slots = n_slots;
    std::memset(sums, 0, 7 * sizeof(uint64_t));
    for (uint32_t yi = 0; yi < height; yi++)
        for (uint32_t xi = 0; xi < width; xi++) {
            VrRay ray = vr_make_ray(k, xi, yi);
            HostPath path;
            const HostWidePairs fetch{ nodes, (uint32_t)(n_slots - 2u) };
            VrHit h = ahead_trips ? vr_trace_wide_ahead(k, ray, iso, fetch, path, pop_trips ? pop_trips : 64u, ahead_trips)
                                  : vr_trace_wide(k, ray, iso, fetch, path, pop_trips ? pop_trips : 64u);
            uint64_t pix = (uint64_t)yi * width + xi;
            uint8_t g = h.flag == 1u ? vr_shade(h.t) : 0;
            rgb[pix * 3] = rgb[pix * 3 + 1] = rgb[pix * 3 + 2] = g;
            cell_flag[pix] = h.cell | (h.flag << VR_WIDE_FLAG_SHIFT);
            t_out[pix] = h.flag == 2u ? 0.0f : h.t;
            checked[pix] = h.checked;
            node[pix] = h.node;
            sums[0] += h.checked;
            sums[1] += h.flag == 1u;
            sums[2] += h.flag == 0u;
            sums[3] += h.flag == 2u;
            sums[4] += h.escapes;
            sums[5] += h.moves;
            sums[6] += h.error;
        }
    return 0;
}

#include "../../volumerenderer_amd/csrc/vr_trilerp.h"

/* the library's trilerp and corner fetch on the CPU: n records of 17 floats in, n densities out */
extern "C" void vr_emu_trilerp(const float *in, uint64_t n, float *out)
{
    for (uint64_t i = 0; i < n; i++) {
        const float *p = in + i * 17;
        out[i] = vr_trilerp(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p + 9);
    }
}

extern "C" void vr_emu_corner_dens(const uint8_t *vox, uint32_t ry, uint32_t rz, uint32_t cx, uint32_t cy, uint32_t cz, float out[8])
{
    vr_corner_dens(vox, ry, rz, cx, cy, cz, out);
}

/* ---- the union walk (vr_trace_union.h): ONE traversal per ray for a whole range of iso values ---- */
#include "../../volumerenderer_amd/csrc/vr_trace_union.h"
namespace {
struct HostDone {
    uint32_t w[8];
    uint32_t word(uint32_t i) const { return w[i]; }
    void set(uint32_t i, uint32_t bits) { w[i] |= bits; }
    void unset(uint32_t i, uint32_t bits) { w[i] &= ~bits; }
    void clear(uint32_t i) { w[i] = 0u; }
    void fill(uint32_t i) { w[i] = 0xffffffffu; }
};
struct HostUnionSink {
    const VrViewK *k; const uint32_t *nodes; const VrLane *ray; /* for iso values that are walked on their own */
    uint32_t iso_begin; uint64_t frame, pix;
    uint8_t *gray; uint32_t *cell_flag; float *t_out; uint32_t *node; uint64_t *sums;
    uint64_t n_except = 0, n_except_isos = 0;
    /* one run kept for a front of its own, as the kernel's lanes do; further ones are walked iso value by iso value */
    bool pend = false; float pend_t = 0.0f; uint32_t pend_cell = 0; VrUnionRuns pend_runs = { 0u, 0u, 0u };
    void finish(uint32_t iso, bool hit, float t, uint32_t cell, uint32_t g)
    {
        const uint64_t at = (uint64_t)(iso - iso_begin) * frame + pix;
        gray[at] = hit ? vr_shade(t) : 0;
        cell_flag[at] = (hit ? vr_unpack_cell(cell) : 0u) | ((hit ? 1u : 0u) << 24);
        t_out[at] = t;
        node[at] = hit ? g - 1u : 0u;
        sums[(iso - iso_begin) * 7 + (hit ? 1 : 2)]++;
    }
    void park(uint32_t r0, uint32_t r1, uint32_t checks)
    {
        for (uint32_t i = r0; i <= r1; i++) { sums[(i - iso_begin) * 7 + 0] += checks; sums[(i - iso_begin) * 7 + 5] += 1u; }
    }
    void hit(uint32_t r0, uint32_t r1, uint32_t checks, float t, uint32_t cell, uint32_t g)
    {
        for (uint32_t i = r0; i <= r1; i++) { sums[(i - iso_begin) * 7 + 0] += checks; finish(i, true, t, cell, g); }
    }
    void miss(uint32_t r0, uint32_t r1, float t) { for (uint32_t i = r0; i <= r1; i++) finish(i, false, t, 0, 0); }
    void esc(uint32_t r0, uint32_t r1, uint32_t n) { for (uint32_t i = r0; i <= r1; i++) sums[(i - iso_begin) * 7 + 4] += n; }
    /* iso values whose own MOVE lands elsewhere: each walked on its own from its own state (the checks and the MOVE
     * of the round that parked them are counted already) */
    void except(uint32_t r0, uint32_t r1, float t, uint32_t cell, bool in_root, uint32_t n_esc)
    {
        n_except++;
        if (in_root && !pend) { pend = true; pend_t = t; pend_cell = cell; pend_runs.clear(); }
        if (in_root && cell == pend_cell && vr_f32_bits(t) == vr_f32_bits(pend_t) && pend_runs.room()) {
            n_except_isos += r1 - r0 + 1u;
            esc(r0, r1, n_esc);
            pend_runs.add(r0, r1);
            return;
        }
        for (uint32_t iso = r0; iso <= r1; iso++) {
            n_except_isos++;
            sums[(iso - iso_begin) * 7 + 4] += n_esc;
            if (!in_root) { finish(iso, false, t, 0, 0); continue; }
            HostStack st;
            VrLane A = *ray;
            vr_lane_restart(A, st, *k, t, cell);
            uint32_t moves = 0, e = 0;
            bool hit = false;
            for (;;) {
                const uint32_t go = vr_lane_descend(A, iso, HostFetch{ nodes }, st, 64u);
                if (go == VR_GO_HIT) { hit = true; break; }
                moves++;
                if (!vr_lane_move(A, *k, e)) break;
                while (!vr_lane_pop(A, st, 64u)) {}
            }
            sums[(iso - iso_begin) * 7 + 0] += A.checked; sums[(iso - iso_begin) * 7 + 4] += e; sums[(iso - iso_begin) * 7 + 5] += moves;
            finish(iso, hit, A.t, A.cell, A.g);
        }
    }
};
} // namespace

/* What the lane code's `suspect` stands for, level by level (test only): does any box the point has left -- every ancestor up
 * to, not including, the first one that holds the new point; all of them when the front has left the volume -- not share the
 * front's exit plane?  The lane code decides it from the largest box left alone. */
static bool emu_suspect_by_levels(VrLane L, HostStack st, const VrUnionExit &x, bool leaving)
{
    bool suspect = false;
    const uint32_t cg = L.cell | VR_F_GUARD, cp = L.cell + VR_F_ONES;
    while (st.depth() > 0u) {
        const uint32_t word = st.peek_up(1);
        st.up();
        const bool left = (L.g & 1u) == 0u;
        const uint32_t a_lo = left ? L.lo : word, a_hi = left ? word : L.hi;
        if (!leaving) {
            const uint32_t a = cg - a_lo, b = (a_hi | VR_F_GUARD) - cp;
            if (((a & b) & VR_F_GUARD) == VR_F_GUARD) break;
        }
        suspect = suspect || !vr_union_same_exit((a_hi & L.posmask) | (a_lo & ~L.posmask), x.ex, x.tied);
        L.lo = a_lo; L.hi = a_hi;
        L.g >>= 1;
    }
    return suspect;
}

/* counts_out: [0] node reads of the descents, [1] states (MOVEs of the front), [2] suspect MOVEs resolved exactly,
 * [3] runs of iso values that left the union walk, [4] iso values in them, [5] MOVEs whose `suspect` differs from the
 * level-by-level definition (emu_suspect_by_levels) */
extern "C" int vr_emu_union(const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3], const float src[3],
                            const float forw[3], const float up[3], uint32_t width, uint32_t height, float viewport_width,
                            uint32_t iso_begin, uint32_t iso_count, uint32_t max_trips, uint8_t *gray, uint32_t *cell_flag,
                            float *t_out, uint32_t *node, uint64_t *sums /*[iso_count][7]*/, uint64_t *counts_out /*[6]*/)
{
    uint64_t n_suspect_differs = 0;
    VrViewK k;
    k.ox = src[0]; k.oy = src[1]; k.oz = src[2];
    k.fx = forw[0]; k.fy = forw[1]; k.fz = forw[2];
    k.ux = up[0]; k.uy = up[1]; k.uz = up[2];
    k.rx = k.fy * k.uz - k.fz * k.uy;
    k.ry = k.fz * k.ux - k.fx * k.uz;
    k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = viewport_width / (float)width;
    k.half_w = (float)(width / 2);
    k.half_h = (float)(height / 2);
    k.width = width; k.height = height;
    k.rdx = root_dim[0]; k.rdy = root_dim[1]; k.rdz = root_dim[2];
    k.root_hi = vr_pack10(k.rdx, k.rdy, k.rdz);
    k.n_slots = n_slots;
    std::memset(sums, 0, (size_t)iso_count * 7 * sizeof(uint64_t));
    const uint64_t frame = (uint64_t)width * height;
    uint64_t n_reads = 0, n_states = 0, n_suspect = 0, n_except = 0, n_except_isos = 0;
    for (uint32_t yi = 0; yi < height; yi++)
        for (uint32_t xi = 0; xi < width; xi++) {
            const uint64_t pix = (uint64_t)yi * width + xi;
            VrRay ray = vr_make_ray(k, xi, yi);
            if (ray.oob) {
                for (uint32_t j = 0; j < iso_count; j++) {
                    gray[j * frame + pix] = 0; cell_flag[j * frame + pix] = 2u << 24; t_out[j * frame + pix] = 0.0f;
                    node[j * frame + pix] = 0; sums[j * 7 + 3]++;
                }
                continue;
            }
            HostStack st;
            HostDone done;
            VrUnionLane U;
            vr_union_begin(U, st, done, k, ray.dx, ray.dy, ray.dz, ray.t, iso_begin, iso_count);
            const VrLane ray_lane = U.L;
            HostUnionSink sink{ &k, nodes, &ray_lane, iso_begin, frame, pix, gray, cell_flag, t_out, node, sums };
            const CountingFetch fetch{ nodes, &n_reads, (uint32_t)(n_slots - 1u) & ~1u };
            for (;;) {
                while (U.alive_lo <= U.alive_hi) {
                    const uint32_t go = vr_union_descend<true>(U, fetch, st, done, sink, max_trips);
                    if (go == VR_GO_MORE) continue;
                    if (U.alive_lo > U.alive_hi) break;
                    n_states++;
                    const uint32_t depth_k = st.depth(), cell_before = U.L.cell;
                    const VrUnionExit x = vr_union_move(U.L, k);
                    bool suspect = false;
                    if (!x.in_root) {
                        const float t_new = U.L.t;
                        const bool by_levels = emu_suspect_by_levels(U.L, st, x, true);
                        vr_union_leave(U.L, st, x, suspect, k.root_hi);
                        n_suspect_differs += suspect != by_levels;
                        if (suspect) { n_suspect++; vr_union_resolve(U, k, HostFetch{ nodes }, done, sink, depth_k, cell_before, -1, t_new, 0u, false, VrUnionFrom{ 1u, 0u, k.root_hi }); }
                        else vr_union_credit_esc(done, U, sink, 0u, 255u, x.esc);
                        if (U.alive_lo <= U.alive_hi)
                            vr_union_runs(done, U.alive_lo, U.alive_hi, [&](uint32_t r0, uint32_t r1) { sink.miss(r0, r1, t_new); }, U.holes);
                        break;
                    }
                    VrUnionFrom from = { 1u, 0u, k.root_hi };
                    const bool by_levels = emu_suspect_by_levels(U.L, st, x, false);
                    while (!vr_union_pop(U.L, st, max_trips, x, suspect, from)) {}
                    n_suspect_differs += suspect != by_levels;
                    if (suspect) { n_suspect++; vr_union_resolve(U, k, HostFetch{ nodes }, done, sink, depth_k, cell_before, (int)st.depth() - 1, U.L.t, U.L.cell, true, from); }
                    const bool credit = !suspect;
                    vr_union_after_pop(U, HostFetch{ nodes });
                    if (credit) vr_union_credit_esc(done, U, sink, U.prev_mn, U.prev_mx, x.esc);
                }
                if (!sink.pend) break;
                sink.pend = false;
                vr_union_restart(U, st, done, k, sink.pend_t, sink.pend_cell, sink.pend_runs);
            }
            n_except += sink.n_except; n_except_isos += sink.n_except_isos;
        }
    if (counts_out) { counts_out[0] = n_reads; counts_out[1] = n_states; counts_out[2] = n_suspect; counts_out[3] = n_except; counts_out[4] = n_except_isos; counts_out[5] = n_suspect_differs; }
    return 0;
}

/* ---- the union walk on wide trees (vr_trace_union_wide.h) ---- */
#include "../../volumerenderer_amd/csrc/vr_trace_union_wide.h"
namespace {
struct HostUnionWideSink {
    const VrViewK *k; const uint32_t *nodes; const VrWideLane *ray;
    uint32_t iso_begin; uint64_t frame, pix;
    uint8_t *gray; uint32_t *cell_flag; float *t_out; uint32_t *node; uint64_t *sums;
    uint64_t n_except = 0, n_except_isos = 0;
    bool pend = false; float pend_t = 0.0f; uint32_t pend_cell = 0; VrUnionRuns pend_runs = { 0u, 0u, 0u };
    void finish(uint32_t iso, bool hit, float t, uint32_t cell, uint32_t idx)
    {
        const uint64_t at = (uint64_t)(iso - iso_begin) * frame + pix;
        gray[at] = hit ? vr_shade(t) : 0;
        cell_flag[at] = (hit ? cell : 0u) | ((hit ? 1u : 0u) << VR_WIDE_FLAG_SHIFT);
        t_out[at] = t;
        node[at] = hit ? idx : 0u;
        sums[(iso - iso_begin) * 7 + (hit ? 1 : 2)]++;
    }
    void park(uint32_t r0, uint32_t r1, uint32_t checks)
    {
        for (uint32_t i = r0; i <= r1; i++) { sums[(i - iso_begin) * 7 + 0] += checks; sums[(i - iso_begin) * 7 + 5] += 1u; }
    }
    void hit(uint32_t r0, uint32_t r1, uint32_t checks, float t, uint32_t cell, uint32_t idx)
    {
        for (uint32_t i = r0; i <= r1; i++) { sums[(i - iso_begin) * 7 + 0] += checks; finish(i, true, t, cell, idx); }
    }
    void miss(uint32_t r0, uint32_t r1, float t) { for (uint32_t i = r0; i <= r1; i++) finish(i, false, t, 0, 0); }
    void esc(uint32_t r0, uint32_t r1, uint32_t n) { for (uint32_t i = r0; i <= r1; i++) sums[(i - iso_begin) * 7 + 4] += n; }
    void except(uint32_t r0, uint32_t r1, float t, uint32_t cell, bool in_root, uint32_t n_esc)
    {
        n_except++;
        if (in_root && !pend) { pend = true; pend_t = t; pend_cell = cell; pend_runs.clear(); }
        if (in_root && cell == pend_cell && vr_f32_bits(t) == vr_f32_bits(pend_t) && pend_runs.room()) {
            n_except_isos += r1 - r0 + 1u;
            esc(r0, r1, n_esc);
            pend_runs.add(r0, r1);
            return;
        }
        for (uint32_t iso = r0; iso <= r1; iso++) { /* what does not fit: iso value by iso value, as the second kernel does */
            n_except_isos++;
            sums[(iso - iso_begin) * 7 + 4] += n_esc;
            if (!in_root) { finish(iso, false, t, 0, 0); continue; }
            HostPath path;
            VrWideLane A = *ray;
            vr_wide_restart(A, *k, t, cell);
            uint32_t moves = 0, e = 0;
            bool hit = false;
            for (;;) {
                const uint32_t go = vr_wide_descend(A, *k, iso, HostFetch{ nodes }, path, VR_WIDE_MAX_DEPTH + 1u);
                if (go == VR_GO_HIT) { hit = true; break; }
                moves++;
                if (!vr_wide_move(A, *k, e)) break;
                while (!vr_wide_pop(A, path, 64u)) {}
            }
            sums[(iso - iso_begin) * 7 + 0] += A.checked; sums[(iso - iso_begin) * 7 + 4] += e; sums[(iso - iso_begin) * 7 + 5] += moves;
            finish(iso, hit, A.t, A.c, A.idx);
        }
    }
};
} // namespace

extern "C" int vr_emu_union_wide(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3], const float src[3],
                                 const float forw[3], const float up[3], uint32_t width, uint32_t height, float viewport_width,
                                 uint32_t iso_begin, uint32_t iso_count, uint32_t max_trips, uint8_t *gray, uint32_t *cell_flag,
                                 float *t_out, uint32_t *node, uint64_t *sums /*[iso_count][7]*/, uint64_t *counts_out /*[5]*/)
{
    VrViewK k;
    k.ox = src[0]; k.oy = src[1]; k.oz = src[2];
    k.fx = forw[0]; k.fy = forw[1]; k.fz = forw[2];
    k.ux = up[0]; k.uy = up[1]; k.uz = up[2];
    k.rx = k.fy * k.uz - k.fz * k.uy;
    k.ry = k.fz * k.ux - k.fx * k.uz;
    k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = viewport_width / (float)width;
    k.half_w = (float)(width / 2);
    k.half_h = (float)(height / 2);
    k.width = width; k.height = height;
    k.rdx = root_dim[0]; k.rdy = root_dim[1]; k.rdz = root_dim[2];
    k.root_hi = 0;
    k.n_slots = n_slots;
    std::memset(sums, 0, (size_t)iso_count * 7 * sizeof(uint64_t));
    const uint64_t frame = (uint64_t)width * height;
    uint64_t n_reads = 0, n_states = 0, n_suspect = 0, n_except = 0, n_except_isos = 0;
    for (uint32_t yi = 0; yi < height; yi++)
        for (uint32_t xi = 0; xi < width; xi++) {
            const uint64_t pix = (uint64_t)yi * width + xi;
            VrRay ray = vr_make_ray(k, xi, yi);
            if (ray.oob) {
                for (uint32_t j = 0; j < iso_count; j++) {
                    gray[j * frame + pix] = 0; cell_flag[j * frame + pix] = 2u << VR_WIDE_FLAG_SHIFT; t_out[j * frame + pix] = 0.0f;
                    node[j * frame + pix] = 0; sums[j * 7 + 3]++;
                }
                continue;
            }
            HostPath path;
            HostDone done;
            VrUnionWideLane U;
            vr_uw_begin(U, done, k, ray.dx, ray.dy, ray.dz, ray.t, iso_begin, iso_count);
            const VrWideLane ray_lane = U.L;
            HostUnionWideSink sink{ &k, nodes, &ray_lane, iso_begin, frame, pix, gray, cell_flag, t_out, node, sums };
            const CountingFetch fetch{ nodes, &n_reads, 0u };
            for (;;) {
                while (U.alive_lo <= U.alive_hi) {
                    const uint32_t go = vr_uw_descend<true>(U, fetch, path, done, sink, max_trips);
                    if (go == VR_GO_MORE) continue;
                    if (U.alive_lo > U.alive_hi) break;
                    n_states++;
                    const uint32_t depth_k = U.L.d, cell_before = U.L.c;
                    const VrUnionWideExit x = vr_uw_move(U.L, k);
                    bool suspect = false;
                    if (!x.in_root) {
                        const float t_new = U.L.t;
                        vr_uw_leave(U.L, path, x, suspect, vr_pack10(k.rdx, k.rdy, k.rdz));
                        if (suspect) { n_suspect++; vr_uw_resolve(U, k, HostFetch{ nodes }, done, sink, depth_k, cell_before, -1, t_new, 0u, false, VrUnionFrom{ 0u, 0u, vr_pack10(k.rdx, k.rdy, k.rdz) }); }
                        else vr_union_credit_esc(done, U, sink, 0u, 255u, x.esc);
                        if (U.alive_lo <= U.alive_hi)
                            vr_union_runs(done, U.alive_lo, U.alive_hi, [&](uint32_t r0, uint32_t r1) { sink.miss(r0, r1, t_new); }, U.holes);
                        break;
                    }
                    VrUnionFrom from = { 0u, 0u, vr_pack10(k.rdx, k.rdy, k.rdz) };
                    while (!vr_uw_pop(U.L, path, max_trips, x, suspect, from)) {}
                    if (suspect) { n_suspect++; vr_uw_resolve(U, k, HostFetch{ nodes }, done, sink, depth_k, cell_before, (int)U.L.d - 1, U.L.t, U.L.c, true, from); }
                    const bool credit = !suspect;
                    vr_uw_after_pop(U, HostFetch{ nodes });
                    if (credit) vr_union_credit_esc(done, U, sink, U.prev_mn, U.prev_mx, x.esc);
                }
                if (!sink.pend) break;
                sink.pend = false;
                vr_uw_restart(U, done, k, sink.pend_t, sink.pend_cell, sink.pend_runs);
            }
            n_except += sink.n_except; n_except_isos += sink.n_except_isos;
        }
    if (counts_out) { counts_out[0] = n_reads; counts_out[1] = n_states; counts_out[2] = n_suspect; counts_out[3] = n_except; counts_out[4] = n_except_isos; }
    return 0;
}
