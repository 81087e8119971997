/*
 * vr_oracle.c -- CPU restatement of the reference's isosurface ray-casting path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check and the
 * cpu_baseline leg of bench.py may load it, and only as the checker.  The product library
 * (libvr_hip.so) never links, loads or calls anything in this file.
 *
 * Parity status: PINNED.  The restatement reproduces all 256 images the reference holds under
 * renders/out_iso{0..255}.png (C60.vol, 512x512, camera of src/main.zig:31-43) with zero
 * differing pixels; tests/test_oracle_golden.py checks that on every run from the committed
 * fixtures.  Per-ray hit cells, t bit patterns and nodes_checked are not held by any reference
 * artefact; for those this file is the definition the GPU path must match bit-for-bit.
 *
 * The reference is Zig (default strict float mode: no contraction, IEEE div/sqrt).  Build this
 * file with  -O2 -ffp-contract=off -fno-fast-math  (see oracle/Makefile).  Every function cites
 * the reference lines it follows (paths relative to the reference root).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(__FAST_MATH__)
#error "the oracle must be built without fast-math"
#endif

#define VRO_OK 0
#define VRO_EIO (-1)
#define VRO_EFORMAT (-2)
#define VRO_ENOMEM (-3)
#define VRO_EARG (-4)
#define VRO_ETOODEEP (-5)

/* Coordinate type.  The reference's SPACESIZE is u8 (src/typedef.zig:1), so its volumes end at 256
 * voxels per axis.  Built with -DVRO_WIDE the same code runs on u16 coordinates (up to 1024 voxels per
 * axis) with a node word that holds a 12-bit partition:
 *     bits 0-7 min | 8-15 max | 16-27 partition | 28-29 axis
 * No reference artefact exists for such volumes; the wide build is pinned by agreeing with the u8 build
 * (and through it with the reference's renders) on every volume both can represent. */
#ifdef VRO_WIDE
typedef uint16_t coord_t;
#define VRO_MAX_RES 1024u
#else
typedef uint8_t coord_t;
#define VRO_MAX_RES 256u
#endif

/* ---------------------------------------------------------------- geo.zig primitives ---- */

typedef struct { float x, y, z; } vec3;                 /* src/geo.zig:55-60  Vec3(f32)       */
typedef struct { coord_t min, max; } range8;            /* src/geo.zig:155-159 Range(u8)      */
typedef struct { range8 r[3]; } box8;                   /* src/geo.zig:202-207 Volume(u8)     */

static inline vec3 v_new(float x, float y, float z) { vec3 v = { x, y, z }; return v; }
/* src/geo.zig:91-93 */
static inline vec3 v_scale(vec3 a, float s) { return v_new(a.x * s, a.y * s, a.z * s); }
/* src/geo.zig:95-97 */
static inline vec3 v_add(vec3 a, vec3 b) { return v_new(a.x + b.x, a.y + b.y, a.z + b.z); }
/* src/geo.zig:99-101 */
static inline vec3 v_sub(vec3 a, vec3 b) { return v_new(a.x - b.x, a.y - b.y, a.z - b.z); }
/* src/geo.zig:107-109 : left-to-right sum of the three products */
static inline float v_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* src/geo.zig:111-113 */
static inline float v_len(vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* src/geo.zig:115-121 */
static inline vec3 v_cross(vec3 a, vec3 b)
{
    return v_new(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* src/geo.zig:123-126 : scale by (1.0 / len), not a division per component */
static inline vec3 v_norm(vec3 a) { return v_scale(a, 1.0f / v_len(a)); }
/* src/geo.zig:37-39  Ray.point */
static inline vec3 ray_point(vec3 o, vec3 d, float t) { return v_add(o, v_scale(d, t)); }

/* src/geo.zig:12-18  Axis.baseVector */
static inline vec3 axis_base(int axis)
{
    switch (axis) {
    case 0: return v_new(1, 0, 0);
    case 1: return v_new(0, 1, 0);
    default: return v_new(0, 0, 1);
    }
}

/* src/geo.zig:173-176 */
static inline int range_contains_float(range8 r, float v)
{
    return (float)r.min <= v && v < (float)r.max;
}
/* src/geo.zig:196-198 */
static inline int range_contains_inclusive(range8 r, uint8_t v) { return r.min <= v && v <= r.max; }
/* src/geo.zig:221-226 */
static inline int box_contains_float(box8 b, vec3 p)
{
    return range_contains_float(b.r[0], p.x) && range_contains_float(b.r[1], p.y) &&
           range_contains_float(b.r[2], p.z);
}
/* src/geo.zig:315-317 (range len: src/geo.zig:192-194) */
static inline uint32_t box_size(box8 b)
{
    return (uint32_t)(coord_t)(b.r[0].max - b.r[0].min) * (uint32_t)(coord_t)(b.r[1].max - b.r[1].min) *
           (uint32_t)(coord_t)(b.r[2].max - b.r[2].min);
}
/* src/geo.zig:319-327 */
static inline int box_largest_dim(box8 b)
{
    coord_t lx = b.r[0].max - b.r[0].min, ly = b.r[1].max - b.r[1].min, lz = b.r[2].max - b.r[2].min;
    if (lx > ly && lx > lz) return 0;
    if (ly > lz) return 1;
    return 2;
}
/* src/geo.zig:256-275 : which = 0 -> the half below the split, 1 -> the half at/above it */
static inline box8 box_split_global(box8 b, int axis, coord_t split, int which)
{
    if (which == 0) {
        if (split < b.r[axis].max) b.r[axis].max = split;
    } else {
        if (split > b.r[axis].min) b.r[axis].min = split;
    }
    return b;
}

/* src/geo.zig:406-426  Plane.rayIntersect, written out literally (normal vector products kept) */
static inline float plane_ray_intersect(int axis, coord_t offset, vec3 o, vec3 d)
{
    vec3 normal = axis_base(axis);
    vec3 center = v_scale(normal, (float)offset);
    float denom = v_dot(normal, d);
    if (fabsf(denom) < 1e-8f) return INFINITY;
    float t = v_dot(v_sub(center, o), normal) / denom;
    if (t > 0.0f) return t;
    return -1.0f;
}
/* src/geo.zig:366-375  Volume.planes : index 2*axis -> min plane, 2*axis+1 -> max plane */
static inline coord_t box_plane_offset(box8 b, int plane_idx)
{
    return (plane_idx & 1) ? b.r[plane_idx >> 1].max : b.r[plane_idx >> 1].min;
}

/* --------------------------------------------------------------------- Data.zig -------- */

/* src/Data.zig:90-92 : x slowest, z fastest */
static inline uint8_t data_get(const uint8_t *vox, const uint32_t res[3], uint32_t x, uint32_t y, uint32_t z)
{
    return vox[x * (res[2] * res[1]) + y * res[2] + z];
}

static uint32_t be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

/* src/Data.zig:9-44  loadUCD on an in-memory file image: 3 x u32 BE dims, 4 ignored u32, then
 * x*y*z bytes in x-major / z-fastest order, i.e. the body IS the in-memory array. */
int vro_parse_ucd(const uint8_t *file, uint64_t file_len, uint32_t res_out[3], const uint8_t **body_out)
{
    if (!file || file_len < 28) return VRO_EFORMAT;
    res_out[0] = be32(file + 0);
    res_out[1] = be32(file + 4);
    res_out[2] = be32(file + 8);
    uint64_t n = (uint64_t)res_out[0] * res_out[1] * res_out[2];
    if (n == 0 || file_len < 28 + n) return VRO_EFORMAT;
    *body_out = file + 28;
    return VRO_OK;
}

/* src/Data.zig:9-44 : loads a .vol file; caller frees *vox_out with vro_free */
int vro_load_ucd(const char *path, uint8_t **vox_out, uint32_t res_out[3])
{
    FILE *f = fopen(path, "rb");
    if (!f) return VRO_EIO;
    uint8_t hdr[28];
    if (fread(hdr, 1, 28, f) != 28) { fclose(f); return VRO_EFORMAT; }
    res_out[0] = be32(hdr + 0);
    res_out[1] = be32(hdr + 4);
    res_out[2] = be32(hdr + 8);
    uint64_t n = (uint64_t)res_out[0] * res_out[1] * res_out[2];
    if (n == 0) { fclose(f); return VRO_EFORMAT; }
    uint8_t *v = (uint8_t *)malloc(n);
    if (!v) { fclose(f); return VRO_ENOMEM; }
    if (fread(v, 1, n, f) != n) { free(v); fclose(f); return VRO_EFORMAT; }
    fclose(f);
    *vox_out = v;
    return VRO_OK;
}

void vro_free(void *p) { free(p); }

/* ------------------------------------------------------------------- SlimKDT.zig ------- */

/* src/SlimKDT.zig:9-13 : packed struct, LSB first: min, max, partition, axis(2 bits) */
#ifdef VRO_WIDE
#define NODE_PART_BITS 12
#else
#define NODE_PART_BITS 8
#endif
static inline uint32_t node_pack(uint8_t mn, uint8_t mx, coord_t part, int axis)
{
    return (uint32_t)mn | ((uint32_t)mx << 8) | ((uint32_t)part << 16) | ((uint32_t)(axis & 3) << (16 + NODE_PART_BITS));
}
#define NODE_MIN(n) ((uint8_t)((n) & 0xff))
#define NODE_MAX(n) ((uint8_t)(((n) >> 8) & 0xff))
#define NODE_PART(n) ((coord_t)(((n) >> 16) & ((1u << NODE_PART_BITS) - 1u)))
#define NODE_AXIS(n) ((int)(((n) >> (16 + NODE_PART_BITS)) & 3))

typedef struct {
    uint64_t non_center_partition_count; /* src/SlimKDT.zig:215-219 */
    uint64_t layer_count;
    uint64_t number_of_nodes;
} vro_build_stats;

typedef struct {
    const uint8_t *vox;
    uint32_t res[3];
    uint32_t *nodes;
    uint64_t cap;
    uint64_t max_slots; /* refuse to grow past this: the reference's out-of-memory path */
    int heuristic;
    int err;
} build_ctx;

/* heuristic ids: the functions of src/SlimKDT.zig:327-598 */
enum {
    VRO_H_ALWAYS_CENTER = 0,  /* :359-367 */
    VRO_H_MINIMIZE_SPAN = 1,  /* :327-357 */
    VRO_H_FLEX_FIRST5 = 2,    /* :369-375 */
    VRO_H_FLEX_FIRST10 = 3,   /* :377-383 */
    VRO_H_FLEX_AFTER10 = 4,   /* :385-391 */
    VRO_H_FLEX_AFTER15 = 5,   /* :392-398 */
    VRO_H_FLEX_AFTER20 = 6,   /* :400-406 */
    VRO_H_CW5 = 7,            /* :408-438 */
    VRO_H_CW10 = 8,           /* :440-470 */
    VRO_H_CW30 = 9,           /* :472-502 */
    VRO_H_CW50 = 10,          /* :504-534 */
    VRO_H_CW75 = 11,          /* :536-566 */
    VRO_H_CW100 = 12,         /* :568-598 */
    VRO_H_COUNT = 13
};

/* min/max over the 8 corners of every cell of a box = voxels [min..max] inclusive on each axis
 * (src/SlimKDT.zig:275-281 with src/Data.zig:102-114 and the iterator of src/geo.zig:333-364) */
static void box_corner_range(const build_ctx *c, box8 b, uint8_t *mn_out, uint8_t *mx_out)
{
    uint8_t mn = 255, mx = 0;
    for (uint32_t z = b.r[2].min; z < b.r[2].max; z++)
        for (uint32_t y = b.r[1].min; y < b.r[1].max; y++)
            for (uint32_t x = b.r[0].min; x < b.r[0].max; x++)
                for (int k = 0; k < 8; k++) {
                    uint8_t v = data_get(c->vox, c->res, x + (k & 1), y + ((k >> 1) & 1), z + ((k >> 2) & 1));
                    if (v < mn) mn = v;
                    if (v > mx) mx = v;
                }
    *mn_out = mn;
    *mx_out = mx;
}

/* src/SlimKDT.zig:327-357 and the copies at :408-598 which differ only in the weight */
static float span_heuristic(const build_ctx *c, coord_t p, coord_t mid, box8 space, int axis, float weight)
{
    uint8_t lmin, lmax, rmin, rmax;
    box_corner_range(c, box_split_global(space, axis, p, 0), &lmin, &lmax);
    box_corner_range(c, box_split_global(space, axis, p, 1), &rmin, &rmax);
    float h = 0;
    h += fabsf((float)p - (float)mid) * weight;
    h += fabsf((float)lmax - (float)lmin);
    h += fabsf((float)rmax - (float)rmin);
    return h;
}

/* src/SlimKDT.zig:359-367 */
static float always_center(coord_t p, coord_t mid) { return p == mid ? 0.0f : 1.0f; }

static float heuristic_eval(const build_ctx *c, coord_t p, coord_t mid, box8 space, int axis, uint32_t layer)
{
    switch (c->heuristic) {
    case VRO_H_ALWAYS_CENTER: return always_center(p, mid);
    case VRO_H_MINIMIZE_SPAN: return span_heuristic(c, p, mid, space, axis, 1.0f);
    case VRO_H_FLEX_FIRST5: return layer < 5 ? span_heuristic(c, p, mid, space, axis, 1.0f) : always_center(p, mid);
    case VRO_H_FLEX_FIRST10: return layer < 10 ? span_heuristic(c, p, mid, space, axis, 1.0f) : always_center(p, mid);
    case VRO_H_FLEX_AFTER10: return layer > 10 ? span_heuristic(c, p, mid, space, axis, 1.0f) : always_center(p, mid);
    case VRO_H_FLEX_AFTER15: return layer > 15 ? span_heuristic(c, p, mid, space, axis, 1.0f) : always_center(p, mid);
    case VRO_H_FLEX_AFTER20: return layer > 20 ? span_heuristic(c, p, mid, space, axis, 1.0f) : always_center(p, mid);
    case VRO_H_CW5: return span_heuristic(c, p, mid, space, axis, 5.0f);
    case VRO_H_CW10: return span_heuristic(c, p, mid, space, axis, 10.0f);
    case VRO_H_CW30: return span_heuristic(c, p, mid, space, axis, 30.0f);
    case VRO_H_CW50: return span_heuristic(c, p, mid, space, axis, 50.0f);
    case VRO_H_CW75: return span_heuristic(c, p, mid, space, axis, 75.0f);
    default: return span_heuristic(c, p, mid, space, axis, 100.0f);
    }
}

/* src/SlimKDT.zig:248-322  newHeuristicPartitionInner.  The child boxes follow
 * SpaceTracker.leftHalf/rightHalf (:66-88): split on largestDim at the stored partition. */
static vro_build_stats build_inner(build_ctx *c, box8 space, uint64_t idx, uint32_t layer)
{
    vro_build_stats out = { 0, 0, 0 };
    if (c->err) return out;
    if (box_size(space) < 1) return out; /* :257-259 */

    int axis = box_largest_dim(space); /* :264 */

    if (idx >= c->cap) { /* :266-273 : the reference reallocs; a failed realloc fails the build */
        if (idx >= c->max_slots) { c->err = VRO_ETOODEEP; return out; }
        uint64_t ncap = c->cap;
        while (idx >= ncap) ncap *= 2;
        if (ncap > c->max_slots) ncap = c->max_slots;
        uint32_t *nn = (uint32_t *)realloc(c->nodes, ncap * sizeof(uint32_t));
        if (!nn) { c->err = VRO_ENOMEM; return out; }
        memset(nn + c->cap, 0, (ncap - c->cap) * sizeof(uint32_t));
        c->nodes = nn;
        c->cap = ncap;
    }

    uint8_t mn, mx;
    box_corner_range(c, space, &mn, &mx); /* :275-281 */

    /* :282-284 */
    coord_t mid = (coord_t)(((uint32_t)space.r[axis].min + (uint32_t)space.r[axis].max) / 2);
    coord_t best = mid;
    float best_h = INFINITY;
    /* :289-298 : candidates min+1 .. max-1, strict improvement keeps the first minimum */
    for (uint32_t cand = (uint32_t)space.r[axis].min + 1; cand < space.r[axis].max; cand++) {
        float h = heuristic_eval(c, (coord_t)cand, mid, space, axis, layer);
        if (best_h > h) { best = (coord_t)cand; best_h = h; }
    }

    c->nodes[idx] = node_pack(mn, mx, best, axis); /* :301-308 */

    if (box_size(space) == 1) { /* :310-312 */
        out.number_of_nodes = 1;
        out.layer_count = layer;
        return out;
    }

    vro_build_stats r1 = build_inner(c, box_split_global(space, axis, best, 0), 2 * idx + 1, layer + 1);
    vro_build_stats r2 = build_inner(c, box_split_global(space, axis, best, 1), 2 * idx + 2, layer + 1);

    /* :317-321 */
    out.non_center_partition_count = r1.non_center_partition_count + r2.non_center_partition_count + (mid != best);
    uint64_t lc = r1.layer_count > r2.layer_count ? r1.layer_count : r2.layer_count;
    out.layer_count = lc > layer ? lc : layer;
    out.number_of_nodes = r1.number_of_nodes + r2.number_of_nodes + 1;
    return out;
}

/* src/SlimKDT.zig:222-246 newHeursiticPartition + :127-144 init.
 * Returns a heap-indexed array of packed nodes (unreachable slots are zero; the reference leaves
 * them uninitialised).  *n_slots_out = array length.  Free with vro_free. */
int vro_build_kdt(const uint8_t *vox, const uint32_t res[3], int heuristic, uint64_t max_slots,
                  uint32_t **nodes_out, uint64_t *n_slots_out, coord_t root_dim_out[3], vro_build_stats *stats_out)
{
    if (!vox || !res || heuristic < 0 || heuristic >= VRO_H_COUNT) return VRO_EARG;
    for (int a = 0; a < 3; a++)
        if (res[a] < 2 || res[a] > VRO_MAX_RES) return VRO_EARG; /* SPACESIZE = u8: src/typedef.zig:1, src/SlimKDT.zig:129-133 */
    build_ctx c;
    c.vox = vox;
    c.res[0] = res[0]; c.res[1] = res[1]; c.res[2] = res[2];
    c.cap = (uint64_t)res[0] * res[1] * res[2] * 2; /* :139 */
    c.max_slots = max_slots ? max_slots : ((uint64_t)1 << 32);
    if (c.cap > c.max_slots) c.cap = c.max_slots;
    c.nodes = (uint32_t *)calloc(c.cap, sizeof(uint32_t));
    if (!c.nodes) return VRO_ENOMEM;
    c.heuristic = heuristic;
    c.err = 0;
    box8 root;
    for (int a = 0; a < 3; a++) { root.r[a].min = 0; root.r[a].max = (coord_t)(res[a] - 1); }
    vro_build_stats st = build_inner(&c, root, 0, 0);
    if (c.err) { free(c.nodes); return c.err; }
    *nodes_out = c.nodes;
    *n_slots_out = c.cap;
    for (int a = 0; a < 3; a++) root_dim_out[a] = (coord_t)(res[a] - 1);
    if (stats_out) *stats_out = st;
    return VRO_OK;
}

/* ------------------------------------------------------------------ raytrace.zig ------- */

typedef struct {
    float t;               /* src/raytrace.zig:108-115 RayRes */
    int hit;
    int oob;
    float escape_help_count;
    float enter_vol_t;
    uint64_t checked_nodes;
    /* extra parity taps the reference only holds transiently */
    coord_t cell[3];
    uint32_t node_idx;
    uint32_t moves;
} ray_res;

/* what the reference's debug print shows of every loop iteration (dbp = true, src/raytrace.zig:164-198) */
typedef struct {
    float t, p[3];
    uint32_t node, size;
    uint8_t min, max, isoval, action; /* action: 0 ZOOM, 1 HIT, 2 MOVE */
} vro_ray_step;
typedef struct { vro_ray_step *out; uint32_t cap, n; } step_rec;

static void rec_step(step_rec *rec, float t, vec3 p, uint32_t node, uint32_t size, uint32_t word, uint8_t iso, uint8_t action)
{
    if (!rec) return;
    if (rec->n < rec->cap) {
        vro_ray_step *s = &rec->out[rec->n];
        s->t = t; s->p[0] = p.x; s->p[1] = p.y; s->p[2] = p.z;
        s->node = node; s->size = size;
        s->min = NODE_MIN(word); s->max = NODE_MAX(word); s->isoval = iso; s->action = action;
    }
    rec->n++;
}

/* src/raytrace.zig:117-255 traceRay with the SpaceTracker of src/SlimKDT.zig:48-115 inlined */
static ray_res trace_ray_rec(vec3 o, vec3 d, const uint32_t *nodes, const coord_t root_dim[3], uint8_t iso, step_rec *rec, float *t_entry)
{
    ray_res res;
    memset(&res, 0, sizeof res);
    float t = 0.0f;
    float escape_help_count = 0;

    /* SpaceTracker.new : src/SlimKDT.zig:55-64 */
    box8 root;
    for (int a = 0; a < 3; a++) { root.r[a].min = 0; root.r[a].max = root_dim[a]; }
    box8 space = root;
    uint32_t idx = 0;

    /* :122  do = dir.gtz().cell() : 1 where the component is >= 0 (so -0.0 counts as >= 0) */
    int dox = d.x >= 0.0f ? 1 : 0, doy = d.y >= 0.0f ? 1 : 0, doz = d.z >= 0.0f ? 1 : 0;

    uint64_t checked = 0;

    /* :127-131 entry */
    t = fmaxf(t, plane_ray_intersect(0, box_plane_offset(space, 1 - dox), o, d));
    t = fmaxf(t, plane_ray_intersect(1, box_plane_offset(space, 3 - doy), o, d));
    t = fmaxf(t, plane_ray_intersect(2, box_plane_offset(space, 5 - doz), o, d));
    t += 0.001f;
    if (t_entry) *t_entry = t;

    if (!box_contains_float(space, ray_point(o, d, t))) { /* :148-156 */
        res.oob = 1;
        return res;
    }
    float enter_vol_t = t;

    while (box_contains_float(space, ray_point(o, d, t))) { /* :161 */
        checked += 1;
        vec3 p = ray_point(o, d, t);
        coord_t cell[3] = { (coord_t)p.x, (coord_t)p.y, (coord_t)p.z }; /* :163, src/geo.zig:74-81 */
        uint32_t n = nodes[idx];
        range8 dr = { NODE_MIN(n), NODE_MAX(n) };
        rec_step(rec, t, p, idx, box_size(space), n, iso,
                 range_contains_inclusive(dr, iso) ? (box_size(space) == 1 ? 1 : 0) : 2); /* the print of :164-174 */
        if (range_contains_inclusive(dr, iso)) { /* :177 */
            if (box_size(space) == 1) {          /* :178-187 */
                res.t = t;
                res.hit = 1;
                res.escape_help_count = escape_help_count;
                res.enter_vol_t = enter_vol_t;
                res.checked_nodes = checked;
                res.cell[0] = cell[0]; res.cell[1] = cell[1]; res.cell[2] = cell[2];
                res.node_idx = idx;
                return res;
            }
            /* SpaceTracker.zoom : src/SlimKDT.zig:90-103 */
            int axis = NODE_AXIS(n);
            coord_t part = NODE_PART(n);
            if (cell[axis] < part) {
                space = box_split_global(space, axis, part, 0);
                idx = 2 * idx + 1;
            } else {
                space = box_split_global(space, axis, part, 1);
                idx = 2 * idx + 2;
            }
            continue;
        }
        /* MOVE : :199-205 */
        t = INFINITY;
        t = fminf(t, plane_ray_intersect(0, box_plane_offset(space, 0 + dox), o, d));
        t = fminf(t, plane_ray_intersect(1, box_plane_offset(space, 2 + doy), o, d));
        t = fminf(t, plane_ray_intersect(2, box_plane_offset(space, 4 + doz), o, d));
        t += 0.0001f;
        res.moves++;

        /* escape help : :233-242 */
        float i = 0;
        while (box_contains_float(space, ray_point(o, d, t))) {
            t += 0.001f * i;
            i += 1;
            escape_help_count += 1;
        }
        /* SpaceTracker.reset : src/SlimKDT.zig:105-114 */
        space = root;
        idx = 0;
    }
    res.t = t; /* :248-254 */
    res.hit = 0;
    res.escape_help_count = escape_help_count;
    res.enter_vol_t = enter_vol_t;
    res.checked_nodes = checked;
    return res;
}

static ray_res trace_ray(vec3 o, vec3 d, const uint32_t *nodes, const coord_t root_dim[3], uint8_t iso)
{
    return trace_ray_rec(o, d, nodes, root_dim, iso, NULL, NULL);
}

typedef struct {
    uint64_t nodes_checked;   /* what renderImage returns: src/raytrace.zig:105 */
    uint64_t hit_count, miss_count, oob_count;
    uint64_t escape_help_events;
    uint64_t moves;
    uint64_t max_checked_per_ray;
} vro_render_stats;

/* Pixel shade of src/raytrace.zig:56-61: 255 - u8(trunc(min(200, t - 300))).  The reference's
 * @intFromFloat is undefined for t - 300 <= -1 (camera closer than 300 to the hit); this build
 * defines that case as 0 (so the pixel is 255) -- see the shading note in the header. */
static inline uint8_t shade(float t)
{
    float v = t - 300.0f;
    if (v > 200.0f) v = 200.0f; /* @min(200, v): v is never NaN on a hit */
    if (!(v > 0.0f)) v = 0.0f;
    return (uint8_t)(255 - (uint8_t)v);
}

#define VRO_FLAG_MISS 0u
#define VRO_FLAG_HIT 1u
#define VRO_FLAG_OOB 2u

/* src/raytrace.zig:11-106 renderImage, rows [row_begin,row_end) of a width x height image.
 * rgb24 points at the FULL image (row-major, top row first, 3 B/pixel); only the selected rows
 * are written.  Optional per-pixel taps (full-image sized, may be NULL):
 *   tap_cell_flag : x | y<<8 | z<<16 | flag<<24  (flag: 0 miss, 1 hit, 2 oob; cell is 0 unless hit);
 *                   the wide build packs x | y<<10 | z<<20 | flag<<30
 *   tap_t         : RayRes.t (0 for oob)
 *   tap_checked   : RayRes.checked_nodes
 *   tap_node      : heap index of the hit leaf (0 unless hit) */
int vro_render_rows(const uint32_t *nodes, uint64_t n_slots, const coord_t root_dim[3], const float src[3],
                    const float forw[3], const float up[3], uint32_t width, uint32_t height, float viewport_width,
                    uint8_t iso, uint32_t row_begin, uint32_t row_end, uint8_t *rgb24, vro_render_stats *stats,
                    uint32_t *tap_cell_flag, float *tap_t, uint32_t *tap_checked, uint32_t *tap_node)
{
    if (!nodes || !n_slots || !root_dim || !src || !forw || !up || !rgb24 || !width || !height) return VRO_EARG;
    if (row_end > height || row_begin > row_end) return VRO_EARG;
    vec3 o = v_new(src[0], src[1], src[2]);
    vec3 f = v_new(forw[0], forw[1], forw[2]);
    vec3 u = v_new(up[0], up[1], up[2]);
    float pixel_size = viewport_width / (float)width; /* :24 */
    vec3 right = v_cross(f, u);                       /* :38 */
    vro_render_stats st;
    memset(&st, 0, sizeof st);
    for (uint32_t yi = row_begin; yi < row_end; yi++) {
        float yf = ((float)yi - (float)(height / 2)) * pixel_size; /* :41 */
        for (uint32_t xi = 0; xi < width; xi++) {
            float xf = ((float)xi - (float)(width / 2)) * pixel_size; /* :43 */
            vec3 dir = v_norm(v_add(f, v_add(v_scale(u, yf), v_scale(right, xf)))); /* :45 */
            ray_res r = trace_ray(o, dir, nodes, root_dim, iso);
            uint64_t pix = (uint64_t)yi * width + xi;
            uint8_t g = 0;
            uint32_t flag;
            if (r.oob) { flag = VRO_FLAG_OOB; st.oob_count++; }
            else if (r.hit) { flag = VRO_FLAG_HIT; g = shade(r.t); st.hit_count++; }
            else { flag = VRO_FLAG_MISS; st.miss_count++; }
            rgb24[pix * 3 + 0] = g;
            rgb24[pix * 3 + 1] = g;
            rgb24[pix * 3 + 2] = g;
            st.nodes_checked += r.checked_nodes;
            st.escape_help_events += (uint64_t)r.escape_help_count;
            st.moves += r.moves;
            if (r.checked_nodes > st.max_checked_per_ray) st.max_checked_per_ray = r.checked_nodes;
#ifdef VRO_WIDE
            if (tap_cell_flag)
                tap_cell_flag[pix] = (uint32_t)r.cell[0] | ((uint32_t)r.cell[1] << 10) | ((uint32_t)r.cell[2] << 20) | (flag << 30);
#else
            if (tap_cell_flag)
                tap_cell_flag[pix] = (uint32_t)r.cell[0] | ((uint32_t)r.cell[1] << 8) | ((uint32_t)r.cell[2] << 16) | (flag << 24);
#endif
            if (tap_t) tap_t[pix] = r.oob ? 0.0f : r.t;
            if (tap_checked) tap_checked[pix] = (uint32_t)r.checked_nodes;
            if (tap_node) tap_node[pix] = r.hit ? r.node_idx : 0;
        }
    }
    if (stats) *stats = st;
    (void)n_slots;
    return VRO_OK;
}

int vro_render_image(const uint32_t *nodes, uint64_t n_slots, const coord_t root_dim[3], const float src[3],
                     const float forw[3], const float up[3], uint32_t width, uint32_t height, float viewport_width,
                     uint8_t iso, uint8_t *rgb24, vro_render_stats *stats, uint32_t *tap_cell_flag, float *tap_t,
                     uint32_t *tap_checked, uint32_t *tap_node)
{
    return vro_render_rows(nodes, n_slots, root_dim, src, forw, up, width, height, viewport_width, iso, 0, height,
                           rgb24, stats, tap_cell_flag, tap_t, tap_checked, tap_node);
}

/* Camera of src/main.zig:31-43 : center = f32(res/2) (integer division), forw = norm(center-pos),
 * up = (forw x (0,1,0)) x forw, NOT normalised. */
void vro_main_camera(const uint32_t res[3], const float pos[3], float forw_out[3], float up_out[3])
{
    vec3 center = v_new((float)(res[0] / 2), (float)(res[1] / 2), (float)(res[2] / 2));
    vec3 p = v_new(pos[0], pos[1], pos[2]);
    vec3 dir = v_norm(v_sub(center, p));
    vec3 c = v_cross(dir, v_new(0.0f, 1.0f, 0.0f));
    c = v_cross(c, dir);
    forw_out[0] = dir.x; forw_out[1] = dir.y; forw_out[2] = dir.z;
    up_out[0] = c.x; up_out[1] = c.y; up_out[2] = c.z;
}

/* ---- trilinear refinement (the part of it the reference holds) ---------------------------------------------- */

/* src/raytrace.zig:257-291 trilerp, operation for operation (strict f32):
 * values = v000, v100, v010, v110, v001, v101, v011, v111. */
float vro_trilerp(float x, float y, float z, float x0, float x1, float y0, float y1, float z0, float z1, const float values[8])
{
    const float xd = (x - x0) / (x1 - x0);
    const float yd = (y - y0) / (y1 - y0);
    const float zd = (z - z0) / (z1 - z0);
    const float c00 = values[0] * (1 - xd) + values[1] * xd;
    const float c10 = values[2] * (1 - xd) + values[3] * xd;
    const float c01 = values[4] * (1 - xd) + values[5] * xd;
    const float c11 = values[6] * (1 - xd) + values[7] * xd;
    const float c0 = c00 * (1 - yd) + c10 * yd;
    const float c1 = c01 * (1 - yd) + c11 * yd;
    return c0 * (1 - zd) + c1 * zd;
}

/* n points: in = n x 17 floats (x, y, z, x0, x1, y0, y1, z0, z1, values[8]) */
void vro_trilerp_batch(const float *in, uint64_t n, float *out)
{
    for (uint64_t i = 0; i < n; i++) {
        const float *p = in + i * 17;
        out[i] = vro_trilerp(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p + 9);
    }
}

/* src/Data.zig:102-114 getCornerDens: the 8 voxels at the corners of a cell, in trilerp's vertex order */
void vro_corner_dens(const uint8_t *vox, const uint32_t res[3], uint32_t cx, uint32_t cy, uint32_t cz, uint8_t out[8])
{
    out[0] = data_get(vox, res, cx, cy, cz);
    out[1] = data_get(vox, res, cx + 1, cy, cz);
    out[2] = data_get(vox, res, cx, cy + 1, cz);
    out[3] = data_get(vox, res, cx + 1, cy + 1, cz);
    out[4] = data_get(vox, res, cx, cy, cz + 1);
    out[5] = data_get(vox, res, cx + 1, cy, cz + 1);
    out[6] = data_get(vox, res, cx, cy + 1, cz + 1);
    out[7] = data_get(vox, res, cx + 1, cy + 1, cz + 1);
}

/* The refinement the reference only sketches (trilerpTrace, src/raytrace.zig:295-304, fetches the hit cell's corner
 * densities and stops): for every pixel whose ray hit, the density trilerp gives at the hit point o + d * t inside
 * the hit cell [c, c + 1]^3; 0 elsewhere.  Beyond trilerp and getCornerDens there is nothing in the reference to
 * compare this with: PARITY UNPINNED.  cell_flag / t are the taps of vro_render_rows (full-image sized). */
int vro_refine_rows(const uint8_t *vox, const uint32_t res[3], const float src[3], const float forw[3], const float up[3],
                    uint32_t width, uint32_t height, float viewport_width, uint32_t row_begin, uint32_t row_end,
                    const uint32_t *tap_cell_flag, const float *tap_t, float *density)
{
    if (!vox || !res || !src || !forw || !up || !tap_cell_flag || !tap_t || !density || !width || !height) return VRO_EARG;
    if (row_end > height || row_begin > row_end) return VRO_EARG;
    vec3 o = v_new(src[0], src[1], src[2]);
    vec3 f = v_new(forw[0], forw[1], forw[2]);
    vec3 u = v_new(up[0], up[1], up[2]);
    float pixel_size = viewport_width / (float)width;
    vec3 right = v_cross(f, u);
    for (uint32_t yi = row_begin; yi < row_end; yi++) {
        float yf = ((float)yi - (float)(height / 2)) * pixel_size;
        for (uint32_t xi = 0; xi < width; xi++) {
            uint64_t pix = (uint64_t)yi * width + xi;
            uint32_t w = tap_cell_flag[pix];
#ifdef VRO_WIDE
            uint32_t flag = w >> 30, cx = w & 0x3ffu, cy = (w >> 10) & 0x3ffu, cz = (w >> 20) & 0x3ffu;
#else
            uint32_t flag = w >> 24, cx = w & 0xffu, cy = (w >> 8) & 0xffu, cz = (w >> 16) & 0xffu;
#endif
            density[pix] = 0.0f;
            if (flag != VRO_FLAG_HIT) continue;
            if (cx + 1 >= res[0] || cy + 1 >= res[1] || cz + 1 >= res[2]) return VRO_EARG;
            float xf = ((float)xi - (float)(width / 2)) * pixel_size;
            vec3 dir = v_norm(v_add(f, v_add(v_scale(u, yf), v_scale(right, xf))));
            vec3 p = ray_point(o, dir, tap_t[pix]);
            uint8_t c8[8];
            float c[8];
            vro_corner_dens(vox, res, cx, cy, cz, c8);
            for (int k = 0; k < 8; k++) c[k] = (float)c8[k];
            density[pix] = vro_trilerp(p.x, p.y, p.z, (float)cx, (float)(cx + 1), (float)cy, (float)(cy + 1), (float)cz,
                                       (float)(cz + 1), c);
        }
    }
    return VRO_OK;
}

/* One ray of renderImage's pixel grid (src/raytrace.zig:41-45) traced with every loop iteration recorded: what the
 * reference prints with dbp = true.  report = origin[3], dir[3], t_entry, t_final, then flag, n_steps, n_stored. */
typedef struct { float origin[3], dir[3], t_entry, t_final; uint32_t flag, n_steps, n_stored; } vro_ray_report;

#ifndef VRO_WIDE
int vro_trace_ray(const uint32_t *nodes, const coord_t root_dim[3], const float src[3], const float forw[3], const float up[3],
                  uint32_t width, uint32_t height, float viewport_width, uint8_t iso, uint32_t xi, uint32_t yi,
                  vro_ray_step *steps, uint32_t cap, vro_ray_report *report)
{
    if (!nodes || !root_dim || !src || !forw || !up || !report || !width || !height || xi >= width || yi >= height) return VRO_EARG;
    vec3 o = v_new(src[0], src[1], src[2]);
    vec3 f = v_new(forw[0], forw[1], forw[2]);
    vec3 u = v_new(up[0], up[1], up[2]);
    float pixel_size = viewport_width / (float)width;
    vec3 right = v_cross(f, u);
    float yf = ((float)yi - (float)(height / 2)) * pixel_size;
    float xf = ((float)xi - (float)(width / 2)) * pixel_size;
    vec3 dir = v_norm(v_add(f, v_add(v_scale(u, yf), v_scale(right, xf))));
    step_rec rec = { steps, cap, 0 };
    float t_entry = 0.0f;
    ray_res r = trace_ray_rec(o, dir, nodes, root_dim, iso, &rec, &t_entry);
    report->origin[0] = o.x; report->origin[1] = o.y; report->origin[2] = o.z;
    report->dir[0] = dir.x; report->dir[1] = dir.y; report->dir[2] = dir.z;
    report->t_entry = t_entry;
    report->t_final = r.oob ? 0.0f : r.t;
    report->flag = r.oob ? VRO_FLAG_OOB : (r.hit ? VRO_FLAG_HIT : VRO_FLAG_MISS);
    report->n_steps = rec.n;
    report->n_stored = rec.n < cap ? rec.n : cap;
    return VRO_OK;
}
#endif
