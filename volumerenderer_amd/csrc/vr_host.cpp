/*
 * vr_host.cpp -- host-side producers of the ray-casting path: the .vol loader, the SlimKDT
 * builder and the reference driver's camera.  No GPU code.  Built with -ffp-contract=off so
 * the camera arithmetic matches the reference's strict f32.
 *
 *   vr_load_ucd        Data.loadUCD                       src/Data.zig:9-44
 *   vr_host_build_kdt  SlimKDT.newHeursiticPartition      src/SlimKDT.zig:222-322, heuristics :327-598
 *   vr_main_camera     camera set-up of the driver        src/main.zig:31-43
 *
 * The builder does not scan every cell of a node's box per split candidate the way the reference
 * does (src/SlimKDT.zig:275-298, :327-357).  It uses two identities that give the same numbers:
 *   * a node's density range is the union of its children's ranges (the child boxes share the
 *     voxel plane at the partition, so together they cover exactly the parent's voxels), hence
 *     ranges are produced bottom-up from the 8-corner range of each leaf cell;
 *   * for a split candidate p the left/right ranges are prefix/suffix min/max over per-plane
 *     ("slab") ranges along the split axis, so all candidates of a node cost one pass over its
 *     voxels.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/vr.h"

namespace {

struct Box {
    uint8_t lo[3], hi[3];
    uint32_t len(int a) const { return (uint32_t)(uint8_t)(hi[a] - lo[a]); }
    uint32_t size() const { return len(0) * len(1) * len(2); } /* src/geo.zig:315-317 */
    int largest_dim() const                                     /* src/geo.zig:319-327 */
    {
        if (len(0) > len(1) && len(0) > len(2)) return 0;
        if (len(1) > len(2)) return 1;
        return 2;
    }
};

struct MinMax {
    uint8_t mn, mx;
    void merge(MinMax o)
    {
        if (o.mn < mn) mn = o.mn;
        if (o.mx > mx) mx = o.mx;
    }
};

struct Builder {
    const uint8_t *vox;
    uint32_t res[3];
    uint64_t sx, sy; /* strides of x and y in the voxel array (z stride is 1), src/Data.zig:90-92 */
    std::vector<uint32_t> nodes;
    uint64_t max_slots;
    int heuristic;
    int err;
    vr_build_stats stats;

    /* is the span heuristic (src/SlimKDT.zig:327-357) the one consulted at this layer, and with
     * which deviation weight?  Otherwise AlwaysCenterH (:359-367) decides. */
    bool span_active(uint32_t layer, float *weight) const
    {
        *weight = 1.0f;
        switch (heuristic) {
        case VR_H_ALWAYS_CENTER: return false;
        case VR_H_MINIMIZE_SPAN: return true;
        case VR_H_FLEXIBLE_FIRST5: return layer < 5;
        case VR_H_FLEXIBLE_FIRST10: return layer < 10;
        case VR_H_FLEXIBLE_AFTER10: return layer > 10;
        case VR_H_FLEXIBLE_AFTER15: return layer > 15;
        case VR_H_FLEXIBLE_AFTER20: return layer > 20;
        case VR_H_FLEXIBLE_CENTER_WEIGHTED5: *weight = 5.0f; return true;
        case VR_H_FLEXIBLE_CENTER_WEIGHTED10: *weight = 10.0f; return true;
        case VR_H_FLEXIBLE_CENTER_WEIGHTED30: *weight = 30.0f; return true;
        case VR_H_FLEXIBLE_CENTER_WEIGHTED50: *weight = 50.0f; return true;
        case VR_H_FLEXIBLE_CENTER_WEIGHTED75: *weight = 75.0f; return true;
        default: *weight = 100.0f; return true;
        }
    }

    /* per-plane density range along `axis` over the voxels [lo..hi] (inclusive) of the box */
    void slab_ranges(const Box &b, int axis, MinMax *slab) const
    {
        for (uint32_t s = b.lo[axis]; s <= b.hi[axis]; s++) { slab[s].mn = 255; slab[s].mx = 0; }
        for (uint32_t x = b.lo[0]; x <= b.hi[0]; x++)
            for (uint32_t y = b.lo[1]; y <= b.hi[1]; y++) {
                const uint8_t *row = vox + x * sx + y * sy;
                if (axis == 2) {
                    for (uint32_t z = b.lo[2]; z <= b.hi[2]; z++) {
                        uint8_t v = row[z];
                        if (v < slab[z].mn) slab[z].mn = v;
                        if (v > slab[z].mx) slab[z].mx = v;
                    }
                } else {
                    uint8_t mn = 255, mx = 0;
                    for (uint32_t z = b.lo[2]; z <= b.hi[2]; z++) {
                        uint8_t v = row[z];
                        if (v < mn) mn = v;
                        if (v > mx) mx = v;
                    }
                    uint32_t s = axis == 0 ? x : y;
                    if (mn < slab[s].mn) slab[s].mn = mn;
                    if (mx > slab[s].mx) slab[s].mx = mx;
                }
            }
    }

    /* src/SlimKDT.zig:286-298: first candidate in min+1 .. max-1 with the strictly smallest score */
    uint8_t choose_partition(const Box &b, int axis, uint8_t mid, uint32_t layer) const
    {
        uint32_t lo = b.lo[axis], hi = b.hi[axis];
        if (lo + 1 >= hi) return mid; /* no candidate: the initial value stands (:286) */
        float weight;
        if (!span_active(layer, &weight)) return mid; /* score 0 at mid, 1 elsewhere: mid is in range and wins */
        MinMax slab[256], pre[256], suf[256];
        slab_ranges(b, axis, slab);
        pre[lo] = slab[lo];
        for (uint32_t s = lo + 1; s <= hi; s++) { pre[s] = pre[s - 1]; pre[s].merge(slab[s]); }
        suf[hi] = slab[hi];
        for (uint32_t s = hi; s-- > lo;) { suf[s] = suf[s + 1]; suf[s].merge(slab[s]); }
        uint8_t best = mid;
        float best_h = INFINITY;
        for (uint32_t p = lo + 1; p < hi; p++) {
            /* left half holds cells [lo,p) = voxels lo..p ; right half cells [p,hi) = voxels p..hi */
            float h = 0;
            h += std::fabs((float)p - (float)mid) * weight;
            h += std::fabs((float)pre[p].mx - (float)pre[p].mn);
            h += std::fabs((float)suf[p].mx - (float)suf[p].mn);
            if (best_h > h) { best = (uint8_t)p; best_h = h; }
        }
        return best;
    }

    bool ensure_slot(uint64_t idx)
    {
        if (idx < nodes.size()) return true;
        if (idx >= max_slots) { err = VR_ETOODEEP; return false; } /* the realloc of :266-270 fails */
        uint64_t ncap = nodes.size();
        while (idx >= ncap) ncap *= 2;
        if (ncap > max_slots) ncap = max_slots;
        try {
            nodes.resize(ncap, 0u);
        } catch (const std::bad_alloc &) {
            err = VR_ENOMEM;
            return false;
        }
        return true;
    }

    /* src/SlimKDT.zig:248-322, children by SpaceTracker.leftHalf/rightHalf (:66-88) */
    MinMax build(const Box &b, uint64_t idx, uint32_t layer, uint64_t *non_center, uint64_t *layers, uint64_t *count)
    {
        MinMax r = { 255, 0 };
        if (err) return r;
        if (b.size() < 1) return r; /* :257-259: an empty half adds no node */
        if (!ensure_slot(idx)) return r;
        int axis = b.largest_dim();
        uint8_t mid = (uint8_t)(((uint32_t)b.lo[axis] + (uint32_t)b.hi[axis]) / 2);
        if (b.size() == 1) {
            for (int k = 0; k < 8; k++) {
                uint8_t v = vox[(b.lo[0] + (k & 1)) * sx + (b.lo[1] + ((k >> 1) & 1)) * sy + (b.lo[2] + ((k >> 2) & 1))];
                if (v < r.mn) r.mn = v;
                if (v > r.mx) r.mx = v;
            }
            nodes[idx] = (uint32_t)r.mn | ((uint32_t)r.mx << 8) | ((uint32_t)mid << 16) | ((uint32_t)axis << 24);
            *count += 1;
            if (layer > *layers) *layers = layer;
            return r;
        }
        uint8_t part = choose_partition(b, axis, mid, layer);
        Box l = b, rr = b;
        if (part < l.hi[axis]) l.hi[axis] = part;   /* src/geo.zig:256-275 */
        if (part > rr.lo[axis]) rr.lo[axis] = part;
        r = build(l, 2 * idx + 1, layer + 1, non_center, layers, count);
        MinMax r2 = build(rr, 2 * idx + 2, layer + 1, non_center, layers, count);
        if (err) return r;
        r.merge(r2);
        nodes[idx] = (uint32_t)r.mn | ((uint32_t)r.mx << 8) | ((uint32_t)part << 16) | ((uint32_t)axis << 24);
        *count += 1;
        *non_center += (mid != part);
        if (layer > *layers) *layers = layer;
        return r;
    }
};

uint32_t be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

} // namespace

extern "C" {

const char *vr_strerror(int code)
{
    switch (code) {
    case VR_OK: return "ok";
    case VR_EARG: return "invalid argument";
    case VR_ENOMEM: return "out of host memory";
    case VR_ESTATE: return "no volume / tree on the device yet";
    case VR_ETOODEEP: return "tree exceeds the node slot limit (FAILD TO BUILD KDT)";
    case VR_EIO: return "cannot read file";
    case VR_EFORMAT: return "malformed .vol file";
    case VR_ENODEVICE: return "no gfx950 HIP device";
    case VR_ELIMIT: return "beyond a fixed limit of the entry point";
    case VR_ETREE: return "malformed tree (a reachable node lies outside the array or does not split its box)";
    default: return code <= VR_EHIP_BASE ? "HIP runtime error (code = VR_EHIP_BASE - hipError_t)" : "unknown error";
    }
}

int vr_load_ucd(const char *path, uint8_t **vox_out, uint32_t res_out[3])
{
    if (!path || !vox_out || !res_out) return VR_EARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return VR_EIO;
    uint8_t hdr[28]; /* 3 x u32 BE dims, 1 unused u32, 3 x 4 bytes "true size" (ignored), src/Data.zig:17-27 */
    if (std::fread(hdr, 1, sizeof hdr, f) != sizeof hdr) { std::fclose(f); return VR_EFORMAT; }
    uint32_t r[3] = { be32(hdr), be32(hdr + 4), be32(hdr + 8) };
    uint64_t n = (uint64_t)r[0] * r[1] * r[2];
    if (n == 0 || n > ((uint64_t)1 << 36)) { std::fclose(f); return VR_EFORMAT; }
    uint8_t *v = (uint8_t *)std::malloc(n);
    if (!v) { std::fclose(f); return VR_ENOMEM; }
    /* the file body is already in x-major / z-fastest order (src/Data.zig:30-41 with :94-96) */
    if (std::fread(v, 1, n, f) != n) { std::free(v); std::fclose(f); return VR_EFORMAT; }
    std::fclose(f);
    *vox_out = v;
    res_out[0] = r[0]; res_out[1] = r[1]; res_out[2] = r[2];
    return VR_OK;
}

/* The .vol layout of Data.loadUCD (src/Data.zig:9-44) written back: 3 x u32 BE dims, one unused u32,
 * three BE f32 "true size" values (1.0 here; the loader ignores them), then the voxel bytes verbatim. */
int vr_save_ucd(const char *path, const uint8_t *vox, const uint32_t res[3])
{
    if (!path || !vox || !res || !res[0] || !res[1] || !res[2]) return VR_EARG;
    FILE *f = std::fopen(path, "wb");
    if (!f) return VR_EIO;
    uint8_t hdr[28] = { 0 };
    for (int a = 0; a < 3; a++) {
        hdr[4 * a + 0] = (uint8_t)(res[a] >> 24); hdr[4 * a + 1] = (uint8_t)(res[a] >> 16);
        hdr[4 * a + 2] = (uint8_t)(res[a] >> 8);  hdr[4 * a + 3] = (uint8_t)res[a];
        hdr[16 + 4 * a] = 0x3f; hdr[17 + 4 * a] = 0x80; /* 1.0f, big endian */
    }
    const uint64_t n = (uint64_t)res[0] * res[1] * res[2];
    const bool ok = std::fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr && std::fwrite(vox, 1, n, f) == n;
    return (std::fclose(f) == 0 && ok) ? VR_OK : VR_EIO;
}

/* Tree cache file (the reference only has a stub for it, src/___SlimKDT_.zig:266-269): the heap-indexed
 * SlimNode array with its trailing unoccupied slots cut off.
 *   bytes 0-7   "VRKDT\0\1\0"            magic + format version 1
 *   bytes 8-10  root_dim, byte 11 0
 *   bytes 12-15 0
 *   bytes 16-23 n_slots   (u64 LE)  length of the array in memory
 *   bytes 24-31 n_stored  (u64 LE)  words that follow (index of the last non-zero word + 1)
 *   bytes 32-39 FNV-1a 64 of the stored words' bytes
 *   then n_stored u32 LE words */
namespace {
const uint8_t kKdtMagic[8] = { 'V', 'R', 'K', 'D', 'T', 0, 1, 0 };
uint64_t fnv1a64(const uint8_t *p, uint64_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
void put64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
uint64_t get64(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
} // namespace

int vr_save_kdt(const char *path, const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3])
{
    if (!path || !nodes || !n_slots || !root_dim) return VR_EARG;
    uint64_t n_stored = n_slots;
    while (n_stored > 0 && nodes[n_stored - 1] == 0u) n_stored--;
    uint8_t hdr[40] = { 0 };
    std::memcpy(hdr, kKdtMagic, 8);
    hdr[8] = root_dim[0]; hdr[9] = root_dim[1]; hdr[10] = root_dim[2];
    put64(hdr + 16, n_slots);
    put64(hdr + 24, n_stored);
    put64(hdr + 32, fnv1a64((const uint8_t *)nodes, n_stored * 4));
    FILE *f = std::fopen(path, "wb");
    if (!f) return VR_EIO;
    const bool ok = std::fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr && std::fwrite(nodes, 4, n_stored, f) == n_stored;
    return (std::fclose(f) == 0 && ok) ? VR_OK : VR_EIO;
}

int vr_load_kdt(const char *path, uint32_t **nodes_out, uint64_t *n_slots_out, uint8_t root_dim_out[3])
{
    if (!path || !nodes_out || !n_slots_out || !root_dim_out) return VR_EARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return VR_EIO;
    uint8_t hdr[40];
    if (std::fread(hdr, 1, sizeof hdr, f) != sizeof hdr || std::memcmp(hdr, kKdtMagic, 8) != 0) { std::fclose(f); return VR_EFORMAT; }
    const uint64_t n_slots = get64(hdr + 16), n_stored = get64(hdr + 24);
    if (n_slots == 0 || n_slots > ((uint64_t)1 << 32) || n_stored > n_slots) { std::fclose(f); return VR_EFORMAT; }
    uint32_t *nodes = (uint32_t *)std::calloc(n_slots, 4);
    if (!nodes) { std::fclose(f); return VR_ENOMEM; }
    const bool ok = std::fread(nodes, 4, n_stored, f) == n_stored && fnv1a64((const uint8_t *)nodes, n_stored * 4) == get64(hdr + 32);
    std::fclose(f);
    if (!ok) { std::free(nodes); return VR_EFORMAT; }
    *nodes_out = nodes;
    *n_slots_out = n_slots;
    root_dim_out[0] = hdr[8]; root_dim_out[1] = hdr[9]; root_dim_out[2] = hdr[10];
    return VR_OK;
}

int vr_host_build_kdt(const uint8_t *vox, const uint32_t res[3], int heuristic, uint64_t max_slots,
                      uint32_t **nodes_out, uint64_t *n_slots_out, uint8_t root_dim_out[3], vr_build_stats *stats_out)
{
    if (!vox || !res || !nodes_out || !n_slots_out || !root_dim_out) return VR_EARG;
    if (heuristic < 0 || heuristic >= VR_H_COUNT) return VR_EARG;
    for (int a = 0; a < 3; a++)
        if (res[a] < 2 || res[a] > 256) return VR_EARG; /* SPACESIZE = u8 (src/typedef.zig:1): root_dim = res-1 <= 255 */
    Builder b;
    b.vox = vox;
    b.res[0] = res[0]; b.res[1] = res[1]; b.res[2] = res[2];
    b.sx = (uint64_t)res[1] * res[2];
    b.sy = res[2];
    b.max_slots = max_slots ? max_slots : ((uint64_t)1 << 32);
    b.heuristic = heuristic;
    b.err = 0;
    uint64_t cap = 1;
    uint64_t want = (uint64_t)res[0] * res[1] * res[2] * 2; /* src/SlimKDT.zig:139 */
    while (cap < want) cap *= 2;
    if (cap > b.max_slots) cap = b.max_slots;
    try {
        b.nodes.assign(cap, 0u);
    } catch (const std::bad_alloc &) {
        return VR_ENOMEM;
    }
    Box root;
    for (int a = 0; a < 3; a++) { root.lo[a] = 0; root.hi[a] = (uint8_t)(res[a] - 1); }
    uint64_t non_center = 0, layers = 0, count = 0;
    b.build(root, 0, 0, &non_center, &layers, &count);
    if (b.err) return b.err;
    /* keep the smallest power of two that holds every used slot */
    uint64_t used = (uint64_t)1 << (layers + 1);
    if (used > b.nodes.size()) used = b.nodes.size();
    uint32_t *out = (uint32_t *)std::malloc(used * sizeof(uint32_t));
    if (!out) return VR_ENOMEM;
    std::memcpy(out, b.nodes.data(), used * sizeof(uint32_t));
    *nodes_out = out;
    *n_slots_out = used;
    for (int a = 0; a < 3; a++) root_dim_out[a] = (uint8_t)(res[a] - 1);
    if (stats_out) {
        stats_out->non_center_partition_count = non_center;
        stats_out->layer_count = layers;
        stats_out->number_of_nodes = count;
    }
    return VR_OK;
}

void vr_main_camera(const uint32_t res[3], const float pos[3], float forw_out[3], float up_out[3])
{
    /* data_center = f32(res / 2) with integer division, src/main.zig:31-35 */
    float c[3] = { (float)(res[0] / 2), (float)(res[1] / 2), (float)(res[2] / 2) };
    float d[3] = { c[0] - pos[0], c[1] - pos[1], c[2] - pos[2] };
    float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); /* src/geo.zig:111-113 */
    float inv = 1.0f / len;                                         /* src/geo.zig:123-125 */
    float f[3] = { d[0] * inv, d[1] * inv, d[2] * inv };
    /* c = forw x (0,1,0) ; up = c x forw   (src/main.zig:40-43, cross: src/geo.zig:115-121) */
    float w[3] = { 0.0f, 1.0f, 0.0f };
    float a[3] = { f[1] * w[2] - f[2] * w[1], f[2] * w[0] - f[0] * w[2], f[0] * w[1] - f[1] * w[0] };
    float u[3] = { a[1] * f[2] - a[2] * f[1], a[2] * f[0] - a[0] * f[2], a[0] * f[1] - a[1] * f[0] };
    for (int k = 0; k < 3; k++) { forw_out[k] = f[k]; up_out[k] = u[k]; }
}

uint64_t vr_prepared_words(uint64_t n_slots)
{
    return n_slots + 1; /* 1-based heap: the node at heap index i sits at position i + 1 */
}

int vr_host_prepare_kdt(const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3], uint32_t *prepared_out,
                        uint64_t prepared_words, uint32_t *max_depth_out, uint64_t *n_nodes_out)
{
    if (!nodes || !n_slots || !root_dim) return VR_EARG;
    if (n_slots > ((uint64_t)1 << 32)) return VR_EARG;
    if (root_dim[0] == 0 || root_dim[1] == 0 || root_dim[2] == 0) return VR_EARG;
    if (prepared_out) std::memset(prepared_out, 0, prepared_words * sizeof(uint32_t));
    struct Item { uint64_t idx; Box box; uint32_t depth; };
    std::vector<Item> todo;
    todo.reserve(128);
    Item root;
    root.idx = 0; root.depth = 0;
    for (int a = 0; a < 3; a++) { root.box.lo[a] = 0; root.box.hi[a] = root_dim[a]; }
    todo.push_back(root);
    uint32_t max_depth = 0;
    uint64_t count = 0;
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (it.idx >= n_slots) return VR_ETREE; /* a descent would read past the array */
        const uint32_t n = nodes[it.idx];
        const uint32_t axis = (n >> 24) & 3u, part = (n >> 16) & 0xffu;
        count++;
        if (it.depth > max_depth) max_depth = it.depth;
        uint32_t word;
        const bool leaf = it.box.size() == 1; /* leaf cell: src/raytrace.zig:178 */
        if (leaf) {
            word = (n & 0x00ffffffu) | 0x80000000u;
        } else {
            /* an inner node must split strictly inside its box, else a child is empty and the other one
             * repeats its parent: the reference's zoom (src/SlimKDT.zig:90-103) would never reach a leaf */
            if (axis > 2u || part <= it.box.lo[axis] || part >= it.box.hi[axis]) return VR_ETREE;
            word = (n & 0x00ffffffu) | ((axis * 10u) << 24);
        }
        if (prepared_out) {
            if (it.idx + 1 >= prepared_words) return VR_EARG;
            prepared_out[it.idx + 1] = word;
        }
        if (leaf) continue;
        Item l = it, r = it;
        l.idx = 2 * it.idx + 1; r.idx = 2 * it.idx + 2;
        l.depth = r.depth = it.depth + 1;
        l.box.hi[axis] = (uint8_t)part;
        r.box.lo[axis] = (uint8_t)part;
        todo.push_back(r);
        todo.push_back(l);
    }
    if (max_depth_out) *max_depth_out = max_depth;
    if (n_nodes_out) *n_nodes_out = count;
    return VR_OK;
}

/* Are the density ranges nested along every path of a prepared tree (every reachable node's range inside its
 * parent's)?  True for every tree built from a volume -- a node's range is the union of its children's -- and what the
 * kernel with one traversal per ray for all iso values relies on.  Words of unreachable slots are 0 and skipped
 * (a reachable node's word is never 0: a leaf has bit 31 set, an inner node a partition above its box's lower bound). */
bool vr_host_ranges_nested(const uint32_t *prepared, uint64_t prepared_words)
{
    for (uint64_t g = 2; g < prepared_words; g++) {
        const uint32_t n = prepared[g], p = prepared[g >> 1];
        if (n == 0u || p == 0u) continue;
        if ((n & 0xffu) < (p & 0xffu) || ((n >> 8) & 0xffu) > ((p >> 8) & 0xffu)) return false;
    }
    return true;
}

int vr_host_check_kdt_wide_nested(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3], uint32_t *max_depth_out,
                                  uint64_t *n_nodes_out, bool *nested_out);

int vr_host_check_kdt_wide(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3], uint32_t *max_depth_out,
                           uint64_t *n_nodes_out)
{
    return vr_host_check_kdt_wide_nested(nodes, n_slots, root_dim, max_depth_out, n_nodes_out, nullptr);
}

/* the same walk; *nested_out (if given) = every reachable node's density range lies inside its parent's */
int vr_host_check_kdt_wide_nested(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3], uint32_t *max_depth_out,
                                  uint64_t *n_nodes_out, bool *nested_out)
{
    if (!nodes || !n_slots || !root_dim) return VR_EARG;
    if (n_slots > ((uint64_t)1 << 31)) return VR_EARG;
    for (int a = 0; a < 3; a++)
        if (root_dim[a] == 0 || root_dim[a] > 1023) return VR_EARG;
    struct Item { uint64_t idx; uint16_t lo[3], hi[3]; uint32_t depth; uint8_t pmn, pmx; };
    std::vector<Item> todo;
    todo.reserve(128);
    Item root;
    root.idx = 0; root.depth = 0; root.pmn = 0; root.pmx = 255;
    bool nested = true;
    for (int a = 0; a < 3; a++) { root.lo[a] = 0; root.hi[a] = root_dim[a]; }
    todo.push_back(root);
    uint32_t max_depth = 0;
    uint64_t count = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.idx >= n_slots) return VR_ETREE; /* a descent would read past the array */
        const uint32_t n = nodes[it.idx];
        const uint32_t axis = (n >> 28) & 3u, part = (n >> 16) & 0xfffu;
        count++;
        if (it.depth > max_depth) max_depth = it.depth;
        if ((n & 0xffu) < it.pmn || ((n >> 8) & 0xffu) > it.pmx) nested = false;
        const uint32_t size = (uint32_t)(it.hi[0] - it.lo[0]) * (uint32_t)(it.hi[1] - it.lo[1]) * (uint32_t)(it.hi[2] - it.lo[2]);
        if (size == 1) continue; /* leaf cell: src/raytrace.zig:178 */
        /* an inner node must split strictly inside its box (see vr_host_prepare_kdt) */
        if (axis > 2u || part <= it.lo[axis] || part >= it.hi[axis]) return VR_ETREE;
        Item l = it, r = it;
        l.idx = 2 * it.idx + 1; r.idx = 2 * it.idx + 2;
        l.depth = r.depth = it.depth + 1;
        l.pmn = r.pmn = (uint8_t)(n & 0xffu); l.pmx = r.pmx = (uint8_t)((n >> 8) & 0xffu);
        l.hi[axis] = (uint16_t)part;
        r.lo[axis] = (uint16_t)part;
        todo.push_back(r);
        todo.push_back(l);
    }
    if (max_depth_out) *max_depth_out = max_depth;
    if (n_nodes_out) *n_nodes_out = count;
    if (nested_out) *nested_out = nested;
    return VR_OK;
}

void vr_free(void *p) { std::free(p); }

} // extern "C"

/* The layout of the reference's per-ray debug print (traceRay with dbp = true, src/raytrace.zig:133-146, 149-154,
 * 164-198); the lines it prints inside a MOVE that needs escape help (:207-240) are not reproduced. */
extern "C" int vr_print_ray_trace(const vr_ray_report *r, const vr_ray_step *steps, const char *path)
{
    if (!r || (r->n_stored && !steps)) return VR_EARG;
    FILE *f = path ? std::fopen(path, "w") : stderr;
    if (!f) return VR_EIO;
    const float t = r->t_entry;
    std::fprintf(f, " \n\n ### NEW RAY ### \nd= (%.2f %.2f %.2f)  o= (%.2f %.2f %.2f)\nskip to (%.2f %.2f %.2f) t=%.3f\n",
                 r->dir[0], r->dir[1], r->dir[2], r->origin[0], r->origin[1], r->origin[2], r->origin[0] + r->dir[0] * t,
                 r->origin[1] + r->dir[1] * t, r->origin[2] + r->dir[2] * t, t);
    if (r->flag == VR_FLAG_OOB) std::fprintf(f, "\n oob\n \n");
    static const char *const what[3] = { "ZOOM!", "HIT!", "MOVE!" };
    for (uint32_t i = 0; i < r->n_stored; i++) {
        const vr_ray_step &s = steps[i];
        std::fprintf(f, " \nt=%.3f  r=(%.2f %.2f %.2f)\nnode=%u   size=%u   iso=[%u-%u] v %u\n  res: %s\n", s.t, s.p[0], s.p[1], s.p[2],
                     s.node, s.size, (unsigned)s.min, (unsigned)s.max, (unsigned)s.isoval, what[s.action < 3 ? s.action : 2]);
    }
    if (r->n_stored < r->n_steps) std::fprintf(f, "... %u more iterations not stored\n", r->n_steps - r->n_stored);
    if (path) std::fclose(f);
    return VR_OK;
}
