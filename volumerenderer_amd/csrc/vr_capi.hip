/*
 * vr_capi.hip -- the extern "C" boundary of libvr_hip.so (declared in include/vr.h): device
 * contexts, uploads of the volume and the SlimKDT node array, and the render entry points that
 * stand in for raytrace.renderImage (src/raytrace.zig:11-106) and the iso loop of the reference
 * driver (src/main.zig:111-138).
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "vr_internal.h"
#include "vr_trace_wide.h"

#define VR_HIP(call)                                           \
    do {                                                       \
        hipError_t e__ = (call);                               \
        if (e__ != hipSuccess) return VR_EHIP_BASE - (int)e__; \
    } while (0)

struct vr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr; /* device-to-host copies of finished frames, next to the rendering of the next ones */
    hipEvent_t chunk_done[4] = { nullptr, nullptr, nullptr, nullptr }; /* one per chunk of a sweep (256 iso values / 64) */
    /* volume (only the tree builder reads it) */
    uint8_t *d_vox = nullptr;
    uint32_t res[3] = { 0, 0, 0 };
    uint8_t *h_vox = nullptr; /* host copy for the host-side builder */
    /* tree */
    uint32_t *d_nodes = nullptr;      /* SlimKDT.nodes as the reference lays them out          */
    uint32_t *d_prepared = nullptr;   /* the validated, flattened form the fast kernels walk   */
    void *d_prepared_raw = nullptr;   /* the allocation d_prepared lies in                     */
    uint64_t n_slots = 0;
    uint32_t max_depth = 0;
    uint32_t root_dim[3] = { 0, 0, 0 };
    bool wide = false; /* d_nodes holds wide node words (a volume beyond 256 voxels per axis); no flattened form */
    int n_cus = 0;
    /* scratch of the pipeline kernels, grown on demand */
    void *d_work = nullptr;
    size_t work_cap = 0;
    /* staging for the host-buffer entry points, grown on demand */
    uint8_t *d_rgb = nullptr;
    size_t rgb_cap = 0;
    vr_render_stats *d_stats = nullptr;
    size_t stats_cap = 0;
    void *d_tap[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t tap_cap[4] = { 0, 0, 0, 0 };
    uint32_t last_ipl = 0; /* items per lane of the last sweep-kernel launch (0: one traversal per ray, nothing to cut) */
    bool ranges_nested = true; /* node ranges nested along every path: built trees always, uploaded ones when checked so */
    size_t info_off = 0; /* where the last sweep-kernel launch in d_work left its report; 0 = none yet */
    const void *order_flag = nullptr; /* where its pre-pass said whether it ordered the tiles; null = it was not asked to */
    /* The sweep kernels' cut of the iso range, tuned per workload from their own reports.  Every sweep launch in the
     * context's scratch is tagged and followed by an asynchronous copy of its report into page-locked memory; a later
     * launch reads whatever report has arrived (it never waits) and credits it to the workload that launched it.  Fewer,
     * longer items share more of their walks and balance worse; where the best trade lies depends on the volume.  Single
     * launches scatter by several per cent (which CU gets which item is decided by atomics), so a cut is judged by the
     * shortest of a few launches (three; two for the kernel with one traversal per ray, whose cut matters less).  For a new
     * workload the base cut runs first (8 items per lane for long sweeps, 3 for the union kernel, fewer for short sweeps:
     * sweep_cut_base; the very first launch, on cold caches, does not count), then the next smaller cut (3/4 of the base)
     * and the next larger one (2x), and on from the better of the two (1/2, 1/4 or 4x, 8x) while it pays.  A candidate gets
     * exactly its launches, all other launches use the best cut known; it stays when it was at least 1.5 % shorter.  Results
     * never depend on the cut.
     * A workload is what decides the cut: tree, image size, number of rows and of iso values, kernel -- not the camera, the
     * first row or the first iso value, so that the chunks of one sweep and the views of one scene share what was learnt.
     * The context remembers the last kRemember workloads. */
    struct SweepTune {
        uint32_t seq = 0;        /* launches of this workload so far */
        uint32_t base = VR_SWEEP_ITEMS_DEFAULT, best_ipl = VR_SWEEP_ITEMS_DEFAULT;
        uint64_t best_span = 0;
        int dir = 0, step = 0;   /* candidate being tried: sweep_cut_candidate(base, dir, step) */
        bool down_better = false; /* the first step down beat the base */
        int phase = 0;           /* 0 measuring the default, 1 trying candidates, 2 settled */
        int samples = 3;         /* launches a cut is judged by */
        int n_launched = 0, n_reports = 0; /* of the cut being measured */
        uint32_t last_seq = 0;             /* its newest launch */
        uint64_t cur_span = ~0ull;         /* its shortest launch so far */
    };
    static constexpr int kRemember = 8;
    struct TuneEntry {
        uint64_t key = 0, used = 0; /* key 0: free */
        SweepTune t;
    } tunes[kRemember];
    /* the launches whose reports may still arrive: tag -> workload, its launch number, the cut it ran with */
    struct Sent {
        uint32_t tag = 0, local_seq = 0, ipl = 0;
        int slot = -1;
        uint64_t key = 0;
    } sent[16];
    uint32_t tag = 0, tag_seen = 0;
    int cur_tune = -1; /* the workload of the last sweep launch */
    /* What the kernel with one traversal per ray leaves for the next launch of the same CALL (tree, view with its camera,
     * rows, iso range): per tile of 64 rays the length of the tile's longest item, in a device buffer of the call's own, from
     * which the next launch starts its tiles longest first.  The context remembers the last kRemember calls, so a caller that
     * alternates between a few views, or renders a sweep in chunks, keeps every one of them in its fast order. */
    struct LptEntry {
        bool have = false;
        uint64_t used = 0;
        uint64_t tree_id = 0;
        VrViewK view;
        uint32_t iso_begin = 0, iso_count = 0, row_begin = 0, row_end = 0;
        /* what the buffer holds: lengths kept by a launch with this cut of the iso range (0 = none, the words are zero), and an
         * order made from the lengths of a launch with this cut (0 = none).  The lengths of a call repeat from launch to launch,
         * so once the order comes from the cut in use nothing is kept or sorted any more. */
        uint32_t lengths_ipl = 0, order_ipl = 0;
        void *mem = nullptr;
        size_t cap = 0;
    } lpts[kRemember];
    uint64_t use_clock = 0;
    uint64_t tree_id = 0;                 /* changes with every upload / build */
    /* The ray list that the last pipeline launch left in d_work: it depends on the view (camera, image size, the volume's box),
     * the rows and whether the kernel wants rays from pixel centres outwards -- not on the iso values or the tree -- so the next
     * call with the same ones on the same stream does not make it again (an iso slider at a fixed camera, a repeated sweep:
     * two launches fewer per call). */
    struct ListKey {
        bool valid = false;
        VrViewK view;
        uint32_t row_begin = 0, row_end = 0;
        bool centered = false;
        hipStream_t stream = nullptr;
    } list;
    unsigned long long *h_info = nullptr; /* page-locked: walks, items, 3 ticks, sequence number */
    /* options */
    int variant = VR_KERNEL_DEFAULT;
    int timing = 0;
    std::vector<hipEvent_t> ev; /* start/stop pairs of the timed launches not yet read back */
    size_t ev_used = 0;
};

namespace {

int ensure(void **p, size_t *cap, size_t need)
{
    if (*cap >= need) return VR_OK;
    if (*p) { VR_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    VR_HIP(hipMalloc(p, need));
    *cap = need;
    return VR_OK;
}

/* Device array that does not cross a 4 GiB address boundary (the persistent kernel adds 32-bit offsets to
 * the low word of its address).  Always possible up to 2 GiB; beyond that the array is taken as it comes and
 * the launcher falls back to a kernel with 64-bit addressing if it does cross. */
hipError_t alloc_within_4g(size_t bytes, void **raw, uint32_t **use)
{
    hipError_t e = hipMalloc(raw, bytes);
    if (e != hipSuccess) return e;
    uintptr_t p = (uintptr_t)*raw;
    if ((p >> 32) == ((p + bytes - 1) >> 32) || bytes > ((size_t)2 << 30)) {
        *use = (uint32_t *)p;
        return hipSuccess;
    }
    (void)hipFree(*raw);
    *raw = nullptr;
    if ((e = hipMalloc(raw, 2 * bytes)) != hipSuccess) return e;
    p = (uintptr_t)*raw;
    const uintptr_t boundary = ((p >> 32) + 1) << 32; /* if [p, p + bytes) crosses it, [boundary, boundary + bytes) fits */
    *use = (uint32_t *)((p >> 32) == ((p + bytes - 1) >> 32) ? p : boundary);
    return hipSuccess;
}

void free_prepared(vr_ctx *c)
{
    if (c->d_prepared_raw) (void)hipFree(c->d_prepared_raw);
    c->d_prepared_raw = nullptr;
    c->d_prepared = nullptr;
}

/* the per-image constants of renderImage, src/raytrace.zig:22-24,38; computed in strict f32 */
VrViewK make_view(const vr_ctx *ctx, const vr_view *view)
{
    VrViewK k;
    k.ox = view->src[0]; k.oy = view->src[1]; k.oz = view->src[2];
    k.fx = view->forw[0]; k.fy = view->forw[1]; k.fz = view->forw[2];
    k.ux = view->up[0]; k.uy = view->up[1]; k.uz = view->up[2];
    /* right = forw.cross(up), src/geo.zig:115-121 */
    k.rx = k.fy * k.uz - k.fz * k.uy;
    k.ry = k.fz * k.ux - k.fx * k.uz;
    k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = view->viewport_width / (float)view->width;
    k.half_w = (float)(view->width / 2);
    k.half_h = (float)(view->height / 2);
    k.width = view->width;
    k.height = view->height;
    k.rdx = ctx->root_dim[0]; k.rdy = ctx->root_dim[1]; k.rdz = ctx->root_dim[2];
    k.root_hi = vr_pack10(k.rdx, k.rdy, k.rdz);
    k.n_slots = ctx->n_slots;
    return k;
}

int check_view(const vr_view *view)
{
    if (!view || view->width == 0 || view->height == 0) return VR_EARG;
    if ((uint64_t)view->width * view->height > ((uint64_t)1 << 31)) return VR_EARG;
    return VR_OK;
}

} // namespace

extern "C" {

int vr_alloc_host(size_t bytes, void **out)
{
    if (!out || !bytes) return VR_EARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VR_ENODEVICE;
    VR_HIP(hipHostMalloc(out, bytes, hipHostMallocPortable)); /* page-locked for every device of the process */
    return VR_OK;
}

void vr_free_host(void *p)
{
    if (p) (void)hipHostFree(p);
}

int vr_ctx_create(int device, vr_ctx **out)
{
    if (!out) return VR_EARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VR_ENODEVICE;
    if (device < 0 || device >= n) return VR_EARG;
    hipDeviceProp_t prop;
    VR_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VR_ENODEVICE;
    vr_ctx *c = new (std::nothrow) vr_ctx();
    if (!c) return VR_ENOMEM;
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = vr_device_setup(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->chunk_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_info, 64, hipHostMallocDefault);
    if (e == hipSuccess) std::memset(c->h_info, 0, 64);
    if (e != hipSuccess) {
        vr_ctx_destroy(c);
        return VR_EHIP_BASE - (int)e;
    }
    *out = c;
    return VR_OK;
}

void vr_ctx_destroy(vr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 4; i++)
        if (c->chunk_done[i]) (void)hipEventDestroy(c->chunk_done[i]);
    if (c->h_info) (void)hipHostFree(c->h_info);
    if (c->d_vox) (void)hipFree(c->d_vox);
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    free_prepared(c);
    if (c->d_work) (void)hipFree(c->d_work);
    for (vr_ctx::LptEntry &l : c->lpts)
        if (l.mem) (void)hipFree(l.mem);
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    if (c->d_stats) (void)hipFree(c->d_stats);
    for (int i = 0; i < 4; i++)
        if (c->d_tap[i]) (void)hipFree(c->d_tap[i]);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    std::free(c->h_vox);
    delete c;
}

int vr_upload_volume(vr_ctx *c, const uint8_t *vox, uint32_t rx, uint32_t ry, uint32_t rz)
{
    if (!c || !vox || rx < 2 || ry < 2 || rz < 2) return VR_EARG;
    if (rx > VR_WIDE_MAX_RES || ry > VR_WIDE_MAX_RES || rz > VR_WIDE_MAX_RES) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)rx * ry * rz;
    if (c->d_vox) { VR_HIP(hipFree(c->d_vox)); c->d_vox = nullptr; }
    std::free(c->h_vox);
    c->h_vox = nullptr;
    c->res[0] = c->res[1] = c->res[2] = 0;
    if (rx <= 256 && ry <= 256 && rz <= 256) { /* the host builder of the span heuristics reads this copy */
        c->h_vox = (uint8_t *)std::malloc(n);
        if (!c->h_vox) return VR_ENOMEM;
        std::memcpy(c->h_vox, vox, n);
    }
    VR_HIP(hipMalloc((void **)&c->d_vox, n));
    VR_HIP(hipMemcpy(c->d_vox, vox, n, hipMemcpyHostToDevice));
    c->res[0] = rx; c->res[1] = ry; c->res[2] = rz;
    return VR_OK;
}

int vr_upload_kdt(vr_ctx *c, const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3])
{
    if (!c || !nodes || !n_slots || !root_dim) return VR_EARG;
    if (n_slots > ((uint64_t)1 << 32)) return VR_EARG; /* heap indices are u32, src/SlimKDT.zig:24-34 */
    if (root_dim[0] == 0 || root_dim[1] == 0 || root_dim[2] == 0) return VR_EARG;
    /* validate and flatten on the host before anything reaches the device */
    uint32_t max_depth = 0;
    const uint64_t prep_words = vr_prepared_words(n_slots);
    uint32_t *prepared = (uint32_t *)std::malloc(prep_words * sizeof(uint32_t));
    if (!prepared) return VR_ENOMEM;
    int rc = vr_host_prepare_kdt(nodes, n_slots, root_dim, prepared, prep_words, &max_depth, nullptr);
    if (rc != VR_OK) { std::free(prepared); return rc; }
    const bool nested = vr_host_ranges_nested(prepared, prep_words);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess && c->d_nodes) { e = hipFree(c->d_nodes); c->d_nodes = nullptr; }
    free_prepared(c);
    c->n_slots = 0;
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_nodes, n_slots * sizeof(uint32_t));
    if (e == hipSuccess) e = alloc_within_4g(prep_words * sizeof(uint32_t), &c->d_prepared_raw, &c->d_prepared);
    if (e == hipSuccess) e = hipMemcpy(c->d_nodes, nodes, n_slots * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_prepared, prepared, prep_words * sizeof(uint32_t), hipMemcpyHostToDevice);
    std::free(prepared);
    if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
    c->n_slots = n_slots;
    c->tree_id++;
    c->max_depth = max_depth;
    c->root_dim[0] = root_dim[0]; c->root_dim[1] = root_dim[1]; c->root_dim[2] = root_dim[2];
    c->wide = false;
    c->ranges_nested = nested;
    return VR_OK;
}

int vr_upload_kdt_wide(vr_ctx *c, const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3])
{
    if (!c || !nodes || !n_slots || !root_dim) return VR_EARG;
    if (n_slots > VR_WIDE_MAX_SLOTS) return VR_EARG;
    for (int a = 0; a < 3; a++)
        if (root_dim[a] == 0 || root_dim[a] >= VR_WIDE_MAX_RES) return VR_EARG;
    /* validate on the host before anything reaches the device: the kernels walk the array without checks */
    uint32_t max_depth = 0;
    bool nested = false;
    const int rc = vr_host_check_kdt_wide_nested(nodes, n_slots, root_dim, &max_depth, nullptr, &nested);
    if (rc != VR_OK) return rc;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess && c->d_nodes) { e = hipFree(c->d_nodes); c->d_nodes = nullptr; }
    free_prepared(c);
    c->n_slots = 0;
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_nodes, n_slots * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(c->d_nodes, nodes, n_slots * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
    c->n_slots = n_slots;
    c->tree_id++;
    c->max_depth = max_depth;
    for (int a = 0; a < 3; a++) c->root_dim[a] = root_dim[a];
    c->wide = true;
    c->ranges_nested = nested;
    return VR_OK;
}

int vr_build_kdt(vr_ctx *c, int heuristic, vr_build_stats *stats_out)
{
    if (!c) return VR_EARG;
    if (!c->d_vox) return VR_ESTATE;
    if (heuristic < 0 || heuristic >= VR_H_COUNT) return VR_EARG;
    const bool wide = c->res[0] > 256 || c->res[1] > 256 || c->res[2] > 256;
    if (wide) {
        /* beyond the reference's coordinate range: centre split only, built on the device in wide node words */
        if (heuristic != VR_H_ALWAYS_CENTER) return VR_EARG;
        const uint32_t depth = vr_center_tree_depth(c->res[0], c->res[1], c->res[2]);
        if (depth > VR_WIDE_MAX_DEPTH) return VR_ETOODEEP;
        const uint64_t n_slots = (uint64_t)1 << (depth + 1);
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess && c->d_nodes) { e = hipFree(c->d_nodes); c->d_nodes = nullptr; }
        free_prepared(c);
        c->n_slots = 0;
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_nodes, n_slots * sizeof(uint32_t));
        if (e == hipSuccess) e = vr_launch_build_center(c->stream, c->d_vox, c->res[0], c->res[1], c->res[2], depth, c->d_nodes, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
        c->n_slots = n_slots;
        c->tree_id++;
        c->max_depth = depth;
        for (int a = 0; a < 3; a++) c->root_dim[a] = c->res[a] - 1u;
        c->wide = true;
        c->ranges_nested = true; /* a node's range is the union of its children's */
        if (stats_out) {
            const uint64_t cells = (uint64_t)(c->res[0] - 1u) * (c->res[1] - 1u) * (c->res[2] - 1u);
            stats_out->non_center_partition_count = 0;
            stats_out->layer_count = depth;
            stats_out->number_of_nodes = 2 * cells - 1;
        }
        return VR_OK;
    }
    if (!c->h_vox) return VR_ESTATE;
    if (heuristic != VR_H_ALWAYS_CENTER) {
        /* span heuristics: built on the device level by level, valid by construction */
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess && c->d_nodes) { e = hipFree(c->d_nodes); c->d_nodes = nullptr; }
        free_prepared(c);
        c->n_slots = 0;
        if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
        uint32_t depth = 0;
        bool too_deep = false;
        e = vr_build_span_tree(c->stream, c->d_vox, c->res[0], c->res[1], c->res[2], heuristic, (uint64_t)1 << 32, alloc_within_4g,
                               &c->d_nodes, &c->d_prepared_raw, &c->d_prepared, &depth, stats_out, &too_deep);
        if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
        if (too_deep) return VR_ETOODEEP;
        c->n_slots = (uint64_t)1 << (depth + 1);
        c->tree_id++;
        c->max_depth = depth;
        for (int a = 0; a < 3; a++) c->root_dim[a] = c->res[a] - 1u;
        c->wide = false;
        c->ranges_nested = true; /* a node's range is the union of its children's */
        return VR_OK;
    }
    /* centre split: built on the device, valid by construction (no host walk, no PCIe traffic) */
    const uint32_t depth = vr_center_tree_depth(c->res[0], c->res[1], c->res[2]);
    const uint64_t n_slots = (uint64_t)1 << (depth + 1);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess && c->d_nodes) { e = hipFree(c->d_nodes); c->d_nodes = nullptr; }
    free_prepared(c);
    c->n_slots = 0;
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_nodes, n_slots * sizeof(uint32_t));
    if (e == hipSuccess) e = alloc_within_4g((n_slots + 1) * sizeof(uint32_t), &c->d_prepared_raw, &c->d_prepared);
    if (e == hipSuccess) e = vr_launch_build_center(c->stream, c->d_vox, c->res[0], c->res[1], c->res[2], depth, c->d_nodes, c->d_prepared);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
    c->n_slots = n_slots;
    c->tree_id++;
    c->max_depth = depth;
    for (int a = 0; a < 3; a++) c->root_dim[a] = c->res[a] - 1u;
    c->wide = false;
    c->ranges_nested = true; /* a node's range is the union of its children's */
    if (stats_out) {
        /* a full binary tree over the cells: 2 * cells - 1 nodes; every partition is the centre one */
        const uint64_t cells = (uint64_t)(c->res[0] - 1u) * (c->res[1] - 1u) * (c->res[2] - 1u);
        stats_out->non_center_partition_count = 0;
        stats_out->layer_count = depth;
        stats_out->number_of_nodes = 2 * cells - 1;
    }
    return VR_OK;
}

int vr_download_kdt(vr_ctx *c, uint32_t *nodes_out, uint64_t cap_slots, uint64_t *n_slots_out, uint8_t root_dim_out[3])
{
    if (!c) return VR_EARG;
    if (!c->d_nodes || c->wide) return VR_ESTATE;
    if (n_slots_out) *n_slots_out = c->n_slots;
    if (root_dim_out)
        for (int a = 0; a < 3; a++) root_dim_out[a] = (uint8_t)c->root_dim[a];
    if (nodes_out) {
        if (cap_slots < c->n_slots) return VR_EARG;
        VR_HIP(hipSetDevice(c->device));
        VR_HIP(hipMemcpy(nodes_out, c->d_nodes, c->n_slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return VR_OK;
}

int vr_download_kdt_wide(vr_ctx *c, uint32_t *nodes_out, uint64_t first_slot, uint64_t n_slots, uint64_t *n_slots_out,
                         uint16_t root_dim_out[3])
{
    if (!c) return VR_EARG;
    if (!c->d_nodes || !c->wide) return VR_ESTATE;
    if (n_slots_out) *n_slots_out = c->n_slots;
    if (root_dim_out)
        for (int a = 0; a < 3; a++) root_dim_out[a] = (uint16_t)c->root_dim[a];
    if (nodes_out && n_slots) {
        if (first_slot >= c->n_slots || n_slots > c->n_slots - first_slot) return VR_EARG;
        VR_HIP(hipSetDevice(c->device));
        VR_HIP(hipMemcpy(nodes_out, c->d_nodes + first_slot, n_slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return VR_OK;
}

/* where the search for a workload's cut starts: short sweeps have few distinct walks per ray to share, and every cut costs
 * one more */
static uint32_t sweep_cut_base(uint32_t iso_count, bool union_kernel = false)
{
    if (union_kernel) { /* one traversal per ray and block: blocks only shorten the longest walk and fill the lanes */
        const uint32_t b = iso_count / 64u;
        return b < 1u ? 1u : (b > VR_UNION_ITEMS_DEFAULT ? VR_UNION_ITEMS_DEFAULT : b);
    }
    const uint32_t b = iso_count / 8u;
    return b < 2u ? 2u : (b > VR_SWEEP_ITEMS_DEFAULT ? VR_SWEEP_ITEMS_DEFAULT : b);
}
/* the step-th candidate below (dir < 0: 3/4, 1/2, 1/4 of the base) or above it (2x, 4x, 8x: the time is flat between
 * neighbouring cuts -- the 256^3 stand-in: 3.8 ms at 3 items per lane, 3.5 at 5, 3.4 at 6, 2.9 at 12 -- and a search in
 * small steps stops on the plateau); 0 = none left */
static uint32_t sweep_cut_candidate(uint32_t base, int dir, int step)
{
    static const uint32_t num[2][3] = { { 6u, 4u, 2u }, { 16u, 32u, 64u } };
    if (step > 2 || num[dir < 0 ? 0 : 1][step] == 0u) return 0u;
    const uint32_t c = (base * num[dir < 0 ? 0 : 1][step] + 4u) / 8u;
    const uint32_t prev = step == 0 ? base : (base * num[dir < 0 ? 0 : 1][step - 1] + 4u) / 8u;
    return c == prev || c == 0u ? 0u : c;
}
static bool tune_debug()
{
    static const bool on = std::getenv("VR_TUNE_DEBUG") != nullptr; /* VR_TUNE_DEBUG=1: the decisions, on stderr */
    return on;
}

/* The newest report that has arrived goes to the workload that launched it (see vr_ctx::SweepTune). */
static void sweep_tune_report(vr_ctx *c)
{
    const volatile unsigned long long *h = c->h_info;
    const uint32_t tag = (uint32_t)h[5];
    if (tag == 0u || tag == c->tag_seen) return;
    const unsigned long long t_begin = h[2], t_empty = h[3], t_end = h[4];
    if ((uint32_t)h[5] != tag) return; /* the next report is landing: read it on the next call */
    c->tag_seen = tag;
    const vr_ctx::Sent rec = c->sent[tag & 15u];
    if (rec.tag != tag || rec.slot < 0 || c->tunes[rec.slot].key != rec.key || t_end <= t_begin) return; /* too old to tell whose it was */
    vr_ctx::SweepTune &t = c->tunes[rec.slot].t;
    if (t.phase == 2) return;
    const uint64_t span = t_end - t_begin;
    if (tune_debug())
        std::fprintf(stderr, "[sweep tune] workload %d: report of its launch %u (now at %u): %u items per lane, span %.3f ms, tail %.3f, phase %d\n",
                     rec.slot, rec.local_seq, t.seq, rec.ipl, (double)span / 1e5, (double)(t_end - (t_empty < t_end ? t_empty : t_end)) / (double)span, t.phase);
    const uint32_t under_test = t.phase == 0 ? t.base : sweep_cut_candidate(t.base, t.dir, t.step);
    if (rec.ipl == under_test && rec.local_seq >= 2u) { /* the workload's first launch, on cold caches, does not count */
        t.cur_span = span < t.cur_span ? span : t.cur_span;
        t.n_reports++;
    }
    if (t.phase == 1 && t.n_launched == t.samples && rec.local_seq >= t.last_seq && t.n_reports < t.samples)
        t.n_launched = t.n_reports; /* reports overwritten before they were read (the host ran ahead): launch again */
    if (t.n_reports < t.samples) return;
    if (t.phase == 0) {
        t.best_ipl = under_test; t.best_span = t.cur_span;
        t.dir = sweep_cut_candidate(t.base, -1, 0) ? -1 : 1; t.step = 0;
        t.phase = 1;
    } else {
        const bool better = (double)t.cur_span < 0.985 * (double)t.best_span;
        if (better) { t.best_ipl = under_test; t.best_span = t.cur_span; }
        if (t.step == 0 && t.dir < 0) { /* the first step down is in: the first step up is measured as well */
            t.down_better = better;
            if (sweep_cut_candidate(t.base, 1, 0)) t.dir = 1;
            else if (better) t.step = 1;
            else t.phase = 2;
        } else if (t.step == 0) {       /* both first steps are in: on in the direction of the best one */
            if (better) t.step = 1;
            else if (t.down_better) { t.dir = -1; t.step = 1; }
            else t.phase = 2;
        } else if (better) t.step++;
        else t.phase = 2;
    }
    if (t.phase == 1 && !sweep_cut_candidate(t.base, t.dir, t.step)) t.phase = 2;
    t.n_reports = 0; t.n_launched = 0; t.cur_span = ~0ull;
}

/* Picks the items-per-lane of this launch from what the context knows about its workload, and tags the launch. */
static void sweep_tune_next(vr_ctx *c, const vr_view *view, uint32_t iso_count, uint32_t n_rows, uint32_t band_rows, bool union_kernel,
                            uint32_t *ipl_out, uint32_t *tag_out)
{
    sweep_tune_report(c);
    uint64_t key = 1469598103934665603ull; /* FNV-1a over what decides the cut */
    const uint64_t parts[] = { c->tree_id, view->width, view->height, n_rows, band_rows, iso_count, union_kernel };
    for (uint64_t p : parts) { key ^= p; key *= 1099511628211ull; }
    key |= 1ull; /* 0 means a free slot */
    int slot = -1, oldest = 0;
    for (int i = 0; i < vr_ctx::kRemember; i++) {
        if (c->tunes[i].key == key) slot = i;
        if (c->tunes[i].used < c->tunes[oldest].used) oldest = i;
    }
    if (slot < 0) {
        slot = oldest;
        c->tunes[slot] = vr_ctx::TuneEntry();
        c->tunes[slot].key = key;
        vr_ctx::SweepTune &n = c->tunes[slot].t;
        n.base = n.best_ipl = sweep_cut_base(iso_count, union_kernel);
        n.samples = union_kernel ? 2 : 3;
    }
    c->tunes[slot].used = ++c->use_clock;
    c->cur_tune = slot;
    vr_ctx::SweepTune &t = c->tunes[slot].t;
    uint32_t ipl = t.best_ipl;
    const bool cand = t.phase == 1 && t.n_launched < t.samples;
    if (cand) { ipl = sweep_cut_candidate(t.base, t.dir, t.step); t.n_launched++; }
    t.seq++;
    if (cand) t.last_seq = t.seq;
    if (++c->tag == 0u) c->tag = 1u;
    vr_ctx::Sent &rec = c->sent[c->tag & 15u];
    rec.tag = c->tag; rec.local_seq = t.seq; rec.ipl = ipl; rec.slot = slot; rec.key = key;
    if (tune_debug()) std::fprintf(stderr, "[sweep tune] workload %d launch %u: %u items per lane, phase %d\n", slot, t.seq, ipl, t.phase);
    *ipl_out = ipl;
    *tag_out = c->tag;
}

/* the same view, field by field (the struct has padding; the fifteen floats at its start are compared as bit patterns) */
static bool same_view(const VrViewK &a, const VrViewK &b)
{
    return std::memcmp(&a.ox, &b.ox, 15u * sizeof(float)) == 0 && a.width == b.width && a.height == b.height && a.root_hi == b.root_hi &&
           a.rdx == b.rdx && a.rdy == b.rdy && a.rdz == b.rdz && a.n_slots == b.n_slots && a.band_rows == b.band_rows &&
           a.band_pitch == b.band_pitch;
}

/* Shared body of the two device-resident entry points.  `fixed_scratch` = null: the context's own workspace,
 * grown on demand (the only allocation on that path).  Otherwise the pipeline runs in `fixed_scratch`
 * (fixed_bytes long) and nothing is allocated or created. */
static int render_device(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin,
                         uint32_t row_end, void *d_rgb24, void *d_stats, const vr_taps *d_taps, void *stream,
                         void *fixed_scratch, size_t fixed_bytes, uint32_t band_rows = 0, uint32_t band_pitch = 0)
{
    /* [row_begin, row_end) counts OUTPUT rows from image row row_begin on; with bands they are spread over the image */
    if (!c || !d_rgb24) return VR_EARG;
    int rc = check_view(view);
    if (rc != VR_OK) return rc;
    if (!c->d_nodes) return VR_ESTATE;
    if (iso_count == 0 || iso_count > 65535u || iso_begin > 255u || iso_begin + iso_count > 256u) return VR_EARG;
    if (row_begin >= row_end || row_end > view->height) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    VrViewK k = make_view(c, view);
    k.band_rows = band_rows;
    k.band_pitch = band_pitch;
    VrTapsK taps = { nullptr, nullptr, nullptr, nullptr };
    if (d_taps) {
        taps.cell_flag = d_taps->cell_flag;
        taps.t = d_taps->t;
        taps.checked = d_taps->checked;
        taps.node = d_taps->node;
    }
    const int variant = c->variant;
    const bool per_pixel = variant == VR_KERNEL_RESTART_GLOBAL || variant == VR_KERNEL_RESTART_LDS || variant == VR_KERNEL_RESUME;
    void *work = nullptr;
    if (fixed_scratch) {
        if ((size_t)view->width * (row_end - row_begin) > VR_NOALLOC_MAX_PIXELS) return VR_ELIMIT;
        if (vr_pipeline_workspace_bytes(view->width, row_end - row_begin, iso_count) > fixed_bytes) return VR_ELIMIT;
        work = fixed_scratch;
    } else {
        if (!per_pixel) {
            /* growing the scratch is the only allocation on this path; it happens when a larger view comes */
            const size_t need = vr_pipeline_workspace_bytes(view->width, row_end - row_begin, iso_count);
            if (need > c->work_cap) {
                VR_HIP(hipStreamSynchronize(s));
                c->list.valid = false; /* the list goes with the old buffer */
            }
            rc = ensure(&c->d_work, &c->work_cap, need);
            if (rc != VR_OK) return rc;
        }
        work = c->d_work;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool timing = c->timing && !fixed_scratch; /* the probe may have to create events */
    if (timing) {
        while (c->ev.size() < c->ev_used + 2) {
            hipEvent_t e;
            VR_HIP(hipEventCreate(&e));
            c->ev.push_back(e);
        }
        ev0 = c->ev[c->ev_used];
        ev1 = c->ev[c->ev_used + 1];
    }
    const bool any_tap = taps.cell_flag || taps.t || taps.checked || taps.node;
    /* one traversal per ray for the whole iso range: nothing to cut, nothing to tune (same rule as vr_launch_render) */
    const bool union_kernel = c->ranges_nested && !any_tap &&
                              (variant == VR_KERNEL_UNION || (variant == VR_KERNEL_DEFAULT && iso_count >= VR_UNION_MIN_ISOS));
    const bool sweep_kernel = !union_kernel && (variant == VR_KERNEL_SWEEP || variant == VR_KERNEL_UNION ||
                                                (variant == VR_KERNEL_DEFAULT && iso_count >= 8u));
    uint32_t ipl = sweep_cut_base(iso_count, union_kernel), seq = 0; /* the no-allocation entry point stays on this */
    static const uint32_t forced_ipl = std::getenv("VR_SWEEP_ITEMS") ? (uint32_t)std::atoi(std::getenv("VR_SWEEP_ITEMS")) : 0u; /* for tuning runs */
    if (forced_ipl) ipl = forced_ipl;
    else if ((sweep_kernel || union_kernel) && !fixed_scratch)
        sweep_tune_next(c, view, iso_count, row_end - row_begin, band_rows, union_kernel, &ipl, &seq);
    /* tiles longest first when the context remembers this very call (see vr_ctx::LptEntry) */
    static const bool lpt_off = std::getenv("VR_UNION_LPT") && std::atoi(std::getenv("VR_UNION_LPT")) == 0; /* for A/B runs */
    int lpt_mode = 0;
    vr_ctx::LptEntry *lpt = nullptr;
    if (!fixed_scratch && union_kernel && !lpt_off) {
        int oldest = 0;
        for (int i = 0; i < vr_ctx::kRemember; i++) {
            vr_ctx::LptEntry &u = c->lpts[i];
            if (u.have && u.tree_id == c->tree_id && same_view(u.view, k) && u.iso_begin == iso_begin && u.iso_count == iso_count &&
                u.row_begin == row_begin && u.row_end == row_end)
                lpt = &u;
            if (u.used < c->lpts[oldest].used) oldest = i;
        }
        if (!lpt) { /* a call the context does not remember: the oldest entry makes room (its buffer stays) */
            lpt = &c->lpts[oldest];
            lpt->have = false; lpt->lengths_ipl = 0; lpt->order_ipl = 0;
        }
        const size_t need = vr_pipeline_lpt_bytes(view->width, row_end - row_begin);
        if (need > lpt->cap) {
            /* Every entry of the table gets a buffer of this size now (at most 8 x 8.1 MiB): a caller whose every call is a new one
             * -- a moving camera -- allocates on its first call and never again, like the rest of the context's scratch. */
            VR_HIP(hipStreamSynchronize(s)); /* an earlier launch may still use a buffer */
            c->order_flag = nullptr;         /* it points into one of them */
            for (int i = 0; i < vr_ctx::kRemember; i++) {
                vr_ctx::LptEntry &u = c->lpts[i];
                if (u.cap >= need) continue;
                rc = ensure(&u.mem, &u.cap, need);
                u.have = false; u.lengths_ipl = 0; u.order_ipl = 0;
                if (rc != VR_OK) return rc;
            }
        }
        vr_ctx::LptEntry &u = *lpt;
        if (u.have && u.order_ipl == ipl) {
            lpt_mode = VR_LPT_USE;
        } else if (u.have && u.lengths_ipl == ipl) {
            lpt_mode = VR_LPT_SORT | VR_LPT_USE;
            u.order_ipl = ipl; u.lengths_ipl = 0;
        } else if (u.have && u.lengths_ipl) { /* lengths of another cut: good enough to order by, and this cut's are kept */
            lpt_mode = VR_LPT_SORT | VR_LPT_USE | VR_LPT_RECORD;
            u.order_ipl = u.lengths_ipl; u.lengths_ipl = ipl;
        } else if (u.have && u.order_ipl) {   /* an order from another cut, no lengths */
            lpt_mode = VR_LPT_USE | VR_LPT_RECORD;
            u.lengths_ipl = ipl;
        } else {                               /* a first call: the order comes from the rays' chords through the volume */
            static const bool chord_off = std::getenv("VR_UNION_CHORD") && std::atoi(std::getenv("VR_UNION_CHORD")) == 0; /* A/B runs */
            lpt_mode = VR_LPT_RECORD | VR_LPT_CLEAR | (chord_off ? 0 : VR_LPT_CHORD | VR_LPT_USE);
            u.lengths_ipl = ipl; u.order_ipl = 0;
        }
        u.tree_id = c->tree_id; u.view = k;
        u.iso_begin = iso_begin; u.iso_count = iso_count; u.row_begin = row_begin; u.row_end = row_end;
        u.used = ++c->use_clock;
        u.have = false; /* until the launch is known to be enqueued */
    }
    void *lpt_mem = lpt ? lpt->mem : nullptr;
    /* the ray list of the context's last pipeline launch, when it is this call's (see vr_ctx::ListKey) */
    static const bool keep_list_off = std::getenv("VR_KEEP_LIST") && std::atoi(std::getenv("VR_KEEP_LIST")) == 0; /* for A/B runs */
    const bool own_list = !fixed_scratch && !per_pixel; /* per-pixel kernels have no list and leave the one in d_work alone */
    const bool centered = sweep_kernel || union_kernel;
    bool list_state = own_list && !keep_list_off && c->list.valid && c->list.stream == s && same_view(c->list.view, k) &&
                      c->list.row_begin == row_begin && c->list.row_end == row_end && c->list.centered == centered;
    bool *const list_arg = own_list ? &list_state : nullptr;
    if (own_list) c->list.valid = false; /* until the launch is known to be enqueued */
    if (c->wide)
        VR_HIP(vr_launch_render_wide(variant, s, k, c->d_nodes, iso_begin, iso_count, row_begin, row_end, (uint8_t *)d_rgb24,
                                     (vr_render_stats *)d_stats, taps, work, c->n_cus, ev0, ev1, ipl, seq, c->ranges_nested, lpt_mode, lpt_mem,
                                     list_arg));
    else
        VR_HIP(vr_launch_render(variant, s, k, c->d_nodes, c->d_prepared, c->max_depth, iso_begin, iso_count, row_begin,
                                row_end, (uint8_t *)d_rgb24, (vr_render_stats *)d_stats, taps, work, c->n_cus, ev0, ev1, ipl, seq,
                                c->ranges_nested, lpt_mode, lpt_mem, list_arg));
    if (own_list && list_state) {
        c->list.valid = true; c->list.view = k; c->list.row_begin = row_begin; c->list.row_end = row_end;
        c->list.centered = centered; c->list.stream = s;
    }
    if (lpt) lpt->have = true;
    if (timing) c->ev_used += 2;
    if ((sweep_kernel || union_kernel) && !fixed_scratch) {
        c->info_off = vr_pipeline_info_offset(view->width, row_end - row_begin);
        c->last_ipl = ipl;
        c->order_flag = (lpt_mode & VR_LPT_USE) && lpt_mem ? (const char *)lpt_mem + 4u : nullptr;
        const bool settled = forced_ipl || (c->cur_tune >= 0 && c->tunes[c->cur_tune].t.phase == 2);
        if (!settled) /* the report follows the launch on its stream; a later launch reads it if it has arrived */
            VR_HIP(hipMemcpyAsync(c->h_info, (const char *)c->d_work + c->info_off, 48, hipMemcpyDeviceToHost, s));
    }
    return VR_OK;
}

int vr_render_device(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin,
                     uint32_t row_end, void *d_rgb24, void *d_stats, const vr_taps *d_taps, void *stream)
{
    return render_device(c, view, iso_begin, iso_count, row_begin, row_end, d_rgb24, d_stats, d_taps, stream, nullptr, 0);
}

int vr_render_device_bands(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin,
                           uint32_t band_rows, uint32_t band_pitch, uint32_t n_bands, void *d_rgb24, void *d_stats,
                           const vr_taps *d_taps, void *stream)
{
    if (!view || band_rows == 0 || n_bands == 0 || band_pitch < band_rows) return VR_EARG;
    const uint64_t last = (uint64_t)row_begin + (uint64_t)(n_bands - 1u) * band_pitch; /* first row of the last band */
    if (last >= view->height) return VR_EARG;
    const uint32_t last_rows = view->height - (uint32_t)last < band_rows ? view->height - (uint32_t)last : band_rows;
    const uint64_t n_rows = (uint64_t)(n_bands - 1u) * band_rows + last_rows;
    if (row_begin + n_rows > view->height) return VR_EARG; /* cannot happen with pitch >= rows; keeps the shared checks valid */
    return render_device(c, view, iso_begin, iso_count, row_begin, row_begin + (uint32_t)n_rows, d_rgb24, d_stats, d_taps, stream,
                         nullptr, 0, n_bands > 1u ? band_rows : 0u, band_pitch);
}

int vr_render_device_noalloc(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin,
                             uint32_t row_end, void *d_rgb24, void *d_stats, const vr_taps *d_taps, void *stream)
{
    if (!c) return VR_EARG;
    void *fixed = vr_noalloc_scratch(c->device); /* looked up by vr_ctx_create on this device */
    if (!fixed) return VR_ESTATE;
    return render_device(c, view, iso_begin, iso_count, row_begin, row_end, d_rgb24, d_stats, d_taps, stream, fixed,
                         VR_NOALLOC_SCRATCH_BYTES);
}

/* VR_TRACE_RAY=x,y in the environment: the reference's per-ray debug print (traceRay with dbp = true) for that pixel
 * and the first iso value of the call, to stderr */
static void trace_ray_from_env(vr_ctx *c, const vr_view *view, uint32_t iso)
{
    const char *env = std::getenv("VR_TRACE_RAY");
    if (!env || !*env || c->wide) return;
    unsigned x = 0, y = 0;
    if (std::sscanf(env, "%u,%u", &x, &y) != 2 || x >= view->width || y >= view->height) return;
    const uint32_t cap = 1u << 16;
    std::vector<vr_ray_step> steps(cap);
    vr_ray_report rep;
    if (vr_trace_ray(c, view, (uint8_t)iso, x, y, steps.data(), cap, &rep) == VR_OK) (void)vr_print_ray_trace(&rep, steps.data(), nullptr);
}

static int render_sweep(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint8_t *rgb24_out,
                        vr_render_stats *stats_out, const vr_taps *taps);

int vr_render_sweep(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint8_t *rgb24_out,
                    vr_render_stats *stats_out, const vr_taps *taps)
{
    const int rc = render_sweep(c, view, iso_begin, iso_count, rgb24_out, stats_out, taps);
    if (rc == VR_OK) trace_ray_from_env(c, view, iso_begin);
    return rc;
}

static int render_sweep(vr_ctx *c, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint8_t *rgb24_out,
                        vr_render_stats *stats_out, const vr_taps *taps)
{
    if (!c || !rgb24_out) return VR_EARG;
    int rc = check_view(view);
    if (rc != VR_OK) return rc;
    if (!c->d_nodes) return VR_ESTATE;
    if (iso_count == 0 || iso_begin > 255u || iso_begin + iso_count > 256u) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    const size_t npix = (size_t)view->width * view->height * iso_count;
    rc = ensure((void **)&c->d_rgb, &c->rgb_cap, npix * 3);
    if (rc != VR_OK) return rc;
    rc = ensure((void **)&c->d_stats, &c->stats_cap, sizeof(vr_render_stats) * iso_count);
    if (rc != VR_OK) return rc;
    vr_taps dt = { nullptr, nullptr, nullptr, nullptr };
    void *host_tap[4] = { nullptr, nullptr, nullptr, nullptr };
    if (taps) {
        host_tap[0] = taps->cell_flag; host_tap[1] = taps->t; host_tap[2] = taps->checked; host_tap[3] = taps->node;
        for (int i = 0; i < 4; i++)
            if (host_tap[i]) {
                rc = ensure(&c->d_tap[i], &c->tap_cap[i], npix * 4);
                if (rc != VR_OK) return rc;
            }
        dt.cell_flag = host_tap[0] ? (uint32_t *)c->d_tap[0] : nullptr;
        dt.t = host_tap[1] ? (float *)c->d_tap[1] : nullptr;
        dt.checked = host_tap[2] ? (uint32_t *)c->d_tap[2] : nullptr;
        dt.node = host_tap[3] ? (uint32_t *)c->d_tap[3] : nullptr;
    }
    /* A long sweep without taps into page-locked memory (vr_alloc_host) goes in chunks of iso values: while one chunk's frames cross PCIe on the copy stream,
     * the next chunk renders.  64 iso values per chunk: four such renders (2.4 ms each for Skull at 1024^2, against 4.5 ms
     * for all 256 in one launch -- shorter sweeps share fewer walks) still stay ahead of the copies, and only the first one
     * is not hidden behind them. */
    const uint32_t kChunk = 64;
    const size_t frame_bytes = (size_t)view->width * view->height * 3;
    bool page_locked = false; /* pageable destinations are staged by the runtime: chunks only get in each other's way there */
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, rgb24_out) == hipSuccess) page_locked = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError(); /* an ordinary heap pointer is "invalid value" to this query */
    }
    if (page_locked && !taps && iso_count >= 2 * kChunk && frame_bytes * kChunk >= ((size_t)32 << 20)) {
        /* on any failure inside the loop the copies of earlier chunks may still be writing into the caller's
         * buffer: both streams are drained before the error is returned */
        uint32_t n_chunk = 0;
        hipError_t e = hipSuccess;
        rc = VR_OK;
        for (uint32_t k = 0; k < iso_count && rc == VR_OK && e == hipSuccess; k += kChunk, n_chunk++) {
            const uint32_t n = iso_count - k < kChunk ? iso_count - k : kChunk;
            rc = vr_render_device(c, view, iso_begin + k, n, 0, view->height, c->d_rgb + (size_t)k * frame_bytes,
                                  (char *)c->d_stats + sizeof(vr_render_stats) * k, nullptr, c->stream);
            if (rc != VR_OK) break;
            hipEvent_t ev = c->chunk_done[n_chunk]; /* at most 4 chunks: iso_count <= 256 */
            if ((e = hipEventRecord(ev, c->stream)) != hipSuccess) break;
            if ((e = hipStreamWaitEvent(c->copy_stream, ev, 0)) != hipSuccess) break;
            e = hipMemcpyAsync(rgb24_out + (size_t)k * frame_bytes, c->d_rgb + (size_t)k * frame_bytes, frame_bytes * n,
                               hipMemcpyDeviceToHost, c->copy_stream);
        }
        if (rc == VR_OK && e == hipSuccess && stats_out)
            e = hipMemcpyAsync(stats_out, c->d_stats, sizeof(vr_render_stats) * iso_count, hipMemcpyDeviceToHost, c->stream);
        const hipError_t e1 = hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(c->copy_stream);
        if (rc != VR_OK) return rc;
        if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
        return e == hipSuccess ? VR_OK : VR_EHIP_BASE - (int)e;
    }
    rc = vr_render_device(c, view, iso_begin, iso_count, 0, view->height, c->d_rgb, c->d_stats, &dt, c->stream);
    if (rc != VR_OK) return rc;
    VR_HIP(hipMemcpyAsync(rgb24_out, c->d_rgb, npix * 3, hipMemcpyDeviceToHost, c->stream));
    if (stats_out)
        VR_HIP(hipMemcpyAsync(stats_out, c->d_stats, sizeof(vr_render_stats) * iso_count, hipMemcpyDeviceToHost, c->stream));
    for (int i = 0; i < 4; i++)
        if (host_tap[i]) VR_HIP(hipMemcpyAsync(host_tap[i], c->d_tap[i], npix * 4, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(hipStreamSynchronize(c->stream));
    return VR_OK;
}

int vr_render(vr_ctx *c, const vr_view *view, uint8_t isoval, uint8_t *rgb24_out, vr_render_stats *stats_out,
              const vr_taps *taps)
{
    return vr_render_sweep(c, view, isoval, 1, rgb24_out, stats_out, taps);
}

int vr_mgpu_render_sweep(vr_ctx *const *ctxs, uint32_t n_ctx, const vr_view *view, uint32_t iso_begin, uint32_t iso_count,
                         int shard, uint8_t *rgb24_out, vr_render_stats *stats_out)
{
    if (!ctxs || n_ctx == 0 || n_ctx > 64u || !rgb24_out) return VR_EARG;
    if (shard != VR_SHARD_ISO && shard != VR_SHARD_ROWS) return VR_EARG;
    int rc = check_view(view);
    if (rc != VR_OK) return rc;
    if (iso_count == 0 || iso_begin > 255u || iso_begin + iso_count > 256u) return VR_EARG;
    for (uint32_t i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return VR_EARG;
        if (!ctxs[i]->d_nodes) return VR_ESTATE;
        for (uint32_t j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return VR_EARG; /* a context renders one part at a time */
    }
    const size_t row_bytes = (size_t)view->width * 3u, frame_bytes = row_bytes * view->height;
    const uint32_t total = shard == VR_SHARD_ISO ? iso_count : view->height;
    struct Part { uint32_t iso_b, iso_n, r0, r1; };
    Part part[64];
    /* enqueue everything first, so that all devices work at the same time; after a failure every context that has
     * work in flight is drained before the error is returned (its copies write into the caller's buffers) */
    auto step = [](hipError_t e) { return e == hipSuccess ? VR_OK : VR_EHIP_BASE - (int)e; };
    uint32_t n_enqueued = 0;
    rc = VR_OK;
    for (uint32_t i = 0; i < n_ctx && rc == VR_OK; i++) {
        const uint32_t base = total / n_ctx, rem = total % n_ctx;
        const uint32_t begin = i * base + (i < rem ? i : rem), count = base + (i < rem ? 1u : 0u);
        Part &p = part[i];
        if (shard == VR_SHARD_ISO) { p.iso_b = iso_begin + begin; p.iso_n = count; p.r0 = 0; p.r1 = view->height; }
        else { p.iso_b = iso_begin; p.iso_n = iso_count; p.r0 = begin; p.r1 = begin + count; }
        n_enqueued = i + 1u;
        if (count == 0) continue; /* more contexts than frames / rows */
        vr_ctx *c = ctxs[i];
        const size_t part_frame = row_bytes * (p.r1 - p.r0);
        if ((rc = step(hipSetDevice(c->device))) != VR_OK) break;
        if ((rc = ensure((void **)&c->d_rgb, &c->rgb_cap, part_frame * p.iso_n)) != VR_OK) break;
        if ((rc = ensure((void **)&c->d_stats, &c->stats_cap, sizeof(vr_render_stats) * p.iso_n)) != VR_OK) break;
        if ((rc = vr_render_device(c, view, p.iso_b, p.iso_n, p.r0, p.r1, c->d_rgb, c->d_stats, nullptr, c->stream)) != VR_OK) break;
        /* frame k of the part: rows [r0, r1) of frame (iso_b - iso_begin + k) of the output */
        uint8_t *dst = rgb24_out + (size_t)(p.iso_b - iso_begin) * frame_bytes + (size_t)p.r0 * row_bytes;
        rc = step(hipMemcpy2DAsync(dst, frame_bytes, c->d_rgb, part_frame, part_frame, p.iso_n, hipMemcpyDeviceToHost, c->stream));
    }
    if (stats_out) std::memset(stats_out, 0, sizeof(vr_render_stats) * iso_count);
    std::vector<vr_render_stats> tmp;
    for (uint32_t i = 0; i < n_enqueued; i++) {
        const Part &p = part[i];
        if ((shard == VR_SHARD_ISO ? p.iso_n : p.r1 - p.r0) == 0) continue;
        vr_ctx *c = ctxs[i];
        int rc_i = step(hipSetDevice(c->device));
        if (rc_i == VR_OK && rc == VR_OK && stats_out) {
            tmp.resize(p.iso_n);
            rc_i = step(hipMemcpyAsync(tmp.data(), c->d_stats, sizeof(vr_render_stats) * p.iso_n, hipMemcpyDeviceToHost, c->stream));
        }
        const int rc_s = step(hipStreamSynchronize(c->stream)); /* always: nothing of this call stays in flight */
        if (rc_i == VR_OK) rc_i = rc_s;
        if (rc == VR_OK) rc = rc_i;
        if (rc == VR_OK && stats_out)
            for (uint32_t k = 0; k < p.iso_n; k++) {
                vr_render_stats &o = stats_out[p.iso_b - iso_begin + k];
                const vr_render_stats &s = tmp[k];
                o.nodes_checked += s.nodes_checked; o.hit_count += s.hit_count; o.miss_count += s.miss_count;
                o.oob_count += s.oob_count; o.escape_help_events += s.escape_help_events; o.moves += s.moves;
                o.errors += s.errors;
            }
    }
    return rc;
}

int vr_trace_ray(vr_ctx *c, const vr_view *view, uint8_t isoval, uint32_t xi, uint32_t yi, vr_ray_step *steps_out, uint32_t cap,
                 vr_ray_report *report_out)
{
    if (!c || !report_out || (cap && !steps_out)) return VR_EARG;
    int rc = check_view(view);
    if (rc != VR_OK) return rc;
    if (xi >= view->width || yi >= view->height) return VR_EARG;
    if (!c->d_nodes || c->wide) return VR_ESTATE;
    VR_HIP(hipSetDevice(c->device));
    const VrViewK k = make_view(c, view);
    vr_ray_step *d_steps = nullptr;
    vr_ray_report *d_report = nullptr;
    hipError_t e = hipMalloc((void **)&d_report, sizeof(vr_ray_report));
    if (e == hipSuccess && cap) e = hipMalloc((void **)&d_steps, sizeof(vr_ray_step) * (size_t)cap);
    if (e == hipSuccess) e = vr_launch_trace_ray(c->stream, k, c->d_nodes, isoval, xi, yi, d_steps, cap, d_report);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(report_out, d_report, sizeof(vr_ray_report), hipMemcpyDeviceToHost);
    if (e == hipSuccess && cap && report_out->n_stored)
        e = hipMemcpy(steps_out, d_steps, sizeof(vr_ray_step) * (size_t)report_out->n_stored, hipMemcpyDeviceToHost);
    if (d_steps) (void)hipFree(d_steps);
    if (d_report) (void)hipFree(d_report);
    return e == hipSuccess ? VR_OK : VR_EHIP_BASE - (int)e;
}

int vr_selftest(vr_ctx *c, uint64_t n_samples, uint64_t *mismatches_out)
{
    if (!c || !mismatches_out || !n_samples) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    unsigned long long *d_bad = nullptr, bad[3] = { 0, 0, 0 }; /* floor, carry, division */
    hipError_t e = hipMalloc((void **)&d_bad, sizeof bad);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof bad, c->stream);
    if (e == hipSuccess) e = vr_launch_selftest(c->stream, n_samples, d_bad);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
    if (d_bad) (void)hipFree(d_bad);
    if (e != hipSuccess) return VR_EHIP_BASE - (int)e;
    for (int i = 0; i < 3; i++) mismatches_out[i] = bad[i];
    return VR_OK;
}

int vr_trilerp_device(vr_ctx *c, const float *d_in, uint64_t n, float *d_out, void *stream)
{
    if (!c || (n && (!d_in || !d_out))) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    VR_HIP(vr_launch_trilerp(stream ? (hipStream_t)stream : c->stream, d_in, n, d_out));
    return VR_OK;
}

int vr_refine_device(vr_ctx *c, const vr_view *view, uint32_t n_frames, uint32_t row_begin, uint32_t row_end,
                     const uint32_t *d_cell_flag, const float *d_t, float *d_density, void *stream)
{
    if (!c || !d_cell_flag || !d_t || !d_density || n_frames == 0) return VR_EARG;
    int rc = check_view(view);
    if (rc != VR_OK) return rc;
    if (row_begin >= row_end || row_end > view->height) return VR_EARG;
    if (!c->d_vox || !c->d_nodes) return VR_ESTATE; /* the corner voxels come from the uploaded volume */
    for (int a = 0; a < 3; a++)
        if (c->root_dim[a] + 1u != c->res[a]) return VR_ESTATE; /* the tree on the context is of another volume */
    VR_HIP(hipSetDevice(c->device));
    const VrViewK k = make_view(c, view);
    VR_HIP(vr_launch_refine(stream ? (hipStream_t)stream : c->stream, k, c->wide, c->d_vox, c->res[1], c->res[2], row_begin,
                            row_end - row_begin, n_frames, d_cell_flag, d_t, d_density));
    return VR_OK;
}

int vr_sync(vr_ctx *c)
{
    if (!c) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    VR_HIP(hipStreamSynchronize(c->stream));
    return VR_OK;
}

int vr_sweep_info_read(vr_ctx *c, vr_sweep_info *out)
{
    if (!c || !out) return VR_EARG;
    if (!c->d_work || !c->info_off) return VR_ESTATE;
    VR_HIP(hipSetDevice(c->device));
    VR_HIP(hipStreamSynchronize(c->stream));
    VR_HIP(hipMemcpy(out, (const char *)c->d_work + c->info_off, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint32_t runs_redo[2] = { 0u, 0u }; /* the union kernel's count of runs on its list, and its request to render the call again */
    VR_HIP(hipMemcpy(runs_redo, (const char *)c->d_work + c->info_off + 64, sizeof runs_redo, hipMemcpyDeviceToHost));
    out->list_runs = runs_redo[0];
    out->rendered_again = runs_redo[1] ? 1u : 0u;
    out->reserved = 0u;
    out->tiles_longest_first = 0;
    if (c->order_flag) { /* 1 = ordered by item lengths; 2 (a first call's order by chords) is not reported as that */
        uint32_t made_from = 0;
        VR_HIP(hipMemcpy(&made_from, c->order_flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
        out->tiles_longest_first = made_from == 1u ? 1u : 0u;
    }
    out->items_per_lane = c->last_ipl;
    out->cut_settled = c->cur_tune >= 0 && c->tunes[c->cur_tune].t.phase == 2 ? 1u : 0u;
    return VR_OK;
}

int vr_set_timing(vr_ctx *c, int enabled)
{
    if (!c) return VR_EARG;
    c->timing = enabled ? 1 : 0;
    return VR_OK;
}

int vr_timing_read(vr_ctx *c, double *total_ms_out, uint32_t *launches_out)
{
    if (!c || !total_ms_out || !launches_out) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    double total = 0.0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.0f;
        VR_HIP(hipEventSynchronize(c->ev[i + 1]));
        VR_HIP(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        total += ms;
    }
    *total_ms_out = total;
    *launches_out = (uint32_t)(c->ev_used / 2);
    c->ev_used = 0;
    return VR_OK;
}

int vr_set_kernel(vr_ctx *c, int variant)
{
    if (!c || variant < 0 || variant >= VR_KERNEL_COUNT) return VR_EARG;
    c->variant = variant;
    return VR_OK;
}

const char *vr_kernel_name(int variant)
{
    switch (variant) {
    case VR_KERNEL_DEFAULT: return "default";
    case VR_KERNEL_RESTART_GLOBAL: return "restart-global";
    case VR_KERNEL_RESTART_LDS: return "restart-lds";
    case VR_KERNEL_RESUME: return "resume";
    case VR_KERNEL_PIPELINE: return "pipeline";
    case VR_KERNEL_SWEEP: return "sweep";
    case VR_KERNEL_UNION: return "union";
    default: return nullptr;
    }
}

#ifdef VR_MEASURE
/* measuring build only (not declared in vr.h): the step-by-step timeline of one ray; out = 25 u64, lat = 4096 u32 (host) */
int vr_measure_ray_timeline(vr_ctx *c, const vr_view *view, uint8_t isoval, uint32_t xi, uint32_t yi, unsigned long long *out, uint32_t *lat)
{
    if (!c || !out || !lat || check_view(view) != VR_OK || !c->d_prepared || c->wide) return VR_EARG;
    VR_HIP(hipSetDevice(c->device));
    const VrViewK k = make_view(c, view);
    unsigned long long *d_out = nullptr;
    uint32_t *d_lat = nullptr;
    hipError_t e = hipMalloc((void **)&d_out, 25 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&d_lat, 4096 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, 25 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = vr_launch_ray_timeline(c->stream, k, c->d_prepared, c->max_depth, xi, yi, isoval, d_out, d_lat);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 25 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(lat, d_lat, 4096 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (d_out) (void)hipFree(d_out);
    if (d_lat) (void)hipFree(d_lat);
    return e == hipSuccess ? VR_OK : VR_EHIP_BASE - (int)e;
}
#endif

} // extern "C"
