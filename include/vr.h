/*
 * vr.h -- C ABI of libvr_hip.so: the MI355X (gfx950) implementation of the reference's
 * isosurface ray-casting path.
 *
 * The reference (Zig) has no FFI layer; its boundary for this path is the ordinary call
 *     raytrace.renderImage(src, forw, up, *img, viewport_width, *const SlimKDT, isoval, ...)
 *         !struct { nodes_checked: u64 }                         (src/raytrace.zig:11-21)
 * plus the objects it consumes: Data (src/Data.zig:4-6) and SlimKDT (src/SlimKDT.zig:15-19).
 * Zig's packed Vec3 and zigimg.Image cannot cross a C ABI, so every entry point below takes plain
 * pointers, sizes and scalars.  bindings/vr_hip.zig shows the `extern fn` declarations and the
 * wrapper that keeps the renderImage signature.
 *
 * Conventions: every int-returning function returns VR_OK (0) or a negative error code
 * (vr_strerror names it).  No exceptions, no global state besides contexts.  A context is bound to
 * one device; calls on one context must be serialised by the caller (the reference renderer is
 * single-threaded, src/main.zig:96-163).  Different contexts may be used from different threads.
 */
#ifndef VR_H
#define VR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_OK 0
#define VR_EARG (-1)      /* null pointer, zero size, out-of-range value                          */
#define VR_ENOMEM (-2)    /* host allocation failed                                               */
#define VR_ESTATE (-3)    /* call needs a volume / tree that has not been uploaded or built       */
#define VR_ETOODEEP (-4)  /* tree heap index exceeds the slot limit: the reference's              */
                          /* "FAILD TO BUILD KDT" case (src/main.zig:100-104)                     */
#define VR_EIO (-5)       /* file could not be read                                               */
#define VR_EFORMAT (-6)   /* malformed .vol file                                                  */
#define VR_ENODEVICE (-7) /* no usable HIP device / the device is not gfx950                      */
#define VR_ETREE (-8)     /* uploaded tree is malformed: a reachable node lies outside the array,  */
                          /* or an inner node's partition is not strictly inside its box          */
#define VR_ELIMIT (-9)    /* the call is beyond a fixed limit of the entry point (vr_render_device_noalloc)  */
#define VR_EHIP_BASE (-1000) /* VR_EHIP_BASE - hipError_t                                         */

/* Per-pixel flag stored in bits 24..31 of a cell/flag tap word. */
#define VR_FLAG_MISS 0u /* ray entered the volume and left it without a hit (src/raytrace.zig:66-68) */
#define VR_FLAG_HIT 1u  /* src/raytrace.zig:56-65                                                    */
#define VR_FLAG_OOB 2u  /* ray never entered the root box (src/raytrace.zig:148-156, 53-55)          */

/* Tree-partition heuristics, src/SlimKDT.zig:327-598 (ids are this library's). */
enum vr_heuristic {
    VR_H_ALWAYS_CENTER = 0,                 /* AlwaysCenterH              :359-367 */
    VR_H_MINIMIZE_SPAN = 1,                 /* MinimizeSpanHeuristic      :327-357 */
    VR_H_FLEXIBLE_FIRST5 = 2,               /* FlexibleFirst5             :369-375 */
    VR_H_FLEXIBLE_FIRST10 = 3,              /* FlexibleFirst10            :377-383 */
    VR_H_FLEXIBLE_AFTER10 = 4,              /* FlexibleAfter10            :385-391 */
    VR_H_FLEXIBLE_AFTER15 = 5,              /* FlexibleAfter15            :392-398 */
    VR_H_FLEXIBLE_AFTER20 = 6,              /* FlexibleAfter20            :400-406 */
    VR_H_FLEXIBLE_CENTER_WEIGHTED5 = 7,     /* FlexibleCenterWeighted5    :408-438 */
    VR_H_FLEXIBLE_CENTER_WEIGHTED10 = 8,    /* FlexibleCenterWeighted10   :440-470 */
    VR_H_FLEXIBLE_CENTER_WEIGHTED30 = 9,    /* FlexibleCenterWeighted30   :472-502 */
    VR_H_FLEXIBLE_CENTER_WEIGHTED50 = 10,   /* FlexibleCenterWeighted50   :504-534 */
    VR_H_FLEXIBLE_CENTER_WEIGHTED75 = 11,   /* FlexibleCenterWeighted75   :536-566 */
    VR_H_FLEXIBLE_CENTER_WEIGHTED100 = 12,  /* FlexibleCenterWeighted100  :568-598 */
    VR_H_COUNT = 13
};

/* SlimKDT.PartitionBuildData, src/SlimKDT.zig:215-219. */
typedef struct vr_build_stats {
    uint64_t non_center_partition_count;
    uint64_t layer_count;
    uint64_t number_of_nodes;
} vr_build_stats;

/* Counters renderImage keeps (src/raytrace.zig:27-35); only nodes_checked is returned by the
 * reference (:105), the rest are what its debug print shows (:79-104). One record per iso value. */
typedef struct vr_render_stats {
    uint64_t nodes_checked;      /* reference-defined count: one per traceRay loop iteration (:162) */
    uint64_t hit_count;
    uint64_t miss_count;
    uint64_t oob_count;
    uint64_t escape_help_events; /* :241 */
    uint64_t moves;              /* MOVE steps (:199-246); not counted by the reference            */
    uint64_t errors;             /* rays abandoned: on a malformed tree (index or depth out of range), or after 2^23 checks.   */
                                 /* The reference's loop does not end on every ray: a ray that runs along a lattice plane can   */
                                 /* alternate between two states for ever (MOVE takes the exit plane's t, which may lie behind  */
                                 /* the point, src/raytrace.zig:199-205) -- about 1 random far-away view in 100 has such a ray;  */
                                 /* the reference's own views have none.  Here the ray is given up as a miss and counted.        */
} vr_render_stats;

/* Camera + image description: the scalar arguments of renderImage (src/raytrace.zig:12-18). */
typedef struct vr_view {
    float src[3];          /* ray origin                                  */
    float forw[3];         /* view direction (the caller normalises it)   */
    float up[3];           /* up vector, used as given (not normalised)   */
    float viewport_width;  /* width of the image plane at distance 1      */
    uint32_t width;        /* img.width                                   */
    uint32_t height;       /* img.height                                  */
} vr_view;

/* Optional per-pixel parity taps, each `frames * rows * width` elements, any may be NULL.
 * They expose RayRes (src/raytrace.zig:108-115) plus the hit cell, which the reference only holds
 * transiently. */
typedef struct vr_taps {
    uint32_t *cell_flag; /* x | y<<8 | z<<16 | VR_FLAG_*<<24 ; cell is 0 unless the ray hit   */
                         /* (wide trees: x | y<<10 | z<<20 | VR_FLAG_*<<30)                    */
    float *t;            /* RayRes.t ; 0 for OOB rays                                          */
    uint32_t *checked;   /* RayRes.checked_nodes                                               */
    uint32_t *node;      /* heap index of the hit leaf ; 0 unless the ray hit                  */
} vr_taps;

typedef struct vr_ctx vr_ctx;

const char *vr_strerror(int code);

/* ---- host-side producers (no GPU needed) ------------------------------------------------ */

/* Data.loadUCD (src/Data.zig:9-44): reads a .vol file.  *vox_out receives res[0]*res[1]*res[2]
 * bytes, x slowest / z fastest (src/Data.zig:90-92); release with vr_free. */
int vr_load_ucd(const char *path, uint8_t **vox_out, uint32_t res_out[3]);

/* The inverse of vr_load_ucd: writes vox in the .vol layout Data.loadUCD reads (src/Data.zig:9-44). */
int vr_save_ucd(const char *path, const uint8_t *vox, const uint32_t res[3]);

/* Tree cache file: what the reference's saveToFile stub (src/___SlimKDT_.zig:266-269) was meant to
 * be.  Stores the SlimNode array (src/SlimKDT.zig:15-19: root_dim + nodes) without its trailing
 * unoccupied slots, with a checksum; vr_load_kdt returns the full-length array (release with
 * vr_free) ready for vr_upload_kdt.  VR_EFORMAT on a bad magic, size or checksum. */
int vr_save_kdt(const char *path, const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3]);
int vr_load_kdt(const char *path, uint32_t **nodes_out, uint64_t *n_slots_out, uint8_t root_dim_out[3]);

/* SlimKDT.newHeursiticPartition (src/SlimKDT.zig:222-322) on the host.  Produces the heap-indexed
 * (children of i at 2i+1 / 2i+2, :24-30) array of 4-byte SlimNode words (:9-13: byte0 min, byte1
 * max, byte2 partition, byte3 axis).  Slots no node occupies are 0.  max_slots = 0 means 2^32.
 * Returns VR_ETOODEEP where the reference's realloc would fail.  Release *nodes_out with vr_free. */
int vr_host_build_kdt(const uint8_t *vox, const uint32_t res[3], int heuristic, uint64_t max_slots,
                      uint32_t **nodes_out, uint64_t *n_slots_out, uint8_t root_dim_out[3],
                      vr_build_stats *stats_out);

/* The flattening step in front of the GPU.  Walks the heap-indexed SlimNode array from the root with
 * the boxes SpaceTracker would hold (src/SlimKDT.zig:55-103), checks that every reachable node lies
 * inside the array and that every inner node's partition is strictly inside its box (what the
 * reference's builder always produces; anything else never reaches a leaf in the reference's zoom
 * either: VR_ETREE), and writes the array the kernels walk:
 *   word    bits 0-23 unchanged (min, max, partition), bits 24-28 = axis * 10, bit 31 = leaf
 *   layout  the same heap, 1-based: the node at heap index i goes to position i + 1 (children of
 *           position g at 2g and 2g + 1, parent g >> 1, sibling g ^ 1); unoccupied words are 0.
 * prepared_out may be NULL (validate only); otherwise it must hold prepared_words =
 * vr_prepared_words(n_slots) words.  max_depth_out = deepest level (root = 0). */
uint64_t vr_prepared_words(uint64_t n_slots);
int vr_host_prepare_kdt(const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3],
                        uint32_t *prepared_out, uint64_t prepared_words, uint32_t *max_depth_out,
                        uint64_t *n_nodes_out);

/* The validating walk of vr_host_prepare_kdt for a tree in wide node words (volumes beyond the reference's
 * u8 coordinates, see vr_upload_kdt_wide): VR_ETREE unless every reachable node lies inside the array and every
 * inner node splits strictly inside its box.  No flattened copy is made; the kernels walk wide trees as they
 * are.  max_depth_out = deepest level (root = 0; at most 30 with n_slots <= 2^31). */
int vr_host_check_kdt_wide(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3],
                           uint32_t *max_depth_out, uint64_t *n_nodes_out);

/* Camera of the reference driver (src/main.zig:31-43): forw = norm(f32(res/2) - pos),
 * up = (forw x (0,1,0)) x forw (not normalised). */
void vr_main_camera(const uint32_t res[3], const float pos[3], float forw_out[3], float up_out[3]);

void vr_free(void *p);

/* Page-locked host memory for the outputs of vr_render / vr_render_sweep (the reference's image buffer is
 * ordinary heap memory, src/main.zig:112; frames copied into pageable memory cross PCIe several times
 * slower).  No counterpart in the reference.  Needs a GPU; release with vr_free_host. */
int vr_alloc_host(size_t bytes, void **out);
void vr_free_host(void *p);

/* Output side of the reference driver (src/main.zig:116,137 writes renders/out_iso{d}.png through
 * zigimg): an 8-bit RGB PNG (stored deflate blocks, readable by any PNG decoder) or a binary PPM of
 * a finished width*height*3 frame. */
int vr_write_png_rgb24(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb24);
int vr_write_ppm_rgb24(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb24);

/* ---- device contexts -------------------------------------------------------------------- */

/* One context per device.  Fails with VR_ENODEVICE when no HIP device is present or the device is
 * not gfx950: there is no CPU fallback. */
int vr_ctx_create(int device, vr_ctx **out);
void vr_ctx_destroy(vr_ctx *ctx);

/* Uploads Data.data verbatim (src/Data.zig:4-6, layout :90-92).  Needed by vr_build_kdt only:
 * the render path never reads voxels (src/raytrace.zig:117-255 reads kdt.nodes alone).
 * 2..256 voxels per axis is the reference's range; up to 1024 per axis is accepted for wide trees (below). */
int vr_upload_volume(vr_ctx *ctx, const uint8_t *vox, uint32_t rx, uint32_t ry, uint32_t rz);

/* Uploads SlimKDT.nodes verbatim (src/SlimKDT.zig:17) with root_dim (:15).  Bits 26..31 of each
 * word are ignored (undefined in the Zig packed struct).  The array is validated and flattened
 * with vr_host_prepare_kdt first: VR_ETREE for a tree the reference could not traverse either. */
int vr_upload_kdt(vr_ctx *ctx, const uint32_t *nodes, uint64_t n_slots, const uint8_t root_dim[3]);

/* SlimKDT.newHeursiticPartition (src/SlimKDT.zig:222-322) on the uploaded volume, built on the GPU and kept
 * there: the centre split in two data-parallel passes, the span heuristics (:327-598) level by level.  The
 * node array is word for word what vr_host_build_kdt produces (2^(depth+1) slots).  VR_ETOODEEP where a heap
 * index would pass 2^32: the reference's "FAILD TO BUILD KDT" case (src/main.zig:100-104). */
int vr_build_kdt(vr_ctx *ctx, int heuristic, vr_build_stats *stats_out);

/* Copies the device-resident tree back (for parity checks and caching). nodes_out may be NULL to
 * query n_slots only.  VR_ESTATE when the context holds a wide tree (vr_download_kdt_wide). */
int vr_download_kdt(vr_ctx *ctx, uint32_t *nodes_out, uint64_t cap_slots, uint64_t *n_slots_out,
                    uint8_t root_dim_out[3]);

/* ---- volumes beyond the reference's coordinate range ------------------------------------ */

/* The reference keeps coordinates in u8 (SPACESIZE, src/typedef.zig:1) and the partition in one byte of
 * SlimNode (src/SlimKDT.zig:9-13): it ends at 256 voxels per axis.  These entry points have no
 * counterpart in it.  They run the same algorithm (same f32 operations in the same order, same heap
 * layout, src/SlimKDT.zig:24-30) on up to 1024 voxels per axis with a wider node word:
 *     wide node word   bits 0-7 min | 8-15 max | 16-27 partition | 28-29 axis
 * A context holds a wide tree after vr_upload_kdt_wide, or after vr_build_kdt on a volume with an axis
 * above 256 voxels (VR_H_ALWAYS_CENTER only, built on the GPU; 1024^3 voxels give 2^31 slots = 8 GiB).
 * vr_render / vr_render_sweep / vr_render_device then work as for any other tree, with one difference:
 * the cell/flag tap word is  x | y<<10 | z<<20 | VR_FLAG_*<<30.  vr_upload_kdt_wide validates the array with
 * vr_host_check_kdt_wide first (VR_ETREE for a tree the reference's zoom could not traverse either), which is
 * what lets the kernels walk it without per-node checks.  A wide tree has at most 2^31 slots (levels 0..30). */
int vr_upload_kdt_wide(vr_ctx *ctx, const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3]);

/* Copies slots [first_slot, first_slot + n_slots) of the device-resident wide tree back (the whole
 * array can be 8 GiB).  nodes_out may be NULL to query the array length and root_dim only. */
int vr_download_kdt_wide(vr_ctx *ctx, uint32_t *nodes_out, uint64_t first_slot, uint64_t n_slots,
                         uint64_t *n_slots_out, uint16_t root_dim_out[3]);

/* ---- rendering -------------------------------------------------------------------------- */

/* raytrace.renderImage (src/raytrace.zig:11-106) for one iso value into a HOST buffer of
 * width*height*3 bytes (row-major, top row first, RGB8: img.pixels.rgb24).  stats_out and taps
 * (host pointers) may be NULL.  Synchronous. */
int vr_render(vr_ctx *ctx, const vr_view *view, uint8_t isoval, uint8_t *rgb24_out,
              vr_render_stats *stats_out, const vr_taps *taps);

/* The iso loop of the reference driver (src/main.zig:111-138): iso_count consecutive frames
 * starting at iso_begin, frame k at rgb24_out + k*width*height*3; stats_out[k] per frame. Host
 * buffers.  Synchronous. */
int vr_render_sweep(vr_ctx *ctx, const vr_view *view, uint32_t iso_begin, uint32_t iso_count,
                    uint8_t *rgb24_out, vr_render_stats *stats_out, const vr_taps *taps);

/* Device-resident, asynchronous form used for sharded and timed runs: renders image rows
 * [row_begin,row_end) of iso_count frames into DEVICE memory.  Frame k, row r (relative to
 * row_begin) starts at d_rgb24 + ((k*(row_end-row_begin) + r) * width) * 3.  d_stats (device,
 * iso_count records, may be NULL) is zeroed and filled on the stream; a sweep call without it runs a kernel with
 * the per-iso counting compiled out (15 ... 28 % faster: the reference's renderImage returns nodes_checked,
 * src/raytrace.zig:21,105, so a caller that stands in for it passes d_stats).  d_taps members are device
 * pointers.  `stream` is a hipStream_t (NULL = the context's own stream).  Returns after
 * enqueueing.  A context's scratch serves one call at a time: calls on one context that may overlap in time
 * (different streams) are serialised by the caller or made on contexts of their own.  The scratch keeps the
 * call's ray list -- which rays enter the volume: camera, image size and rows decide, not the iso values -- and
 * the next call with the same view and rows on the same stream uses it instead of making it again. */
int vr_render_device(vr_ctx *ctx, const vr_view *view, uint32_t iso_begin, uint32_t iso_count,
                     uint32_t row_begin, uint32_t row_end, void *d_rgb24, void *d_stats,
                     const vr_taps *d_taps, void *stream);

/* vr_render_device on rows spread over the image in bands, for sharding one frame (or one sweep) over GPUs with an
 * even load: the rows that never meet the volume sit at the top and the bottom of the image (src/raytrace.zig:40 walks
 * them in order), so contiguous blocks of rows give the outer ranks nothing to do.  Band j (j < n_bands) = image rows
 * [row_begin + j * band_pitch, + band_rows); the last band is clipped at the bottom edge of the image.  The bands
 * are packed one after the other in the output: with n_rows their total number of rows, frame k, packed row r starts
 * at d_rgb24 + ((k * n_rows + r) * width) * 3, and the taps likewise.  Rank i of n takes row_begin = i * band_rows,
 * band_pitch = n * band_rows.  No counterpart in the reference (one thread renders all rows). */
int vr_render_device_bands(vr_ctx *ctx, const vr_view *view, uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin,
                           uint32_t band_rows, uint32_t band_pitch, uint32_t n_bands, void *d_rgb24, void *d_stats,
                           const vr_taps *d_taps, void *stream);

/* vr_render_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference (its renderImage allocates nothing
 * either, src/raytrace.zig:11-106, but runs on the CPU).  Same arguments, same results as vr_render_device, but
 * the scratch of the pipeline kernels (ray list, tile offsets, queue heads) is a static device array of 24 MiB
 * per GPU instead of a buffer the context grows on demand, and the call makes no hipMalloc / hipFree and creates
 * no events (the kernel-time probe of vr_set_timing does not cover it).
 * Limit: (row_end - row_begin) * width <= 2^20 pixels per call (e.g. one 1024 x 1024 frame range; render larger
 * frames in row bands); beyond it VR_ELIMIT.  Any number of iso values.
 * All contexts on one GPU share the one array: calls of this entry point that overlap in time on one GPU (two
 * streams, two contexts, two threads) race on it and must be serialised by the caller; it does not touch the
 * scratch of vr_render_device, so a call may overlap with calls of the other entry points. */
int vr_render_device_noalloc(vr_ctx *ctx, const vr_view *view, uint32_t iso_begin, uint32_t iso_count,
                             uint32_t row_begin, uint32_t row_end, void *d_rgb24, void *d_stats,
                             const vr_taps *d_taps, void *stream);

/* Self-test of the device: the traversal kernels use three GPU-only fast paths -- floor through one conversion
 * instruction, 2g + bit through the carry of the lane mask, and the exit-plane division of the MOVE
 * (src/geo.zig:406-426) by a per-ray reciprocal with FMA corrections -- whose portable definitions are what the CPU
 * emulation of the lane code runs.  This evaluates each on n_samples pseudo-random operands on the GPU next to its
 * portable definition (the division next to the correctly rounded hardware divide) and counts the results that
 * differ, one count per fast path (floor, carry, division): all 0 on a healthy device.  No counterpart in the
 * reference. */
int vr_selftest(vr_ctx *ctx, uint64_t n_samples, uint64_t mismatches_out[3]);

/* ---- trilinear refinement ---------------------------------------------------------------- */

/* raytrace.trilerp (src/raytrace.zig:257-291) on the GPU, bit for bit (strict f32, the reference's operation
 * order): d_in holds n records of 17 floats -- x, y, z, x0, x1, y0, y1, z0, z1, values[8] (v000, v100, v010, v110,
 * v001, v101, v011, v111) -- and d_out receives n results.  Device pointers; asynchronous on `stream`
 * (NULL = the context's own). */
int vr_trilerp_device(vr_ctx *ctx, const float *d_in, uint64_t n, float *d_out, void *stream);

/* EXPERIMENTAL: no behaviour of the reference to match (its function is an empty stub); the signature may change.
 * The refinement the reference only sketches: trilerpTrace (src/raytrace.zig:295-304) fetches the hit cell's 8
 * corner densities (Data.getCornerDens, src/Data.zig:102-114) and stops there.  For every pixel whose ray hit, this
 * post-pass over the taps of vr_render_device (same view, same rows, n_frames frames) evaluates trilerp at the hit
 * point src + dir * t inside the hit cell and stores the interpolated density; 0 for the other pixels.  It is the
 * one step of the path that reads voxels at render time, so the context needs both the volume
 * (vr_upload_volume) and its tree.  Nothing in the reference pins the result beyond trilerp and getCornerDens
 * themselves: parity unpinned. */
int vr_refine_device(vr_ctx *ctx, const vr_view *view, uint32_t n_frames, uint32_t row_begin, uint32_t row_end,
                     const uint32_t *d_cell_flag, const float *d_t, float *d_density, void *stream);

/* ---- one ray, step by step ------------------------------------------------------------------ */

/* The reference's per-ray debug print (traceRay compiled with dbp = true, src/raytrace.zig:133-146, 164-198) as
 * data: one record per iteration of its loop, in order.  Not covered: what the reference prints inside a MOVE whose point is
 * still inside the box it left ("VOLUME NOT ESQd!!!" with the node box, "previous node not escaped!", "!escape help needed!",
 * src/raytrace.zig:207-240) -- there is no record field for it, so the trace of a ray with escape helps
 * (vr_render_stats.escape_help_events counts them; 0 on every volume tried) lacks those lines. */
#define VR_STEP_ZOOM 0u /* "ZOOM!" :189-192 */
#define VR_STEP_HIT 1u  /* "HIT!"  :179-187 */
#define VR_STEP_MOVE 2u /* "MOVE!" :197-205 */
typedef struct vr_ray_step {
    float t;           /* at the top of the iteration (:167)                      */
    float p[3];        /* ray.point(t)                                            */
    uint32_t node;     /* st.node_idx                                             */
    uint32_t size;     /* st.node_space.size()                                    */
    uint8_t min, max;  /* kdt.nodes[node].density_range                           */
    uint8_t isoval;
    uint8_t action;    /* VR_STEP_*                                               */
} vr_ray_step;
typedef struct vr_ray_report {
    float origin[3], dir[3]; /* the ray of the pixel (src/raytrace.zig:41-45)                                  */
    float t_entry;           /* t after the skip into the volume (:127-131)                                    */
    float t_final;           /* RayRes.t (0 for an OOB ray)                                                    */
    uint32_t flag;           /* VR_FLAG_*                                                                      */
    uint32_t n_steps;        /* iterations the ray took = RayRes.checked_nodes; may exceed the records stored  */
    uint32_t n_stored;       /* records written to steps_out (at most cap)                                     */
} vr_ray_report;

/* Traces the ray of pixel (xi, yi) of `view` for one iso value on the GPU with the reference's own loop (restart at
 * the root after every MOVE, every node read from the SlimNode array as uploaded) and returns its iterations.
 * Host pointers; steps_out may be NULL with cap = 0 to get the report alone.  Trees of vr_upload_kdt / vr_build_kdt
 * on volumes within the reference's range (VR_ESTATE for wide trees).  Synchronous. */
int vr_trace_ray(vr_ctx *ctx, const vr_view *view, uint8_t isoval, uint32_t xi, uint32_t yi, vr_ray_step *steps_out,
                 uint32_t cap, vr_ray_report *report_out);

/* Writes a trace in the layout of the reference's debug print ("### NEW RAY ###", then per iteration
 * "t=...  r=(...)" / "node=...   size=...   iso=[min-max] v iso" / "  res: ZOOM!|HIT!|MOVE!") to `path`
 * (NULL = stderr).  vr_render and vr_render_sweep do this by themselves for the pixel named by the environment
 * variable VR_TRACE_RAY=x,y (first iso value of the call, to stderr). */
int vr_print_ray_trace(const vr_ray_report *report, const vr_ray_step *steps, const char *path);

/* One host thread driving several contexts -- normally one per GPU of the node, each holding the same tree
 * (contexts on one device work too).  The iso loop of src/main.zig:111-138 is split over the contexts with no
 * exchange between them: VR_SHARD_ISO gives context i a contiguous block of the iso values (whole frames),
 * VR_SHARD_ROWS a contiguous block of the rows of every frame; the first (count mod n) contexts get one more.
 * All contexts render at the same time (vr_render_device on their own streams); their parts are copied straight
 * into place in the HOST buffers, which end up exactly as vr_render_sweep on one context leaves them (allocate
 * rgb24_out with vr_alloc_host for overlapping copies).  stats_out (iso_count records, may be NULL) holds the
 * per-iso counters summed over the contexts.  No counterpart in the reference, which is single-threaded on one
 * CPU; for the one-process-per-GPU form use vr_render_device under any launcher (bench.py does). */
#define VR_SHARD_ISO 0
#define VR_SHARD_ROWS 1
int vr_mgpu_render_sweep(vr_ctx *const *ctxs, uint32_t n_ctx, const vr_view *view, uint32_t iso_begin,
                         uint32_t iso_count, int shard, uint8_t *rgb24_out, vr_render_stats *stats_out);

/* Blocks until everything enqueued on the context's own stream has finished. */
int vr_sync(vr_ctx *ctx);

/* Kernel-time probe: while enabled, every vr_render_device call brackets its traversal kernel with
 * a pair of HIP events on the launch stream.  vr_timing_read waits for the pairs recorded since the
 * previous read, returns the sum of their elapsed times and their number, and forgets them. */
int vr_set_timing(vr_ctx *ctx, int enabled);
int vr_timing_read(vr_ctx *ctx, double *total_ms_out, uint32_t *launches_out);

/* What the sweep kernel reports about its last launch on this context through vr_render_device / vr_render_sweep
 * (measurement only, no counterpart in the reference): the traversals it ran (one serves every iso value that
 * cannot change a range test on its way), its work items, and when -- in ticks of the GPU's 100 MHz clock -- the
 * first wave started, the first wave found the queue empty, and the last wave ended.  Waits for the context's own
 * stream; a caller that rendered on a stream of its own synchronises that one first.  VR_ESTATE when no
 * sweep-kernel launch has used the context's scratch yet. */
typedef struct vr_sweep_info {
    uint64_t walks;
    uint64_t items;
    uint64_t tick_begin, tick_queue_empty, tick_end;
    uint32_t items_per_lane; /* the cut of the iso range that launch ran with */
    uint32_t cut_settled;    /* 1 once the context has stopped trying other cuts on this workload */
    uint32_t list_runs;      /* union kernel: runs of iso values that its second kernel had to finish (normally 0) */
    uint32_t tiles_longest_first; /* union kernel: 1 when the launch repeated a call the context remembers (one of its last 8) and
                                   * started its tiles of 64 rays in the order of the item lengths that call left (longest first);
                                   * 0 = a call it has not seen: the tiles in the order of their rays' chords, or list order */
    uint32_t rendered_again; /* union kernel: 1 when it could not finish the call exactly (more than 32 768 runs of iso values left
                              * its walks, or one front passed 2^17 MOVEs) and the call was rendered again, on the same stream and
                              * before anything the caller enqueued behind it, by the kernel that shares traversals between iso
                              * values; frames and counters are the same either way */
    uint32_t reserved;
} vr_sweep_info;
int vr_sweep_info_read(vr_ctx *ctx, vr_sweep_info *out);

/* Traversal kernel selection: 0 = default (fastest verified: from 8 iso values on the sweep
 * kernel, which traces a ray once per interval of iso values that cannot change the outcome of any
 * density-range test and gives every iso value of the interval that result), other values name the
 * variants listed by vr_kernel_name.  All variants produce identical results. */
int vr_set_kernel(vr_ctx *ctx, int variant);
const char *vr_kernel_name(int variant);

#ifdef __cplusplus
}
#endif
#endif /* VR_H */
