/* vr_internal.h -- declarations shared by the kernel file and the C-ABI file of libvr_hip.so. */
#ifndef VR_INTERNAL_H
#define VR_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vr.h"
#include "vr_trace_core.h"

/* device pointers of the optional per-pixel taps */
struct VrTapsK {
    uint32_t *cell_flag;
    float *t;
    uint32_t *checked;
    uint32_t *node;
};

enum {
    VR_KERNEL_DEFAULT = 0,
    VR_KERNEL_RESTART_GLOBAL = 1, /* literal root-restart loop, every node read from global memory     */
    VR_KERNEL_RESTART_LDS = 2,    /* root-restart loop, top tree levels staged in LDS                  */
    VR_KERNEL_RESUME = 3,         /* per-pixel threads, path stack in LDS, resume after a MOVE         */
    VR_KERNEL_PIPELINE = 4,       /* compacted ray list + persistent waves with lane refill, one       */
                                  /* traversal per ray and iso value (default for fewer than 8 iso values)  */
    VR_KERNEL_SWEEP = 5,          /* the same machinery with traversals shared between iso values      */
                                  /* (default from 8 iso values on)                                   */
    VR_KERNEL_UNION = 6,          /* one traversal per ray for the whole iso range (falls back to the sweep */
                                  /* kernel with taps or node ranges that are not nested)                   */
    VR_KERNEL_COUNT = 7
};

/* The sweep kernel cuts the iso range into as many blocks as it takes to have this many (ray, iso block) items per
 * lane of the grid: more blocks balance better, and every block re-walks its first traversal.  launch_seq tags the
 * launch in the kernel's report (0 = untagged). */
#define VR_SWEEP_ITEMS_DEFAULT 8u
/* (ray, iso block) items per lane of the grid that the union kernel starts from for long sweeps (tuned per workload); it is what
 * a workload's first launch runs with, whose tiles are not yet in the order of their lengths: fewer, longer items do better
 * there (Skull, first call: 1.40 ms with 2 or 3 items per lane, 1.56 with 4, 1.75 with 6) */
#define VR_UNION_ITEMS_DEFAULT 3u
/* iso values per call from which the default is the kernel with one traversal per ray for all of them */
#define VR_UNION_MIN_ISOS 8u
/* runs of iso values that may leave the union walks of one call and be finished one iso value at a time (a few per thousand
 * rays do); a call that has more is rendered again with the kernel that shares traversals between iso values */
#define VR_UNION_EXCEPT_CAP 32768u

/* Are the density ranges of a prepared tree nested along every path (vr_host.cpp)? */
extern "C" bool vr_host_ranges_nested(const uint32_t *prepared, uint64_t prepared_words);
extern "C" int vr_host_check_kdt_wide_nested(const uint32_t *nodes, uint64_t n_slots, const uint16_t root_dim[3], uint32_t *max_depth_out,
                                             uint64_t *n_nodes_out, bool *nested_out);

/* Device scratch the pipeline variant needs for one call (ray list, tile offsets, queue heads). */
size_t vr_pipeline_workspace_bytes(uint32_t width, uint32_t n_rows, uint32_t iso_count);

/* What the kernel with one traversal per ray keeps from launch to launch of the same call, in a device buffer of its own
 * (lpt_mem of the launchers, vr_pipeline_lpt_bytes long; the context holds one per call it remembers): per tile of 64 rays the
 * length of its longest item, and the order of the tiles made from it (longest first).  lpt_mode of the launchers is 0 (nothing
 * kept, list order; always so without a buffer) or a sum of: */
enum {
    VR_LPT_RECORD = 1, /* the launch keeps the lengths of its items (they start from what is there: zero after a sort or a clear) */
    VR_LPT_SORT = 2,   /* the lengths of an earlier launch of this very call are in place: the pre-pass makes the order from them
                        * and leaves them at zero */
    VR_LPT_USE = 4,    /* the tiles are handed out in the order that is in place (after the sort, if there is one) */
    VR_LPT_CLEAR = 8,  /* the lengths are set to zero before the launch */
    VR_LPT_CHORD = 16  /* a first call: the pre-pass leaves every tile's longest chord through the volume in the (cleared) words,
                        * makes the order from them and leaves them at zero (with VR_LPT_USE: the launch follows it) */
};
#define VR_LPT_STRIDE_WORDS 64u   /* a tile's length lies in a 256-byte line of its own */
#define VR_LPT_MAX_TILES 32768u  /* lists of more tiles are neither measured nor ordered */
size_t vr_pipeline_lpt_bytes(uint32_t width, uint32_t n_rows);

/* Byte offset, inside the pipeline workspace of a call, of the sweep kernel's 40-byte report (vr_sweep_info). */
size_t vr_pipeline_info_offset(uint32_t width, uint32_t n_rows);

/* Enqueues the kernels that render rows [row_begin,row_end) of iso_count frames.  d_nodes_raw is the
 * reference's SlimNode array, d_nodes_prepared the flattened form of vr_host_prepare_kdt.  The
 * dominant traversal kernel is bracketed by ev_begin / ev_end when they are not null.
 * list_state (may be null): in, true = the workspace still holds the ray list of this very view, rows and kind of
 * kernel (the caller's last pipeline launch on it, same stream), so the pre-pass does not make it again; out,
 * true = the launch went through the pipeline and the workspace holds its list now. */
hipError_t vr_launch_render(int variant, hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes_raw,
                            const uint32_t *d_nodes_prepared, uint32_t max_depth, uint32_t iso_begin, uint32_t iso_count,
                            uint32_t row_begin, uint32_t row_end, uint8_t *d_rgb, vr_render_stats *d_stats,
                            const VrTapsK &taps, void *workspace, int n_cus, hipEvent_t ev_begin, hipEvent_t ev_end,
                            uint32_t sweep_items_per_lane, uint32_t launch_seq, bool ranges_nested, int lpt_mode, void *lpt_mem,
                            bool *list_state = nullptr);

/* Once per device, with that device current (vr_ctx_create calls it; safe from several threads): raises the
 * dynamic-LDS ceiling of the persistent kernels, which is a property of the function on the device, and looks
 * up the address of this GPU's copy of the static scratch array. */
hipError_t vr_device_setup(int device);

/* The static scratch array of vr_render_device_noalloc: VR_NOALLOC_SCRATCH_BYTES bytes per GPU, enough for the
 * pipeline workspace of any view of up to VR_NOALLOC_MAX_PIXELS pixels (20 B per pixel for the ray list, up to
 * 2 MiB of tile offsets for one-pixel-wide views, 256 queue heads).  vr_noalloc_scratch returns the device
 * address on `device` (null before vr_device_setup ran there). */
#define VR_NOALLOC_MAX_PIXELS ((size_t)1 << 20)
#define VR_NOALLOC_SCRATCH_BYTES ((size_t)24 << 20)
void *vr_noalloc_scratch(int device);

/* GPU build of the centre-split tree (vr_build.hip): depth of the deepest leaf for a volume of rx x ry x rz
 * voxels, and the two passes that fill 2^(depth+1) raw slots and 2^(depth+1) + 1 prepared words. */
uint32_t vr_center_tree_depth(uint32_t rx, uint32_t ry, uint32_t rz);
hipError_t vr_launch_build_center(hipStream_t stream, const uint8_t *d_vox, uint32_t rx, uint32_t ry, uint32_t rz,
                                  uint32_t depth, uint32_t *d_raw, uint32_t *d_prepared);


/* The span heuristics built on the device (vr_build.hip), level by level.  `place` allocates the flattened array
 * (the C-ABI file places it inside one 4 GiB window). */
hipError_t vr_build_span_tree(hipStream_t stream, const uint8_t *d_vox, uint32_t rx, uint32_t ry, uint32_t rz, int heuristic,
                              uint64_t max_slots, hipError_t (*place)(size_t, void **, uint32_t **), uint32_t **raw_out,
                              void **prepared_raw_out, uint32_t **prepared_out, uint32_t *depth_out, vr_build_stats *stats_out,
                              bool *too_deep);

/* Volumes beyond 256 voxels per axis, on the wide node array: the pipeline with the wide lane code, or the
 * per-pixel kernel when `variant` names one.  `workspace` as for vr_launch_render.
 * vr_launch_build_center with d_prepared = nullptr builds that array (wide node words) instead. */
hipError_t vr_launch_render_wide(int variant, hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes,
                                 uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin, uint32_t row_end, uint8_t *d_rgb,
                                 vr_render_stats *d_stats, const VrTapsK &taps, void *workspace, int n_cus,
                                 hipEvent_t ev_begin, hipEvent_t ev_end, uint32_t sweep_items_per_lane, uint32_t launch_seq,
                                 bool ranges_nested, int lpt_mode, void *lpt_mem, bool *list_state = nullptr);


/* One ray with the literal loop, every iteration recorded (vr_refine.hip): d_steps holds cap records, d_report one. */
hipError_t vr_launch_trace_ray(hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes_raw, uint32_t iso, uint32_t xi,
                               uint32_t yi, vr_ray_step *d_steps, uint32_t cap, vr_ray_report *d_report);

/* n samples each of the lane code's GPU-only fast paths against their portable definitions; *d_bad += mismatches */
hipError_t vr_launch_selftest(hipStream_t stream, uint64_t n, unsigned long long *d_bad);

/* Trilinear refinement (vr_refine.hip): trilerp on n packed argument records, and the post-pass over the hit taps. */
hipError_t vr_launch_trilerp(hipStream_t stream, const float *d_in, uint64_t n, float *d_out);
hipError_t vr_launch_refine(hipStream_t stream, const VrViewK &view, bool wide, const uint8_t *d_vox, uint32_t ry, uint32_t rz,
                            uint32_t row_begin, uint32_t n_rows, uint32_t n_frames, const uint32_t *d_cell_flag, const float *d_t,
                            float *d_density);

#ifdef VR_MEASURE
/* measuring build: one ray walked by one lane with every step stamped (vr_k_ray_timeline) */
hipError_t vr_launch_ray_timeline(hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes_prepared, uint32_t max_depth, uint32_t xi,
                                  uint32_t yi, uint32_t iso, unsigned long long *d_out, uint32_t *d_lat);
#endif

#endif
