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
    VR_KERNEL_RESUME = 3,         /* path stack in LDS, resume from the lowest ancestor after a MOVE   */
    VR_KERNEL_COUNT = 4
};

/* Enqueues the traversal kernel: rows [row_begin,row_end) of iso_count frames. */
hipError_t vr_launch_render(int variant, hipStream_t stream, const VrViewK &view, const uint32_t *d_nodes,
                            uint32_t iso_begin, uint32_t iso_count, uint32_t row_begin, uint32_t row_end,
                            uint8_t *d_rgb, vr_render_stats *d_stats, const VrTapsK &taps);

#endif
