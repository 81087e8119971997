/*
 * vr_trilerp.h -- trilinear interpolation inside a cell, the one finished piece of the reference's iso refinement
 * (trilerp, src/raytrace.zig:257-291; its caller trilerpTrace, :295-304, is a stub that fetches the hit cell's
 * corner densities and stops).  Strict f32, the reference's operations in the reference's order; compiled for the
 * GPU by the library and for the host by tests/emu.
 */
#ifndef VR_TRILERP_H
#define VR_TRILERP_H

#include "vr_trace_core.h"

/* values = v000, v100, v010, v110, v001, v101, v011, v111 (x fastest), src/raytrace.zig:267 */
VR_HD float vr_trilerp(float x, float y, float z, float x0, float x1, float y0, float y1, float z0, float z1, const float values[8])
{
    const float xd = (x - x0) / (x1 - x0);
    const float yd = (y - y0) / (y1 - y0);
    const float zd = (z - z0) / (z1 - z0);
    const float c00 = values[0] * (1.0f - xd) + values[1] * xd;
    const float c10 = values[2] * (1.0f - xd) + values[3] * xd;
    const float c01 = values[4] * (1.0f - xd) + values[5] * xd;
    const float c11 = values[6] * (1.0f - xd) + values[7] * xd;
    const float c0 = c00 * (1.0f - yd) + c10 * yd;
    const float c1 = c01 * (1.0f - yd) + c11 * yd;
    return c0 * (1.0f - zd) + c1 * zd;
}

/* Data.getCornerDens (src/Data.zig:102-114) as floats, in trilerp's vertex order; vox is x slowest / z fastest (:90-92) */
VR_HD void vr_corner_dens(const uint8_t *vox, uint32_t ry, uint32_t rz, uint32_t cx, uint32_t cy, uint32_t cz, float out[8])
{
    const uint64_t sx = (uint64_t)ry * rz, sy = rz;
    const uint64_t b = cx * sx + cy * sy + cz;
    out[0] = (float)vox[b];
    out[1] = (float)vox[b + sx];
    out[2] = (float)vox[b + sy];
    out[3] = (float)vox[b + sx + sy];
    out[4] = (float)vox[b + 1];
    out[5] = (float)vox[b + sx + 1];
    out[6] = (float)vox[b + sy + 1];
    out[7] = (float)vox[b + sx + sy + 1];
}

#endif /* VR_TRILERP_H */
