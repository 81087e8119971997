/*
 * vr_build.hip -- the centre-split SlimKDT built on the GPU from the uploaded u8 volume
 * (SlimKDT.newHeursiticPartition with AlwaysCenterH, src/SlimKDT.zig:222-322, :359-367).
 *
 * The reference recurses depth first and, per node, scans every corner of every cell of the node's
 * box (:275-281).  Here the same array falls out of two data-parallel passes:
 *   1. one thread per heap slot replays the slot's root-to-node path -- with centre splits the axis
 *      (largestDim, src/geo.zig:319-327) and partition ((min+max)/2, src/SlimKDT.zig:282-284) of every
 *      ancestor follow from the box alone -- and so learns whether the slot holds a node, its box, axis,
 *      partition and whether it is a leaf cell; leaves take their range from their 8 voxel corners
 *      (src/Data.zig:102-114);
 *   2. level by level from the bottom, every inner node takes the union of its children's ranges (the
 *      children's boxes share the voxel plane at the partition, so together they cover exactly the
 *      parent's voxels).
 * Both the reference-format array and the flattened array the render kernels walk are written.
 * HBM-bound integer work: 8 bytes written per slot, 8 voxel bytes read per leaf.
 *
 * WIDE = true builds the tree of a volume beyond the reference's u8 coordinates (up to 1024 voxels per
 * axis) in the wide node word of vr_trace_wide.h (12-bit partition, axis at bit 28); only that one array
 * is written.  No node word of either format is 0 (an inner node's partition is at least 1, a leaf's
 * axis is 2), so 0 marks an empty slot, and a node whose first child slot is empty is a leaf.
 */
#include "vr_internal.h"

namespace {

struct Box3 {
    uint32_t lo[3], hi[3];
    __device__ __forceinline__ uint32_t len(int a) const { return hi[a] - lo[a]; }
    __device__ __forceinline__ bool unit() const { return len(0) * len(1) * len(2) == 1u; }
    __device__ __forceinline__ int largest() const
    {
        if (len(0) > len(1) && len(0) > len(2)) return 0;
        if (len(1) > len(2)) return 1;
        return 2;
    }
};

template <bool WIDE>
__global__ __launch_bounds__(256) void vr_k_build_structure(const uint8_t *__restrict__ vox, uint32_t rx, uint32_t ry,
                                                           uint32_t rz, uint64_t n_slots, uint32_t *__restrict__ raw,
                                                           uint32_t *__restrict__ prepared)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_slots) return;
    const uint32_t g = (uint32_t)idx + 1u;
    const int level = 31 - __clz((int)g);
    Box3 b;
    b.lo[0] = b.lo[1] = b.lo[2] = 0;
    b.hi[0] = rx - 1u; b.hi[1] = ry - 1u; b.hi[2] = rz - 1u;
    bool exists = true;
    for (int s = level - 1; s >= 0; s--) {
        if (b.unit()) { exists = false; break; } /* an ancestor is a leaf cell: nothing below it */
        const int a = b.largest();
        const uint32_t part = (b.lo[a] + b.hi[a]) >> 1;
        if ((g >> s) & 1u) {
            if (a == 0) b.lo[0] = part; else if (a == 1) b.lo[1] = part; else b.lo[2] = part;
        } else {
            if (a == 0) b.hi[0] = part; else if (a == 1) b.hi[1] = part; else b.hi[2] = part;
        }
    }
    uint32_t w_raw = 0, w_prep = 0;
    if (exists) {
        const int a = b.largest();
        const uint32_t part = (b.lo[a] + b.hi[a]) >> 1;
        uint32_t mn = 0, mx = 0;
        const bool leaf = b.unit();
        if (leaf) {
            mn = 255u;
            const uint64_t sx = (uint64_t)ry * rz, sy = rz;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t v = vox[(b.lo[0] + (k & 1u)) * sx + (b.lo[1] + ((k >> 1) & 1u)) * sy + (b.lo[2] + (k >> 2))];
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
        }
        w_raw = mn | (mx << 8) | (part << 16) | ((uint32_t)a << (WIDE ? 28 : 24));
        w_prep = mn | (mx << 8) | (part << 16) | (leaf ? 0x80000000u : ((uint32_t)a * 10u) << 24);
    }
    raw[idx] = w_raw;
    if (!WIDE) {
        prepared[idx + 1] = w_prep;
        if (idx == 0) prepared[0] = 0;
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void vr_k_build_ranges(uint64_t first, uint64_t count, uint32_t *__restrict__ raw,
                                                        uint32_t *__restrict__ prepared)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t idx = first + i;
    uint32_t p = 0;
    if (WIDE) {
        if (raw[idx] == 0u || raw[2 * idx + 1] == 0u) return; /* empty slot or leaf */
    } else {
        p = prepared[idx + 1];
        if (p == 0u || (p & 0x80000000u)) return; /* empty slot or leaf */
    }
    const uint32_t c1 = raw[2 * idx + 1], c2 = raw[2 * idx + 2];
    const uint32_t mn1 = c1 & 0xffu, mn2 = c2 & 0xffu, mx1 = (c1 >> 8) & 0xffu, mx2 = (c2 >> 8) & 0xffu;
    const uint32_t r = (mn1 < mn2 ? mn1 : mn2) | ((mx1 > mx2 ? mx1 : mx2) << 8);
    raw[idx] = (raw[idx] & 0xffff0000u) | r;
    if (!WIDE) prepared[idx + 1] = (p & 0xffff0000u) | r;
}

/* ------------------------------------------------------------------------------------------
 * The span heuristics (src/SlimKDT.zig:327-598) on the GPU: the tree is built top-down, one level per
 * launch, because a node's partition -- and with it the boxes of everything below -- depends on the
 * data.  One wave per node of the level:
 *   * a leaf cell takes its range from its 8 voxel corners (src/Data.zig:102-114);
 *   * otherwise the wave scans the voxels of the node's box once into per-plane ("slab") density
 *     ranges along the split axis (LDS atomics), turns them into prefix / suffix ranges, scores every
 *     candidate plane min+1 .. max-1 with the reference's expression in f32 (:327-357: deviation from
 *     the centre times the weight, plus the spans of the two halves) and keeps the first strictly
 *     smallest (:289-298); layers where AlwaysCenterH decides (:359-406) skip the scan;
 *   * the two children go to the next level's list (heap index, box).
 * Afterwards the nodes are scattered into the heap array and the ranges of inner nodes are filled
 * bottom-up from their children, as in the centre-split build above.
 * ---------------------------------------------------------------------------------------- */
#define VR_SPAN_LEAF 0x80000000u /* list word of a leaf (stripped when the word goes to the heap array) */

struct SpanCtl {
    uint32_t next_count;               /* nodes appended to the next level's list */
    uint32_t error;                    /* a child's heap index would pass the slot limit */
    unsigned long long non_center;     /* inner nodes whose partition is not the centre plane */
};

__device__ __forceinline__ bool span_active(int heuristic, uint32_t layer, float *weight)
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

/* boxes travel as two words: lo = x | y<<8 | z<<16, hi likewise (cell coordinates, hi exclusive).
 * TPN threads work on one node: a wave (4 nodes per 256-thread block) on the lower levels, where nodes are many and
 * small, a whole 1024-thread block on the top levels, where a node's box holds up to the whole volume. */
constexpr uint32_t kSpanWaveNodes = 8; /* nodes per block where a wave works on a node */

/* One node.  Writes its list word; returns true (in the node's thread 0) when it has children, with the heap index of
 * the first and their boxes. */
template <uint32_t TPN>
__device__ __forceinline__ bool span_node(const uint8_t *__restrict__ vox, uint32_t ry, uint32_t rz, int heuristic, uint32_t layer,
                                          uint64_t max_slots, uint32_t n, uint32_t wave, uint32_t lane,
                                          const uint32_t *__restrict__ in_idx, const uint2 *__restrict__ in_box,
                                          uint32_t *__restrict__ word_out, SpanCtl *__restrict__ ctl, uint32_t &child, uint2 &lbox,
                                          uint2 &rbox)
{
    constexpr uint32_t kNodesPerBlock = TPN == 64u ? kSpanWaveNodes : 1u;
    __shared__ uint32_t s_mn[kNodesPerBlock][256], s_mx[kNodesPerBlock][256]; /* per node: slab ranges, then prefix ... */
    __shared__ uint32_t s_smn[kNodesPerBlock][256], s_smx[kNodesPerBlock][256]; /* ... and suffix ranges */
    /* the threads of a node meet here: one wave needs no barrier, a block that works on one node takes the same
     * path in all its threads */
    auto meet = [&]() {
        if (TPN == 64u) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        else __syncthreads();
    };
    const uint32_t idx = in_idx[n];
    const uint2 box = in_box[n];
    uint32_t lo[3] = { box.x & 0xffu, (box.x >> 8) & 0xffu, (box.x >> 16) & 0xffu };
    uint32_t hi[3] = { box.y & 0xffu, (box.y >> 8) & 0xffu, (box.y >> 16) & 0xffu };
    const uint32_t l0 = hi[0] - lo[0], l1 = hi[1] - lo[1], l2 = hi[2] - lo[2];
    const int axis = (l0 > l1 && l0 > l2) ? 0 : (l1 > l2 ? 1 : 2); /* src/geo.zig:319-327 */
    const uint32_t alo = axis == 0 ? lo[0] : (axis == 1 ? lo[1] : lo[2]), ahi = axis == 0 ? hi[0] : (axis == 1 ? hi[1] : hi[2]);
    const uint32_t mid = (alo + ahi) >> 1; /* src/SlimKDT.zig:282-284 */
    const uint64_t sx = (uint64_t)ry * rz, sy = rz;

    if (l0 * l1 * l2 == 1u) { /* leaf cell: src/SlimKDT.zig:310-312 */
        uint32_t v = 0;
        if (lane < 8u) v = vox[(lo[0] + (lane & 1u)) * sx + (lo[1] + ((lane >> 1) & 1u)) * sy + (lo[2] + (lane >> 2))];
        uint32_t mn = lane < 8u ? v : 255u, mx = v;
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            const uint32_t a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (lane == 0) word_out[n] = mn | (mx << 8) | (mid << 16) | ((uint32_t)axis << 24) | VR_SPAN_LEAF;
        return false;
    }

    uint32_t part = mid;
    float weight;
    if (span_active(heuristic, layer, &weight)) {
        /* slab ranges over the voxels lo .. hi (inclusive) of the box */
        for (uint32_t s = alo + lane; s <= ahi; s += TPN) { s_mn[wave][s] = 255u; s_mx[wave][s] = 0u; }
        meet();
        const uint32_t ny = l1 + 1u, nz = l2 + 1u, total = (l0 + 1u) * ny * nz;
        for (uint32_t i0 = lane; i0 < total; i0 += 4u * TPN) { /* four voxels per thread and trip: four loads in flight */
            uint32_t v[4], sl[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t i = i0 + k * TPN;
                const uint32_t z = i % nz, t = i / nz, y = t % ny, x = t / ny;
                sl[k] = axis == 0 ? lo[0] + x : (axis == 1 ? lo[1] + y : lo[2] + z);
                v[k] = i < total ? vox[(lo[0] + x) * sx + (lo[1] + y) * sy + (lo[2] + z)] : 0xffffffffu;
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++)
                if (v[k] != 0xffffffffu) {
                    atomicMin(&s_mn[wave][sl[k]], v[k]);
                    atomicMax(&s_mx[wave][sl[k]], v[k]);
                }
        }
        meet();
        if (lane == 0) { /* prefix and suffix ranges, in place / into the second pair */
            uint32_t mn = 255u, mx = 0u;
            for (uint32_t s = ahi + 1u; s-- > alo;) {
                const uint32_t a = s_mn[wave][s], b = s_mx[wave][s];
                mn = a < mn ? a : mn; mx = b > mx ? b : mx;
                s_smn[wave][s] = mn; s_smx[wave][s] = mx;
            }
            mn = 255u; mx = 0u;
            for (uint32_t s = alo; s <= ahi; s++) {
                const uint32_t a = s_mn[wave][s], b = s_mx[wave][s];
                mn = a < mn ? a : mn; mx = b > mx ? b : mx;
                s_mn[wave][s] = mn; s_mx[wave][s] = mx;
            }
        }
        meet();
        /* candidates alo+1 .. ahi-1 (first wave of the node): the left half holds voxels alo..p, the right half p..ahi */
        float best_h = INFINITY;
        uint32_t best_p = mid;
        if (lane < 64u)
        for (uint32_t p = alo + 1u + lane; p < ahi; p += 64u) {
            float h = 0.0f;
            h += fabsf((float)p - (float)mid) * weight;
            h += fabsf((float)s_mx[wave][p] - (float)s_mn[wave][p]);
            h += fabsf((float)s_smx[wave][p] - (float)s_smn[wave][p]);
            if (best_h > h) { best_h = h; best_p = p; }
        }
        /* first strictly smallest over the wave: smallest score, then smallest plane among equal scores */
        float m = best_h;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
        uint32_t cand = best_h == m ? best_p : 0xffffffffu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(cand, off, 64); cand = o < cand ? o : cand; }
        if (cand != 0xffffffffu) part = cand; /* no candidate at all cannot happen: the axis is at least 2 cells long */
    }

    if (lane != 0) return false;
    word_out[n] = (part << 16) | ((uint32_t)axis << 24);
    if (part != mid) atomicAdd(&ctl->non_center, 1ull);
    const uint64_t c1 = 2ull * idx + 1ull;
    if (c1 + 1ull >= max_slots) { /* the reference's realloc fails here: src/SlimKDT.zig:266-273 */
        atomicExch(&ctl->error, 1u);
        return false;
    }
    uint32_t lhi[3] = { hi[0], hi[1], hi[2] }, rlo[3] = { lo[0], lo[1], lo[2] };
    if (axis == 0) { lhi[0] = part; rlo[0] = part; } else if (axis == 1) { lhi[1] = part; rlo[1] = part; } else { lhi[2] = part; rlo[2] = part; }
    child = (uint32_t)c1;
    lbox = make_uint2(box.x, lhi[0] | (lhi[1] << 8) | (lhi[2] << 16));
    rbox = make_uint2(rlo[0] | (rlo[1] << 8) | (rlo[2] << 16), box.y);
    return true;
}

template <uint32_t TPN>
__global__ __launch_bounds__(TPN == 64u ? 64u * kSpanWaveNodes : TPN) void vr_k_span_level(
    const uint8_t *__restrict__ vox, uint32_t ry, uint32_t rz, int heuristic, uint32_t layer, uint32_t n_nodes, uint64_t max_slots,
    const uint32_t *__restrict__ in_idx, const uint2 *__restrict__ in_box, uint32_t *__restrict__ word_out,
    uint32_t *__restrict__ next_idx, uint2 *__restrict__ next_box, SpanCtl *__restrict__ ctl)
{
    constexpr uint32_t kNodesPerBlock = TPN == 64u ? kSpanWaveNodes : 1u;
    const uint32_t wave = threadIdx.x / TPN, lane = threadIdx.x % TPN; /* node slot in the block, thread of the node */
    const uint32_t n = blockIdx.x * kNodesPerBlock + wave;
    uint32_t child = 0;
    uint2 lbox = make_uint2(0, 0), rbox = make_uint2(0, 0);
    bool emit = false;
    if (n < n_nodes) /* a group of TPN threads takes this branch together */
        emit = span_node<TPN>(vox, ry, rz, heuristic, layer, max_slots, n, wave, lane, in_idx, in_box, word_out, ctl, child, lbox, rbox);
    /* places in the next level's list: one atomic per block (one per node would make the counter the bottleneck of
     * the lower levels, which have millions of nodes) */
    __shared__ uint32_t s_emit[kNodesPerBlock], s_base;
    if (lane == 0) s_emit[wave] = emit ? 2u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t k = 0; k < kNodesPerBlock; k++) { const uint32_t e = s_emit[k]; s_emit[k] = total; total += e; }
        s_base = total ? atomicAdd(&ctl->next_count, total) : 0u;
    }
    __syncthreads();
    if (emit) { /* thread 0 of the node */
        const uint32_t at = s_base + s_emit[wave];
        next_idx[at] = child;
        next_box[at] = lbox;
        next_idx[at + 1u] = child + 1u;
        next_box[at + 1u] = rbox;
    }
}

/* list -> heap array */
__global__ void vr_k_span_scatter(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ word, uint32_t n, uint32_t *__restrict__ raw)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) raw[idx[i]] = word[i] & ~VR_SPAN_LEAF;
}

/* one level, bottom-up: inner nodes take the union of their children's ranges; both arrays get their final words */
__global__ void vr_k_span_ranges(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ word, uint32_t n, uint32_t *__restrict__ raw,
                                 uint32_t *__restrict__ prepared)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t at = idx[i];
    uint32_t w = word[i];
    if (w & VR_SPAN_LEAF) {
        prepared[at + 1] = (w & 0x00ffffffu) | 0x80000000u;
        return;
    }
    const uint32_t c1 = raw[2 * at + 1], c2 = raw[2 * at + 2];
    const uint32_t mn1 = c1 & 0xffu, mn2 = c2 & 0xffu, mx1 = (c1 >> 8) & 0xffu, mx2 = (c2 >> 8) & 0xffu;
    w |= (mn1 < mn2 ? mn1 : mn2) | ((mx1 > mx2 ? mx1 : mx2) << 8);
    raw[at] = w;
    prepared[at + 1] = (w & 0x00ffffffu) | ((((w >> 24) & 3u) * 10u) << 24);
}

uint32_t ceil_log2(uint32_t v)
{
    uint32_t l = 0;
    while ((1u << l) < v) l++;
    return l;
}

} // namespace

uint32_t vr_center_tree_depth(uint32_t rx, uint32_t ry, uint32_t rz)
{
    /* every split halves one axis (floor / ceil); the deepest leaf takes the larger half every time */
    return ceil_log2(rx - 1u) + ceil_log2(ry - 1u) + ceil_log2(rz - 1u);
}

hipError_t vr_launch_build_center(hipStream_t stream, const uint8_t *d_vox, uint32_t rx, uint32_t ry, uint32_t rz,
                                  uint32_t depth, uint32_t *d_raw, uint32_t *d_prepared)
{
    const uint64_t n_slots = (uint64_t)1 << (depth + 1);
    if (!d_prepared) { /* wide node words, one array */
        hipLaunchKernelGGL(vr_k_build_structure<true>, dim3((uint32_t)((n_slots + 255u) / 256u)), dim3(256), 0, stream, d_vox,
                           rx, ry, rz, n_slots, d_raw, (uint32_t *)nullptr);
        for (int level = (int)depth - 1; level >= 0; level--) {
            const uint64_t first = ((uint64_t)1 << level) - 1u, count = (uint64_t)1 << level;
            hipLaunchKernelGGL(vr_k_build_ranges<true>, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, first,
                               count, d_raw, (uint32_t *)nullptr);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(vr_k_build_structure<false>, dim3((uint32_t)((n_slots + 255u) / 256u)), dim3(256), 0, stream, d_vox, rx, ry,
                       rz, n_slots, d_raw, d_prepared);
    for (int level = (int)depth - 1; level >= 0; level--) {
        const uint64_t first = ((uint64_t)1 << level) - 1u, count = (uint64_t)1 << level;
        hipLaunchKernelGGL(vr_k_build_ranges<false>, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, first,
                           count, d_raw, d_prepared);
    }
    return hipGetLastError();
}

/* SlimKDT.newHeursiticPartition with a span heuristic, on the device.  On success *raw_out (2^(depth+1) words, the
 * reference's layout) and *prepared_raw_out / *prepared_out (the flattened form, placed by `place`) are device
 * allocations the caller owns.  Returns hipSuccess with *too_deep = true where the reference's build fails
 * (a heap index beyond max_slots). */
hipError_t vr_build_span_tree(hipStream_t stream, const uint8_t *d_vox, uint32_t rx, uint32_t ry, uint32_t rz, int heuristic,
                              uint64_t max_slots, hipError_t (*place)(size_t, void **, uint32_t **), uint32_t **raw_out,
                              void **prepared_raw_out, uint32_t **prepared_out, uint32_t *depth_out, vr_build_stats *stats_out,
                              bool *too_deep)
{
    *too_deep = false;
    *raw_out = nullptr; *prepared_raw_out = nullptr; *prepared_out = nullptr;
    const uint64_t cells = (uint64_t)(rx - 1u) * (ry - 1u) * (rz - 1u);
    const uint64_t cap_all = 2 * cells; /* a binary tree over the cells has 2 * cells - 1 nodes */
    uint32_t *idx_all = nullptr, *word_all = nullptr;
    uint2 *box[2] = { nullptr, nullptr };
    SpanCtl *ctl = nullptr;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        (void)hipFree(idx_all); (void)hipFree(word_all); (void)hipFree(box[0]); (void)hipFree(box[1]); (void)hipFree(ctl);
    };
    if ((e = hipMalloc((void **)&idx_all, cap_all * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&word_all, cap_all * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&box[0], cells * sizeof(uint2))) != hipSuccess ||
        (e = hipMalloc((void **)&box[1], cells * sizeof(uint2))) != hipSuccess ||
        (e = hipMalloc((void **)&ctl, sizeof(SpanCtl))) != hipSuccess) { cleanup(); return e; }
    const uint32_t root_idx = 0;
    const uint2 root_box = make_uint2(0u, (rx - 1u) | ((ry - 1u) << 8) | ((rz - 1u) << 16));
    if ((e = hipMemcpyAsync(idx_all, &root_idx, sizeof root_idx, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(box[0], &root_box, sizeof root_box, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemsetAsync(ctl, 0, sizeof(SpanCtl), stream)) != hipSuccess) { cleanup(); return e; }
    uint64_t level_off[40];
    uint32_t level_n[40];
    uint32_t n = 1, layer = 0;
    uint64_t off = 0;
    SpanCtl h;
    for (;;) {
        if (off + n > cap_all || layer >= 39u) { cleanup(); return hipErrorInvalidValue; } /* cannot happen: more nodes than a tree over the cells has */
        level_off[layer] = off; level_n[layer] = n;
        if (n <= 2048u) /* few nodes with large boxes: a block per node */
            hipLaunchKernelGGL(vr_k_span_level<1024u>, dim3(n), dim3(1024), 0, stream, d_vox, ry, rz, heuristic, layer, n, max_slots,
                               idx_all + off, box[layer & 1u], word_all + off, idx_all + off + n, box[(layer + 1u) & 1u], ctl);
        else
            hipLaunchKernelGGL(vr_k_span_level<64u>, dim3((n + kSpanWaveNodes - 1u) / kSpanWaveNodes), dim3(64u * kSpanWaveNodes), 0, stream, d_vox, ry, rz, heuristic, layer, n,
                               max_slots, idx_all + off, box[layer & 1u], word_all + off, idx_all + off + n, box[(layer + 1u) & 1u], ctl);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpyAsync(&h, ctl, sizeof h, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
            (e = hipStreamSynchronize(stream)) != hipSuccess) { cleanup(); return e; }
        if (h.error) { cleanup(); *too_deep = true; return hipSuccess; }
        off += n;
        if (h.next_count == 0u) break;
        n = h.next_count;
        layer++;
        if (layer > 31u) { cleanup(); *too_deep = true; return hipSuccess; } /* heap indices are u32: src/SlimKDT.zig:24-34 */
        if ((e = hipMemsetAsync(&ctl->next_count, 0, sizeof(uint32_t), stream)) != hipSuccess) { cleanup(); return e; }
    }
    const uint32_t depth = layer;
    const uint64_t n_slots = (uint64_t)1 << (depth + 1);
    uint32_t *raw = nullptr, *prepared = nullptr;
    void *prepared_raw = nullptr;
    if ((e = hipMalloc((void **)&raw, n_slots * sizeof(uint32_t))) != hipSuccess ||
        (e = place((n_slots + 1) * sizeof(uint32_t), &prepared_raw, &prepared)) != hipSuccess ||
        (e = hipMemsetAsync(raw, 0, n_slots * sizeof(uint32_t), stream)) != hipSuccess ||
        (e = hipMemsetAsync(prepared, 0, (n_slots + 1) * sizeof(uint32_t), stream)) != hipSuccess) {
        (void)hipFree(raw); (void)hipFree(prepared_raw); cleanup(); return e;
    }
    const uint32_t total = (uint32_t)off;
    hipLaunchKernelGGL(vr_k_span_scatter, dim3((total + 255u) / 256u), dim3(256), 0, stream, idx_all, word_all, total, raw);
    for (int l = (int)depth; l >= 0; l--)
        hipLaunchKernelGGL(vr_k_span_ranges, dim3((level_n[l] + 255u) / 256u), dim3(256), 0, stream, idx_all + level_off[l],
                           word_all + level_off[l], level_n[l], raw, prepared);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(stream)) != hipSuccess) {
        (void)hipFree(raw); (void)hipFree(prepared_raw); cleanup(); return e;
    }
    cleanup();
    *raw_out = raw; *prepared_raw_out = prepared_raw; *prepared_out = prepared; *depth_out = depth;
    if (stats_out) {
        stats_out->non_center_partition_count = h.non_center;
        stats_out->layer_count = depth;
        stats_out->number_of_nodes = off;
    }
    return hipSuccess;
}
