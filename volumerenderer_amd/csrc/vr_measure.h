/*
 * vr_measure.h -- everything the measuring build (-DVR_MEASURE) adds to the union kernel, in one place.  The product library
 * is built without VR_MEASURE: every macro below is then empty and nothing of this file reaches the code object.
 *
 *   tools/tune_build.sh items -DVR_MEASURE -DVR_SWEEP_MAX_BLOCKS=256     ->  volumerenderer_amd/tune/libvr_items.so
 *   (add -DVR_MEASURE_COUNTS for the wave-level execution counts, in a build of their own: libvr_counts.so)
 *
 * What it records, for vr_k_union<UnionNarrow / UnionWide> launches through vr_launch_render:
 *   - per (ray, block of iso values) item: rounds and ticks of the 100 MHz clock from its start to the end of its last front,
 *     and when it started (a run of iso values handed to another lane stays a part of its item);
 *   - for ONE item named by the host (VR_MEASURE_TRACE=ray,block,blocks per ray): every round, with the number of lanes of its
 *     wave that hold a front / are descending;
 *   - how often a wave runs through each piece of the lane code (VR_WC(i) sites in vr_trace_union.h and the kernel).
 * VR_MEASURE_DUMP=<file> makes the launcher write the item table after the launch ([rays, blocks per ray, iso values per block,
 * first iso value][rounds][ticks][begin][pixel of every ray], u32 each) and print the wave counts to stderr;
 * VR_MEASURE_TRACE_OUT=<file> receives the traced item's rounds.  tools/item_lengths.py and tools/item_trace.py drive it.
 */
#ifndef VR_MEASURE_H
#define VR_MEASURE_H

#ifdef VR_MEASURE

#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr uint32_t kMeasureItemsCap = 80u << 20;
__device__ uint32_t g_item_rounds[kMeasureItemsCap]; /* [ray * n_blocks + block] */
__device__ uint32_t g_item_ticks[kMeasureItemsCap];
__device__ uint32_t g_item_begin[kMeasureItemsCap];  /* ticks from the launch's first wave, + 1 */
__device__ uint32_t g_item_dims[4];                  /* rays in the list, blocks per ray, iso values per block, first iso value */
__device__ uint32_t g_trace_slot = 0xffffffffu;      /* the item whose rounds are traced one by one */
__device__ uint32_t g_trace_n;
__device__ uint32_t g_trace[2 * 4096]; /* per round: tick since the launch began, fronts in the wave | descending << 8 | popping << 16 | round << 20 */
__device__ unsigned long long g_wave_counts[24];
__device__ uint32_t g_trace_wave = 0xffffffffu;      /* block * 16 + wave of the wave that took the traced item */

/* one count per wave that runs through here, whatever the number of its lanes that do.  Only with -DVR_MEASURE_COUNTS as well: the
 * counts are device-wide atomics on two dozen words and make a launch a hundred times longer, so the item tables and the round
 * trace of a build that has them say nothing about time. */
#if defined(VR_MEASURE_COUNTS) && VR_MEASURE_COUNTS == 2
/* ... = 2: only the wave that holds the traced item counts (from the moment it takes the item): what ONE wave runs through */
#define VR_WC(i) do { if (blockIdx.x * 16u + (threadIdx.x >> 6) == *(volatile uint32_t *)&g_trace_wave) { const unsigned long long e__ = __ballot(1); if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)e__) - 1u) atomicAdd(&g_wave_counts[i], 1ull); } } while (0)
#elif defined(VR_MEASURE_COUNTS)
#define VR_WC(i) do { const unsigned long long e__ = __ballot(1); if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)e__) - 1u) atomicAdd(&g_wave_counts[i], 1ull); } while (0)
#else
#define VR_WC(i) do { } while (0)
#endif

/* per-lane state of the item a lane works on */
struct VrMeasureLane {
    uint32_t slot, t0, rounds;
    __device__ __forceinline__ void begin(uint32_t item_slot, unsigned long long tick_begin)
    {
        slot = item_slot; t0 = (uint32_t)__builtin_amdgcn_s_memrealtime(); rounds = 0;
        if (slot < kMeasureItemsCap) g_item_begin[slot] = t0 - (uint32_t)tick_begin + 1u;
        if (slot == g_trace_slot) g_trace_wave = blockIdx.x * 16u + (threadIdx.x >> 6);
    }
    __device__ __forceinline__ void end() const
    {
        if (slot >= kMeasureItemsCap) return;
        atomicMax(&g_item_rounds[slot], rounds);
        atomicMax(&g_item_ticks[slot], (uint32_t)__builtin_amdgcn_s_memrealtime() - t0);
    }
    /* a round of the wave begins: `alive` lanes hold a front, `popping` ones are on their way up */
    __device__ __forceinline__ void round(bool alive, bool popping, unsigned long long tick_begin)
    {
        if (alive) rounds++;
        const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive)), n_desc = (uint32_t)__popcll(__ballot(alive && !popping));
        if (alive && slot == g_trace_slot) {
            const uint32_t at = atomicAdd(&g_trace_n, 1u);
            if (at < 4096u) {
                g_trace[2u * at] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tick_begin);
                g_trace[2u * at + 1u] = n_alive | (n_desc << 8) | ((uint32_t)popping << 16) | (rounds << 20);
            }
        }
    }
    /* a run of iso values goes from lane `giver` to lane `taker`: it stays a part of the giver's item */
    __device__ __forceinline__ void hand_over(uint32_t giver, bool i_take)
    {
        const uint32_t g_slot = (uint32_t)__builtin_amdgcn_readlane((int)slot, giver), g_t0 = (uint32_t)__builtin_amdgcn_readlane((int)t0, giver);
        const uint32_t g_rounds = (uint32_t)__builtin_amdgcn_readlane((int)rounds, giver);
        if (i_take) { slot = g_slot; t0 = g_t0; rounds = g_rounds; }
    }
};
#define VR_MEASURE_LANE(name) VrMeasureLane name = { 0u, 0u, 0u }
#define VR_MEASURE_DIMS(n_active, n_blocks, block_isos, iso_begin) \
    do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_item_dims[0] = (n_active); g_item_dims[1] = (n_blocks); g_item_dims[2] = (block_isos); g_item_dims[3] = (iso_begin); } } while (0)
#define VR_MEASURE_DO(stmt) do { stmt; } while (0)

/* host side: before a union launch (tables cleared, the traced item named) ... */
static inline void vr_measure_before_launch(hipStream_t stream)
{
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    (void)hipGetSymbolAddress(&p0, HIP_SYMBOL(g_item_rounds));
    (void)hipGetSymbolAddress(&p1, HIP_SYMBOL(g_item_ticks));
    (void)hipGetSymbolAddress(&p2, HIP_SYMBOL(g_item_begin));
    (void)hipMemsetAsync(p0, 0, sizeof(uint32_t) * kMeasureItemsCap, stream);
    (void)hipMemsetAsync(p1, 0, sizeof(uint32_t) * kMeasureItemsCap, stream);
    (void)hipMemsetAsync(p2, 0, sizeof(uint32_t) * kMeasureItemsCap, stream);
    unsigned ray = 0, blk = 0, nb = 1;
    if (getenv("VR_MEASURE_TRACE") && sscanf(getenv("VR_MEASURE_TRACE"), "%u,%u,%u", &ray, &blk, &nb) == 3) {
        const uint32_t slot = ray * nb + blk, zero = 0;
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_slot), &slot, 4);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, 4);
        const uint32_t none = 0xffffffffu;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_wave), &none, 4);
    }
}

/* ... and after it: the dumps the environment asks for */
static inline void vr_measure_after_launch(hipStream_t stream, const uint32_t *d_ray_pix)
{
    if (!getenv("VR_MEASURE_DUMP")) return;
    (void)hipStreamSynchronize(stream);
    uint32_t dims[4] = { 0, 0, 0, 0 };
    (void)hipMemcpyFromSymbol(dims, HIP_SYMBOL(g_item_dims), sizeof dims);
    const size_t n = (size_t)dims[0] * dims[1] < kMeasureItemsCap ? (size_t)dims[0] * dims[1] : kMeasureItemsCap;
    std::vector<uint32_t> buf(n);
    if (FILE *f = fopen(getenv("VR_MEASURE_DUMP"), "wb")) {
        fwrite(dims, sizeof dims, 1, f);
        (void)hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(g_item_rounds), n * sizeof(uint32_t)); fwrite(buf.data(), sizeof(uint32_t), n, f);
        (void)hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(g_item_ticks), n * sizeof(uint32_t)); fwrite(buf.data(), sizeof(uint32_t), n, f);
        (void)hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(g_item_begin), n * sizeof(uint32_t)); fwrite(buf.data(), sizeof(uint32_t), n, f);
        buf.resize(dims[0]);
        (void)hipMemcpy(buf.data(), d_ray_pix, (size_t)dims[0] * sizeof(uint32_t), hipMemcpyDeviceToHost); fwrite(buf.data(), sizeof(uint32_t), dims[0], f);
        fclose(f);
    }
    unsigned long long c[24], z[24] = { 0 };
    (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_wave_counts), sizeof c);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_counts), z, sizeof z);
    static const char *what[20] = { "descend trips", "low-side parks", "high-side parks", "iterations of the run loop (done bits)",
        "done words read by the does-anybody-hold test", "leaves", "iterations of the pixel-store loop", "MOVEs", "iterations of the escape loop",
        "levels left on the way out of the volume", "levels of exact resolutions", "pop trips", "fronts restarted from kept runs", "refills",
        "hand-overs", "rounds", "hit runs", "items begun", "pop phases", "ends of a descent" };
    fprintf(stderr, "[wave counts: how often a wave ran through a piece of the lane code, all launches since the last dump]\n");
    for (int i = 0; i < 20; i++) fprintf(stderr, "  %-50s %12llu  (%.2f per round)\n", what[i], c[i], c[15] ? (double)c[i] / (double)c[15] : 0.0);
    if (getenv("VR_MEASURE_TRACE") && getenv("VR_MEASURE_TRACE_OUT")) {
        uint32_t n_tr = 0;
        static uint32_t tr[2 * 4096];
        (void)hipMemcpyFromSymbol(&n_tr, HIP_SYMBOL(g_trace_n), 4);
        (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof tr);
        if (FILE *f = fopen(getenv("VR_MEASURE_TRACE_OUT"), "w")) {
            for (uint32_t i = 0; i < n_tr && i < 4096u; i++)
                fprintf(f, "%u %.2f %u %u %u\n", tr[2 * i + 1] >> 20, tr[2 * i] / 100.0, tr[2 * i + 1] & 255u, (tr[2 * i + 1] >> 8) & 255u, (tr[2 * i + 1] >> 16) & 1u);
            fclose(f);
        }
    }
}

#else /* product build: nothing */

#define VR_WC(i) do { } while (0)
#define VR_MEASURE_LANE(name) do { } while (0)
#define VR_MEASURE_DIMS(n_active, n_blocks, block_isos, iso_begin) do { } while (0)
#define VR_MEASURE_DO(stmt) do { } while (0)

#endif /* VR_MEASURE */
#endif /* VR_MEASURE_H */
