// Micro-benchmark: cost of a dependent 64-lane gather from an L2-resident table on gfx950.
// Each lane chases idx = table[idx] (precomputed permutation) so every load depends on the previous
// one, like a tree descent.  Variables: lanes active per wave, distinct lines per wave-load, load
// width, waves per CU.  Prints cycles per wave-load instruction per CU (throughput view).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

template <int WIDTH>
__global__ void chase(const uint32_t *__restrict__ tab, uint32_t mask_words, uint32_t iters, uint32_t active_lanes,
                      uint32_t spread, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // spread = how far apart the lanes of a wave start (in words): 1 = same line region, large = all different lines
    uint32_t idx = (gw * 977u + lane * spread) & mask_words;
    uint32_t acc = 0;
    if (lane < active_lanes) {
        for (uint32_t i = 0; i < iters; i++) {
            if (WIDTH == 1) {
                idx = tab[idx];
            } else if (WIDTH == 2) {
                uint2 v = *reinterpret_cast<const uint2 *>(tab + (idx & ~1u));
                idx = v.x; acc += v.y;
            } else {
                uint4 v = *reinterpret_cast<const uint4 *>(tab + (idx & ~3u));
                idx = v.x; acc += v.y + v.z + v.w;
            }
        }
    }
    if (idx == 0xffffffffu) out[0] = acc + idx;
}

int main(int argc, char **argv)
{
    const uint32_t words = 1u << 22; // 16 MiB table: L2/MALL resident, beyond L1
    std::vector<uint32_t> h(words);
    std::mt19937 rng(1);
    uint32_t *d, *out;
    hipMalloc(&d, words * 4); hipMalloc(&out, 64);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // mode 0: near = next index stays within +-64 words of the current one (good locality, like siblings)
    // mode 1: random = next index anywhere in the table
    for (int mode = 0; mode < 2; mode++) {
        for (uint32_t i = 0; i < words; i++) {
            if (mode == 0) h[i] = ((i & ~63u) + (rng() & 63u)) & (words - 1);
            else h[i] = (rng() & (words - 1)) & ~3u;
        }
        if (mode == 0) for (uint32_t i = 0; i < words; i++) h[i] &= ~3u;
        hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
        for (int width : {1, 4}) for (uint32_t lanes : {64u, 16u, 1u}) for (uint32_t spread : {4u, 4096u}) for (int wpc : {4, 8, 16, 32}) {
            const uint32_t iters = 2000;
            dim3 grid(cus * wpc / 4), block(256);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (width == 1) hipLaunchKernelGGL(chase<1>, grid, block, 0, 0, d, words - 1, iters, lanes, spread, out);
                else hipLaunchKernelGGL(chase<4>, grid, block, 0, 0, d, words - 1, iters, lanes, spread, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * 2.4e9;
            printf("mode %s width %d lanes %2u spread %4u waves/CU %2d : %.1f cyc per dependent load (latency view), %.1f cyc per wave-load per CU (throughput view)\n",
                   mode ? "random" : "near  ", width, lanes, spread, wpc, cyc / iters, cyc / iters / wpc);
        }
    }
    return 0;
}
