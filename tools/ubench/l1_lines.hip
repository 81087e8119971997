// Micro-benchmark: cost of an L1-resident 64-lane gather as a function of the number of distinct
// 128-byte lines the wave touches (dependent chain per lane, table small enough to stay in L1),
// and the same gather from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

__global__ void chase_l1(const uint32_t *__restrict__ tab, uint32_t iters, uint32_t lines, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    // lane l starts in line (l % lines); the table maps every word to a word of the SAME line
    uint32_t idx = (lane % lines) * 32u + (lane & 31u);
    for (uint32_t i = 0; i < iters; i++) idx = tab[idx];
    if (idx == 0xffffffffu) out[0] = idx;
}

__global__ void chase_lds(const uint32_t *__restrict__ tab, uint32_t words, uint32_t iters, uint32_t lines, uint32_t *out)
{
    extern __shared__ uint32_t s[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) s[i] = tab[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (lane % lines) * 32u + (lane & 31u);
    for (uint32_t i = 0; i < iters; i++) idx = s[idx];
    if (idx == 0xffffffffu) out[0] = idx;
}

int main()
{
    const uint32_t words = 64 * 32; // 64 lines of 128 B = 8 KiB
    std::vector<uint32_t> h(words);
    std::mt19937 rng(1);
    for (uint32_t i = 0; i < words; i++) h[i] = (i & ~31u) + (rng() & 31u);
    uint32_t *d, *out;
    (void)hipMalloc(&d, words * 4); (void)hipMalloc(&out, 64);
    (void)hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t iters = 4000;
    for (int lds = 0; lds < 2; lds++)
        for (uint32_t lines : {1u, 2u, 4u, 8u, 16u, 32u, 64u})
            for (int wpc : {4, 16, 32}) {
                dim3 grid(cus * wpc / 4), block(256);
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    (void)hipEventRecord(e0);
                    if (lds) hipLaunchKernelGGL(chase_lds, grid, block, words * 4, 0, d, words, iters, lines, out);
                    else hipLaunchKernelGGL(chase_l1, grid, block, 0, 0, d, iters, lines, out);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                const double cyc = ms * 1e-3 * 2.4e9;
                printf("%s lines/wave-load %2u waves/CU %2d : latency %.0f cyc, %.1f cyc per wave-load per CU\n",
                       lds ? "LDS" : "L1 ", lines, wpc, cyc / iters, cyc / iters / wpc);
            }
    return 0;
}
