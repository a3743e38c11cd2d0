// Microbenchmark: random 8-byte reads from an L2-resident table, as a function of how many lanes of
// each wave-instruction take part.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bench tools/micro/gather_bench.hip && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void gather_kernel(const uint2 *tab, uint32_t mask, int act, int iters, unsigned long long *out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if (static_cast<int>(lane) < act) {
        for (int i = 0; i < iters; i += 8) {
            uint2 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x = x * 1664525u + 1013904223u;
                r[k] = tab[(x >> 8) & mask];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += r[k].x ^ r[k].y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    for (uint32_t entries : {1u << 16, 1u << 20, 1u << 23}) {  // 512 KB, 8 MB, 64 MB tables
        uint2 *tab;
        unsigned long long *out;
        hipMalloc(&tab, entries * sizeof(uint2));
        hipMalloc(&out, 8);
        hipMemset(tab, 1, entries * sizeof(uint2));
        for (int act : {64, 32, 16, 10, 4, 1}) {
            const int iters = 512, blocks = 256 * 2, threads = 1024;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            gather_kernel<<<blocks, threads>>>(tab, entries - 1, act, iters, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            gather_kernel<<<blocks, threads>>>(tab, entries - 1, act, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double lanes = double(blocks) * threads / 64 * act * iters;
            const double insts = double(blocks) * threads / 64 * iters;
            printf("table %6u KB  active lanes %2d : %8.1f G lane-reads/s   %7.2f G wave-instr/s   (%.3f ms)\n", entries / 128, act,
                   lanes / ms / 1e6, insts / ms / 1e6, ms);
        }
        hipFree(tab); hipFree(out);
    }
    return 0;
}
