// How should 24-byte tuples that land 100-300 bytes apart leave a wave?  (a) one lane per tuple: a 16-byte and an 8-byte
// store (two instructions, every lane in a different line); (b) two adjacent lanes per tuple: 12 bytes each in ONE
// global_store_dwordx3 (the address coalescer sees both halves of a tuple in the same line of the same instruction).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pairs tools/micro/store_pairs.hip && /tmp/store_pairs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// every wave owns a contiguous stretch of the output; lane l's tuples start `gap` tuples after lane l-1's (as in the emitter:
// a lane's positions are consecutive, so its tuples are, and the lanes' runs follow each other)
template <int MODE>
__global__ __launch_bounds__(1024) void k(char *out, uint32_t tuples_per_lane, uint32_t gap) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t wave_base = static_cast<uint64_t>(wave) * 64 * gap * 24;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    for (uint32_t t = 0; t < tuples_per_lane; ++t) {
        if (MODE == 0) {
            char *p = out + wave_base + (static_cast<uint64_t>(lane) * gap + t) * 24;
            *reinterpret_cast<u32x4 *>(p) = u32x4{t, lane, t + 1, 0};
            *reinterpret_cast<u32x2 *>(p + 16) = u32x2{wave, 0};
        } else {
            // two rounds: even lanes' tuples with their odd neighbours' help, then the odd lanes'
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t owner = (lane & ~1u) | r;
                char *p = out + wave_base + (static_cast<uint64_t>(owner) * gap + t) * 24 + (lane & 1u) * 12;
                *reinterpret_cast<u32x3 *>(p) = u32x3{t, owner, wave};
            }
        }
    }
}

int main() {
    const uint32_t blocks = 256, threads = 1024, tpl = 10;  // 10 tuples per lane and "tile"
    const uint32_t waves = blocks * threads / 64;
    for (uint32_t gap : {10u, 16u}) {
        const size_t bytes = static_cast<size_t>(waves) * 64 * gap * 24;
        char *out;
        CHECK(hipMalloc(&out, bytes));
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                for (int it = 0; it < 20; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, tpl, gap);
                    else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, tpl, gap);
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double tuples = 20.0 * waves * 64 * tpl;
                if (rep == 2) printf("gap %2u tuples  %s: %7.1f G tuples/s  %6.2f TB/s\n", gap, mode == 0 ? "one lane per tuple, 16 + 8 bytes  " : "two lanes per tuple, 12 bytes each",
                                     tuples / ms / 1e6, tuples * 24 / ms / 1e9);
            }
        }
        CHECK(hipFree(out));
    }
    return 0;
}
