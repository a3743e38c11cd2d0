// Microbenchmark: issue rate of integer VALU instructions on gfx950 (wave64), 16 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_bench tools/micro/valu_bench.hip && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void valu_kernel(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a[8];
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 977u + k + seed; f[k] = a[k] * 1e-9f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) a[k] = __umul24(a[k], seed) + 7u;                 // v_mad_u32_u24
            if (OP == 1) a[k] = (a[k] << 3) | seed;                        // v_lshl_or_b32
            if (OP == 2) a[k] = (a[k] >> 5) & seed;                        // v_lshrrev + v_and (2 instr) / v_bfe
            if (OP == 3) a[k] = a[k] + seed;                               // v_add_u32
            if (OP == 4) f[k] = f[k] * 1.0001f + 0.5f;                     // v_fma_f32
            if (OP == 5) a[k] = a[k] * seed;                               // v_mul_lo_u32
            if (OP == 6) a[k] = __popc(a[k]) + seed;                       // v_bcnt_u32_b32
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += a[k] + static_cast<uint32_t>(f[k]);
    if (r == 0x12345u) out[0] = r;
}

template <int OP>
void run(const char *name) {
    uint32_t *out;
    hipMalloc(&out, 4);
    const int iters = 4096, blocks = 256 * 2, threads = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    valu_kernel<OP><<<blocks, threads>>>(out, 3, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    valu_kernel<OP><<<blocks, threads>>>(out, 3, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = double(blocks) * threads / 64 * iters * 8;
    printf("%-28s %8.1f G wave-instr/s  -> %.2f cycles per wave-instr per SIMD at 2.4 GHz (1024 SIMDs)\n", name, winst / ms / 1e6,
           1024 * 2.4e9 / (winst / ms * 1e3));
    hipFree(out);
}

int main() {
    run<0>("v_mad_u32_u24");
    run<1>("v_lshl_or_b32");
    run<2>("shift+and (1-2 instr)");
    run<3>("v_add_u32");
    run<4>("v_fma_f32");
    run<5>("v_mul_lo_u32");
    run<6>("v_bcnt_u32_b32 (+add)");
    return 0;
}
