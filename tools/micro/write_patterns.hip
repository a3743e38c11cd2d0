// Write-bandwidth of the tuple emitter's store pattern in isolation (round 4): what does the memory system give a kernel that writes
// ~10 KB contiguous per wave and tile, 16 bytes per lane, 64 consecutive slots per store instruction?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/write_patterns.hip -o tools/micro/bin/write_patterns && tools/micro/bin/write_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: every wave-instruction writes 1 KiB at a 1 KiB-aligned address (grid-stride)   = fill
// mode 1: the same, everything shifted by `shift` bytes (16-byte aligned only)
// mode 2: tiles: tile t = slots [off[t], off[t+1]) of 16 bytes; a wave writes a tile with ceil(n / 64) store instructions from its first slot on
// mode 3: tiles, store instructions aligned to 128-byte lines (a short head, then whole lines)
// mode 4: mode 2 + a 1 KiB read per tile first (the stream bytes), waited for before the stores
// mode 5: mode 3 + that read
__global__ __launch_bounds__(256) void wr(char *out, const unsigned long long *off, uint32_t ntiles, int mode, uint32_t shift, const u32x4 *src, u32x4 *sink) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    u32x4 v = {lane, wave, 3u, 4u};
    if (mode <= 1) {
        const unsigned long long total = off[ntiles];   // slots
        for (unsigned long long s = (unsigned long long)wave * 64 + lane; s < total; s += (unsigned long long)nwaves * 64)
            *reinterpret_cast<u32x4 *>(out + shift + s * 16) = v;
        return;
    }
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        const unsigned long long b = off[t];
        const uint32_t n = (uint32_t)(off[t + 1] - b);
        if (mode >= 4) { const u32x4 r = src[(size_t)t * 64 + lane]; acc += r; v.z = acc.x; }
        const uint32_t head = (mode == 3 || mode == 5) ? (uint32_t)(b & 7) : 0;
        for (uint32_t s0 = 0; s0 < n + head; s0 += 64) {
            const uint32_t s = s0 + lane - head;
            if (s < n) *reinterpret_cast<u32x4 *>(out + (b + s) * 16) = v;
        }
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}

int main() {
    const uint32_t ntiles = 1u << 20;
    std::vector<unsigned long long> off(ntiles + 1);
    unsigned long long run = 0; uint32_t x = 12345;
    for (uint32_t t = 0; t < ntiles; ++t) { off[t] = run; x = x * 1664525u + 1013904223u; run += 580 + (x >> 26); }   // 580 .. 643 slots per tile
    off[ntiles] = run;
    const size_t bytes = run * 16 + 4096;
    char *out; unsigned long long *doff; u32x4 *src, *sink;
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&doff, (ntiles + 1) * 8)); CK(hipMalloc(&src, (size_t)ntiles * 1024)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(doff, off.data(), (ntiles + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemset(src, 1, (size_t)ntiles * 1024));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"fill, 1 KiB-aligned instructions", "fill, shifted by 16 bytes", "tiles (~10 KB per wave and tile), instructions from the tile's first slot",
                           "tiles, instructions aligned to 128-byte lines", "tiles + 1 KiB read per tile", "tiles aligned + read"};
    for (int wpc : {16, 12, 8}) {
        for (int mode = 0; mode < 6; ++mode) {
            const uint32_t blocks = 256u * wpc / 4;
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, out, doff, ntiles, mode, mode == 1 ? 16u : 0u, src, sink);
                hipEventRecord(e1);
                CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("waves/CU %2d  %-76s %7.3f ms  %7.1f GB/s written\n", wpc, names[mode], best, run * 16 / best / 1e6);
        }
    }
    return 0;
}
