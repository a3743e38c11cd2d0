// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns this repo uses: a known
// number of bytes is streamed once with (a) plain and (b) non-temporal 16-byte-per-lane loads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o c -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(1024) void stream_read(const u32x4 *p, uint64_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x1234567u) out[0] = acc;
}

int main() {
    const uint64_t bytes = 2ull << 30;
    u32x4 *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        stream_read<false><<<2048, 1024>>>(buf, bytes / 16, out);
        stream_read<true><<<2048, 1024>>>(buf, bytes / 16, out);
    }
    hipDeviceSynchronize();
    printf("streamed %llu bytes per launch (plain, nt) x2\n", (unsigned long long)bytes);
    return 0;
}
