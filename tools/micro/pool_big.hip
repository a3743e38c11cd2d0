// Is a stream-ordered pool allocation of more than 32 GiB whole?  (ROCm 7.2, MI355X)  hipcc --offload-arch=gfx950 -O2 -o /tmp/pool_big tools/micro/pool_big.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void touch(char *p, unsigned long long bytes, unsigned long long step) {
    const unsigned long long i = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * step;
    if (i < bytes) p[i] = 1;
}
int main(int argc, char **argv) {
    hipStream_t s; hipStreamCreate(&s);
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    unsigned long long keep = 32ull << 30; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    for (unsigned long long gib : {8ull, 30ull, 34ull, 41ull, 64ull}) {
        for (int async = 1; async >= 0; --async) {
            char *p = nullptr;
            const unsigned long long bytes = gib << 30;
            hipError_t e = async ? hipMallocAsync(reinterpret_cast<void **>(&p), bytes, s) : hipMalloc(reinterpret_cast<void **>(&p), bytes);
            printf("%s %llu GiB: %s", async ? "hipMallocAsync" : "hipMalloc     ", gib, hipGetErrorString(e)); fflush(stdout);
            if (e != hipSuccess) { printf("\n"); continue; }
            const unsigned long long step = 4096, n = bytes / step;
            hipLaunchKernelGGL(touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, bytes, step);
            e = hipStreamSynchronize(s);
            printf("  touched every page: %s\n", hipGetErrorString(e)); fflush(stdout);
            if (async) hipFreeAsync(p, s); else hipFree(p);
            hipStreamSynchronize(s);
        }
    }
    return 0;
}
