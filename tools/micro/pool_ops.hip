// What the stream-ordered pool's calls cost on the host (ROCm 7.2, MI355X): hipMallocAsync / hipFreeAsync of sizes a scan call asks for, with a
// kernel in flight on the stream and without; hipMalloc / hipFree beside them.  hipcc --offload-arch=gfx950 -O2 -o /tmp/pool_ops tools/micro/pool_ops.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long *p, unsigned n) { unsigned long long x = 0; for (unsigned i = 0; i < n; ++i) x += __builtin_readcyclecounter(); if (x == 1) *p = x; }
static double us(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    unsigned long long keep = 32ull << 30; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    unsigned long long *d; hipMalloc(&d, 8);
    const size_t sizes[] = {24, 1 << 20, 64 << 20, 512ull << 20, 1200ull << 20};
    for (int busy = 0; busy < 2; ++busy)
        for (size_t sz : sizes) {
            for (int rep = 0; rep < 4; ++rep) {
                if (busy) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000u);
                auto t0 = std::chrono::steady_clock::now();
                void *p = nullptr; hipMallocAsync(&p, sz, s);
                double a = us(t0);
                hipMemsetAsync(p, 0, 64, s);
                auto t1 = std::chrono::steady_clock::now();
                hipFreeAsync(p, s);
                double f = us(t1);
                auto t2 = std::chrono::steady_clock::now();
                hipStreamSynchronize(s);
                double y = us(t2);
                if (rep) printf("busy=%d %10zu B  rep %d: mallocAsync %7.1f us  freeAsync %7.1f us  sync %7.1f us\n", busy, sz, rep, a, f, y);
            }
        }
    // three buffers at once, as a scan call holds them
    for (int rep = 0; rep < 4; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        void *p[3]; hipMallocAsync(&p[0], 1200ull << 20, s); hipMallocAsync(&p[1], 500ull << 20, s); hipMallocAsync(&p[2], 450ull << 20, s);
        double a = us(t0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 20000u);
        unsigned long long r; hipMemcpyAsync(&r, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        for (auto q : p) hipFreeAsync(q, s);
        double f = us(t1);
        auto t2 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        printf("three buffers rep %d: mallocs %7.1f us  frees %7.1f us  device sync %7.1f us\n", rep, a, f, us(t2));
    }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        void *p; hipMalloc(&p, 1200ull << 20);
        double a = us(t0);
        auto t1 = std::chrono::steady_clock::now();
        hipFree(p);
        printf("hipMalloc 1200 MiB %7.1f us  hipFree %7.1f us\n", a, us(t1));
    }
    // pageable against pinned 24-byte read-back
    unsigned long long r[3], *pin; hipHostMalloc(&pin, 64);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipMemcpyAsync(r, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        double a = us(t0);
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpyAsync(pin, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        printf("read-back of 8 bytes: pageable %7.1f us  pinned %7.1f us\n", a, us(t1));
    }
    return 0;
}
