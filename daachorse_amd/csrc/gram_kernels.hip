// GRAM engine kernels (gfx950): count + checksum of the find_overlapping stream without a state
// chain.  See gram.hpp for the method.  One wavefront streams 1 KiB per step with fully coalesced
// 16-byte loads; every position is independent, so there is no halo and no warm-up:
//
//   per position p (all in LDS): class of the byte, T_{K-1}[last K-1 classes] -> short patterns,
//   W_K bit (+rank -> record) -> patterns of length K, B_{K+1} bit -> "a longer pattern may start
//   K bytes back".  For set B bits the rank directory gives the depth-(K+1) state id; its 16-byte
//   record is the ONLY HBM/L2 access of the fast pass (issued for 8 positions at a time, then
//   consumed).  Branches that continue past depth K+1 are rare; the wave appends them to its own
//   slab (ballot-compacted, no atomics) and finishes them 64 at a time with a goto-only trie walk
//   whenever the slab fills up and at the end of its work.
//
// Roofline: HBM bytes of haystack (1 B read per byte); integer/bit work only, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

typedef uint32_t g_u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long gram_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void gram_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

// Block-level reduction of {count, S1, S2} and one atomic triple per workgroup.
__device__ __forceinline__ void gram_reduce(unsigned long long cnt, uint32_t s1, uint32_t s2, unsigned long long *scratch,
                                            unsigned long long *result) {
    const unsigned long long c = gram_wave_sum(cnt), x1 = gram_wave_sum(s1), x2 = gram_wave_sum(s2);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { scratch[wave * 3] = c; scratch[wave * 3 + 1] = x1; scratch[wave * 3 + 2] = x2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r0 = 0, r1 = 0, r2 = 0;
        for (int w = 0; w < static_cast<int>((blockDim.x + 63) >> 6); ++w) { r0 += scratch[w * 3]; r1 += scratch[w * 3 + 1]; r2 += scratch[w * 3 + 2]; }
        if (r0 | r1 | r2) { atomicAdd(result, r0); atomicAdd(result + 1, r1); atomicAdd(result + 2, r2); }
    }
}

template <int K>
__global__ __launch_bounds__(1024) void gram_count_kernel(const GramDev g, const GramArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gram_copy(smem, g.cls, 256);
    gram_copy(smem + g.off_tshort, g.tshort, g.off_wbits - g.off_tshort);
    gram_copy(smem + g.off_wbits, g.wbits, g.off_wrank - g.off_wbits);
    gram_copy(smem + g.off_wrank, g.wrank, g.off_wown - g.off_wrank);
    gram_copy(smem + g.off_wown, g.wown, g.off_bbits - g.off_wown);
    gram_copy(smem + g.off_bbits, g.bbits, g.off_brank - g.off_bbits);
    gram_copy(smem + g.off_brank, g.brank, g.off_bsuper - g.off_brank);
    gram_copy(smem + g.off_bsuper, g.bsuper, g.off_scratch - g.off_bsuper);
    __syncthreads();
    const uint8_t *l_cls = reinterpret_cast<const uint8_t *>(smem);
    const uint2 *l_short = reinterpret_cast<const uint2 *>(smem + g.off_tshort);
    const uint32_t *l_wbits = reinterpret_cast<const uint32_t *>(smem + g.off_wbits);
    const uint16_t *l_wrank = reinterpret_cast<const uint16_t *>(smem + g.off_wrank);
    const uint2 *l_wown = reinterpret_cast<const uint2 *>(smem + g.off_wown);
    const uint32_t *l_bbits = reinterpret_cast<const uint32_t *>(smem + g.off_bbits);
    const uint16_t *l_brank = reinterpret_cast<const uint16_t *>(smem + g.off_brank);
    const uint32_t *l_bsuper = reinterpret_cast<const uint32_t *>(smem + g.off_bsuper);

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t C = g.C;
    const uint32_t PK1 = K == 3 ? g.CC : g.C;    // C^(K-1)
    const uint32_t PK = K == 3 ? g.CCC : g.CC;   // C^K
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    // this wave's slab of pending walkers: entry = state id | (virtual position of the next byte) << 28
    unsigned long long *__restrict__ slab =
        a.wq + (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * a.wq_slab;
    uint32_t wq_n = 0;  // wave-uniform

    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;

    // chunk at virtual position v (multiple of 16); bytes outside [lead, vlen) become class-0 bytes
    auto load_chunk = [&](uint64_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const g_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const g_u32x4_t *>(hay + v));
        uint4 r{q.x, q.y, q.z, q.w};
        if (v < a.lead || v + 16 > a.vlen) {  // first / last chunk of the haystack only
            uint32_t w[4] = {r.x, r.y, r.z, r.w};
            for (int b = 0; b < 16; ++b) {
                const uint64_t p = v + b;
                if (p < a.lead || p >= a.vlen) w[b >> 2] = (w[b >> 2] & ~(0xffu << (8 * (b & 3)))) | (g.unused_byte << (8 * (b & 3)));
            }
            r = uint4{w[0], w[1], w[2], w[3]};
        }
        return r;
    };
    auto class_at = [&](uint64_t p) -> uint32_t {  // class of the byte at virtual position p
        return (p >= a.lead && p < a.vlen) ? l_cls[hay[p]] : 0u;
    };
    // Finishes the queued branches, 64 per round.  A walker's state was reached by consuming the byte
    // before `vnext`, so its own patterns end at (vnext - lead); it then follows the goto function.
    auto drain = [&]() {
        for (uint32_t i = lane; i < wq_n; i += 64) {
            const unsigned long long w = slab[i];
            uint32_t id = static_cast<uint32_t>(w) & 0x0fffffffu;
            uint64_t vnext = w >> 28;
            for (;;) {
                const uint4 r = g.drec[id];  // {cmap, first_child, own_cnt, own_hsum}
                tot_cnt += r.z;
                tot_s1 += r.w;
                tot_s2 += r.w * static_cast<uint32_t>(vnext - a.lead);
                const uint32_t kn = class_at(vnext);
                if (((r.x >> kn) & 1u) == 0) break;
                id = r.y + __popc(r.x & ((1u << kn) - 1u));
                ++vnext;
            }
        }
        wq_n = 0;
    };

    for (uint64_t region = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6); region < a.nregions;
         region += nwaves) {
        const uint64_t rbase = region * a.region_bytes;
        const uint64_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        // classes of the K bytes before the region, oldest in the low byte
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (rbase >= static_cast<uint64_t>(K - i) ? class_at(rbase - (K - i)) : 0u) << (8 * i);

        uint4 cur = load_chunk(rbase + lane * 16);
        for (uint64_t sb = rbase; sb < rend; sb += 1024) {
            if (wq_n + 1024u > a.wq_slab) drain();  // a step can add at most 16 x 64 walkers
            const uint64_t v = sb + lane * 16;
            const uint4 nxt = (sb + 1024 < rend) ? load_chunk(v + 1024) : uint4{ub4, ub4, ub4, ub4};
            // ---- byte classes of this lane's 16 positions plus K to the left and 1 to the right ----
            uint32_t kx[K + 17];
            {
                const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int b = 0; b < 16; ++b) kx[K + b] = l_cls[(w[b >> 2] >> (8 * (b & 3))) & 0xffu];
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[16 + i] << (8 * i);  // this lane's last K classes, oldest low
            uint32_t left = __shfl_up(pk, 1, 64);
            if (lane == 0) left = carry;
            carry = __shfl(pk, 63, 64);
#pragma unroll
            for (int i = 0; i < K; ++i) kx[i] = (left >> (8 * i)) & 0xffu;
            uint32_t right = __shfl_down(kx[K], 1, 64);
            if (lane == 63) right = class_at(sb + 1024);
            kx[K + 16] = right;

            // ---- fast path: every position independently ------------------------------------------------
            uint32_t ccnt = 0, A = 0, T = 0;       // T = sum over positions of running A (prefix trick for h * end)
            uint32_t hitmask = 0;                  // B_{K+1} hits of this lane
            uint32_t iB[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t iS = 0;
#pragma unroll
                for (int t = 0; t < K - 1; ++t) iS = iS * C + kx[j + 2 + t];     // K-1 classes ending at j
                const uint32_t iW = kx[j + 1] * PK1 + iS;                          // K classes
                iB[j] = kx[j] * PK + iW;                                           // K+1 classes
                uint32_t hs = 0;
                if (g.has_short) {
                    const uint2 t = l_short[iS];
                    ccnt += t.x;
                    hs = t.y;
                }
                if (g.has_word) {
                    const uint32_t ww = l_wbits[iW >> 5];
                    if ((ww >> (iW & 31)) & 1u) {
                        const uint2 o = l_wown[l_wrank[iW >> 5] + __popc(ww & ((1u << (iW & 31)) - 1u))];
                        ccnt += o.x;
                        hs += o.y;
                    }
                }
                A += hs;
                T += A;
                const uint32_t bw = l_bbits[iB[j] >> 5];
                hitmask |= ((bw >> (iB[j] & 31)) & 1u) << j;
            }
            // sum_j hs_j * (e0 + j) with e0 = end of position 0 = (v - lead) + 1:  A * (e0 + 16) - T
            const uint32_t e0 = static_cast<uint32_t>(v - a.lead) + 1u;
            uint32_t s1 = A, s2 = A * (e0 + 16u) - T;

            // ---- deep path: one 16-byte record per B hit, 8 positions in flight ---------------------------
            if (__any(hitmask != 0)) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    uint4 rec[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = half * 8 + jj;
                        rec[jj] = uint4{0, 0, 0, 0};
                        if ((hitmask >> j) & 1u) {
                            const uint32_t w = iB[j] >> 5;
                            const uint32_t bw = l_bbits[w];
                            const uint32_t id = g.level_start + l_bsuper[w >> 6] + l_brank[w] + __popc(bw & ((1u << (iB[j] & 31)) - 1u));
                            rec[jj] = g.drec[id];
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = half * 8 + jj;
                        if ((hitmask >> j) & 1u) {
                            const uint4 r = rec[jj];  // {cmap, first_child, own_cnt, own_hsum}
                            ccnt += r.z;
                            s1 += r.w;
                            s2 += r.w * (e0 + j);
                        }
                    }
                    // rare: the branch goes on past depth K+1 -> queue a walker (wave-ballot compaction)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = half * 8 + jj;
                        const uint32_t kn = kx[K + j + 1];
                        const bool go = ((hitmask >> j) & 1u) && ((rec[jj].x >> kn) & 1u);
                        const unsigned long long m = __ballot(go);
                        if (m != 0) {
                            if (go) {
                                const uint32_t child = rec[jj].y + __popc(rec[jj].x & ((1u << kn) - 1u));
                                const uint32_t slot = wq_n + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                                slab[slot] = static_cast<unsigned long long>(child) | ((v + j + 2) << 28);
                            }
                            wq_n += __popcll(m);
                        }
                    }
                }
            }
            tot_cnt += ccnt;
            tot_s1 += s1;
            tot_s2 += s2;
            cur = nxt;
        }
    }
    drain();
    gram_reduce(tot_cnt, tot_s1, tot_s2, reinterpret_cast<unsigned long long *>(smem), a.result);
}

hipError_t launch_gram_scan(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    hipError_t e;
    if (dev.K == 3) {
        if (dev.lds_bytes > 64 * 1024 &&
            (e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram_count_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(dev.lds_bytes))) != hipSuccess)
            return e;
        hipLaunchKernelGGL(gram_count_kernel<3>, dim3(blocks), dim3(threads), dev.lds_bytes, stream, dev, a);
    } else {
        if (dev.lds_bytes > 64 * 1024 &&
            (e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram_count_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(dev.lds_bytes))) != hipSuccess)
            return e;
        hipLaunchKernelGGL(gram_count_kernel<2>, dim3(blocks), dim3(threads), dev.lds_bytes, stream, dev, a);
    }
    return hipGetLastError();
}

}  // namespace daac
