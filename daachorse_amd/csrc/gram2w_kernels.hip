// GRAM engine kernels for wide alphabets (gfx950): 31 .. 62 byte classes, context K = 2, 64-bit M words and child bitmaps.
// Same step as gram2_kernels.hip (which see); what differs is spelled out in gram2w.hpp.  Roofline: HBM bytes of haystack.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef uint32_t gw_u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kRingW = 128;
constexpr unsigned long long kMaskW = 0x3fffffffffffffffull;
typedef __attribute__((address_space(3))) const uint32_t ldsw_cu32;
typedef __attribute__((address_space(3))) const uint16_t ldsw_cu16;
typedef __attribute__((address_space(3))) const uint8_t ldsw_cu8;
typedef __attribute__((address_space(3))) const unsigned long long ldsw_cu64;

__device__ __forceinline__ uint32_t wpin(uint32_t x) {
    asm("" : "+v"(x));
    return x;
}
__device__ __forceinline__ unsigned long long gw_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void gw_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

}  // namespace

template <bool EXACT>
__global__ __launch_bounds__(1024) void gram2w_kernel(const Gram2WDev g, const GramArgs a) {
    constexpr int P = 16, K = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t offM = EXACT ? g.off_m_exact : g.off_m_count, offS = EXACT ? g.off_s_exact : g.off_s_count;
    const uint32_t offRing = EXACT ? g.off_ring_exact : g.off_ring_count;
    gw_copy(smem, g.cls, 256);
    gw_copy(smem + offM, g.m, g.m_bytes);
    gw_copy(smem + offS, g.sdir, g.s_bytes);
    if (EXACT) {
        gw_copy(smem + kGram2OffH, g.hsum, g.h_bytes);
        gw_copy(smem + g.off_cid, g.cid4, g.cid_bytes);
    }
    __syncthreads();
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<ldsw_cu8 *>(static_cast<uintptr_t>(byte)); };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsw_cu32 *>(static_cast<uintptr_t>(addr)); };
    auto lds_u64 = [&](uint32_t addr) -> unsigned long long { return *reinterpret_cast<ldsw_cu64 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C = g.C;
    const uint32_t cid_bias = g.off_cid - (offM >> 2);  // CID entry of the 2-gram whose M word is at `am`: (am >> 2) + cid_bias
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    // walkers: {low 32 bits of the virtual position p of the last byte of a 3-gram, the depth-4 state reached on the byte at
    // p + 1 | class of the byte at p + 2 << 26}
    uint2 *__restrict__ slab = a.wq + (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg) * a.wq_slab;
    uint32_t wq_n = 0, slab_hi = 0;

    unsigned long long tot_cnt = 0;
    uint32_t cnt32 = 0, tot_s1 = 0, tot_s2 = 0;

    auto load_chunk = [&](uint64_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const gw_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const gw_u32x4_t *>(hay + v));
        uint4 r{q.x, q.y, q.z, q.w};
        if (v < a.lead || v + 16 > a.vlen) {
            uint32_t w[4] = {r.x, r.y, r.z, r.w};
            for (int b = 0; b < 16; ++b) {
                const uint64_t p = v + b;
                if (p < a.lead || p >= a.vlen) w[b >> 2] = (w[b >> 2] & ~(0xffu << (8 * (b & 3)))) | (g.unused_byte << (8 * (b & 3)));
            }
            r = uint4{w[0], w[1], w[2], w[3]};
        }
        return r;
    };
    auto class_at = [&](uint64_t p) -> uint32_t { return (p >= a.lead && p < a.vlen) ? cls_of(hay[p]) : 0u; };
    auto below64 = [](unsigned long long m, uint32_t k) -> uint32_t { return static_cast<uint32_t>(__popcll(m & ((1ull << k) - 1ull))); };

    auto drain = [&]() {
        for (uint32_t i = lane; i < wq_n; i += 64) {
            const uint2 e = slab[i];
            uint64_t vnext = ((static_cast<uint64_t>(slab_hi) << 32) | e.x) + 2;
            uint4 r = g.drec[e.y & 0x03ffffffu];  // {cmap lo, cmap hi, first_child, own_hsum}
            uint32_t kn = e.y >> 26;
            uint32_t ahead = 0, n_ahead = 0;
            for (;;) {
                cnt32 += r.w != 0;
                if (EXACT) {
                    tot_s1 += r.w;
                    tot_s2 += r.w * static_cast<uint32_t>(vnext - a.lead);
                }
                const unsigned long long cm = (static_cast<unsigned long long>(r.y) << 32) | r.x;
                if (((cm >> kn) & 1ull) == 0) break;
                r = g.drec[r.z + below64(cm, kn)];
                ++vnext;
                if (n_ahead == 0) {
                    if (vnext >= a.lead && vnext + 4 <= a.vlen) {
                        __builtin_memcpy(&ahead, hay + vnext, 4);
                    } else {
                        ahead = 0;
                        for (int b = 3; b >= 0; --b) ahead = (ahead << 8) | ((vnext + b >= a.lead && vnext + b < a.vlen) ? hay[vnext + b] : g.unused_byte);
                    }
                    n_ahead = 4;
                }
                kn = cls_of(ahead & 0xffu);
                ahead >>= 8;
                --n_ahead;
            }
        }
        wq_n = 0;
    };

    // hit ring: entry = {2-gram index (12 bits) | class of the hit byte << 12 | classes of the next two bytes << 18 / << 24, position}
    uint2 *ring = reinterpret_cast<uint2 *>(smem + offRing) + wave_in_wg * kRingW;
    uint32_t q_n = 0;
    uint4 pend = uint4{0u, 0u, 0u, 0u};  // {cmap lo, cmap hi, own_hsum, first_child} read for the previous batch
    uint32_t pend_item = 0, pend_pos = 0;
    bool pend_valid = false;
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;
        cnt32 += r.z != 0;
        if (EXACT) {
            tot_s1 += r.z;
            tot_s2 += r.z * (pend_pos - a.lead + 1u);
        }
        const uint32_t k1 = (pend_item >> 18) & 63u;
        const unsigned long long cm = (static_cast<unsigned long long>(r.y) << 32) | r.x;
        const bool go = (cm >> k1) & 1ull;
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go)
                (slab + wq_n)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0))] =
                    uint2{pend_pos, (r.w + below64(cm, k1)) | (((pend_item >> 24) & 63u) << 26)};
            wq_n += __popcll(m);
        }
    };
    auto deep_rank = [&](uint32_t gi, uint32_t d) -> uint32_t {
        const uint32_t grp = gi & ~3u, idx = gi & 3u;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(smem + offM + grp * 8u), q1 = *reinterpret_cast<const uint4 *>(smem + offM + grp * 8u + 16u);
        const unsigned long long w0 = (static_cast<unsigned long long>(q0.y) << 32) | q0.x, w1 = (static_cast<unsigned long long>(q0.w) << 32) | q0.z,
                                 w2 = (static_cast<unsigned long long>(q1.y) << 32) | q1.x, w3 = (static_cast<unsigned long long>(q1.w) << 32) | q1.z;
        const unsigned long long own = idx == 0 ? w0 : idx == 1 ? w1 : idx == 2 ? w2 : w3;
        uint32_t below = below64(own & kMaskW, d);
        below += idx > 0 ? static_cast<uint32_t>(__popcll(w0 & kMaskW)) : 0u;
        below += idx > 1 ? static_cast<uint32_t>(__popcll(w1 & kMaskW)) : 0u;
        below += idx > 2 ? static_cast<uint32_t>(__popcll(w2 & kMaskW)) : 0u;
        return *reinterpret_cast<const uint32_t *>(smem + offS + (gi >> 2) * 4u) + below;
    };
    auto process_batch = [&]() {
        consume_pending();
        const uint32_t n = q_n < 64u ? q_n : 64u;
        q_n -= n;
        pend = uint4{0u, 0u, 0u, 0u};
        pend_item = 0;
        if (lane < n) {
            const uint2 it = ring[q_n + lane];
            pend_item = it.x;
            pend_pos = it.y;
            pend = g.dhit[deep_rank(it.x & 0xfffu, (it.x >> 12) & 63u)];
        }
        pend_valid = true;
    };

    uint64_t region = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    while (region < a.nregions) {
      slab_hi = static_cast<uint32_t>((region * a.region_bytes) >> 32);
      for (; region < a.nregions && static_cast<uint32_t>((region * a.region_bytes) >> 32) == slab_hi; region += nwaves) {
        const uint64_t rbase = region * a.region_bytes;
        const uint64_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (rbase >= static_cast<uint64_t>(K - i) ? class_at(rbase - (K - i)) : 0u) << (8 * i);
        constexpr uint64_t SB = 64ull * P;
        uint4 pf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) pf[i] = (rbase + SB * i < rend) ? load_chunk(rbase + SB * i + lane * P) : uint4{ub4, ub4, ub4, ub4};

        for (uint64_t sb = rbase; sb < rend; sb += SB) {
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint64_t v = sb + lane * P;
            const uint32_t v32 = static_cast<uint32_t>(v);
            const uint4 cur = pf[0];
            pf[0] = pf[1];
            consume_pending();
            pf[1] = (sb + SB * 2 < rend) ? load_chunk(v + SB * 2) : uint4{ub4, ub4, ub4, ub4};
            uint32_t after2;
            if (sb + SB < rend) {
                after2 = __builtin_amdgcn_readfirstlane(pf[0].x);
            } else {
                const uint64_t p0 = sb + SB, p1 = sb + SB + 1;
                after2 = ((p0 >= a.lead && p0 < a.vlen) ? hay[p0] : g.unused_byte) | (((p1 >= a.lead && p1 < a.vlen) ? hay[p1] : g.unused_byte) << 8);
            }
            uint32_t kx[K + P + 2];
            {
                const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    kx[K + b] = wpin(cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu));
                    __builtin_assume(kx[K + b] < 64u);
                }
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);
            uint32_t left = __shfl_up(pk, 1, 64);
            if (lane == 0) left = carry;
            carry = __builtin_amdgcn_readlane(pk, 63);
#pragma unroll
            for (int i = 0; i < K; ++i) { kx[i] = (left >> (8 * i)) & 0xffu; __builtin_assume(kx[i] < 64u); }
            uint32_t right63 = cls_of(after2 & 0xffu) | (cls_of((after2 >> 8) & 0xffu) << 6);
            asm volatile("" : "+v"(right63));
            uint32_t right = __shfl_down(kx[K] | (kx[K + 1] << 6), 1, 64);
            right = lane == 63 ? right63 : right;
            kx[K + P] = right & 63u;
            kx[K + P + 1] = right >> 6;

            // 2-gram ending at j (j = -1 .. P-1) and the LDS address of its M word
            uint32_t gi[P + 1], am[P + 1];
#pragma unroll
            for (int j = -1; j < P; ++j) {
                gi[j + 1] = __umul24(kx[K + j - 1], C) + kx[K + j];
                am[j + 1] = (wpin(gi[j + 1]) << 3) + offM;
            }
            uint32_t tri[P];  // classes of the bytes at j, j+1, j+2 (6 bits each), rolling from the right
            tri[P - 1] = (((kx[K + P + 1] << 6) | kx[K + P]) << 6) | kx[K + P - 1];
#pragma unroll
            for (int j = P - 2; j >= 0; --j) tri[j] = (tri[j + 1] << 6) | kx[K + j];

            uint32_t ccnt = 0, A = 0, T = 0;
            const uint32_t e0 = static_cast<uint32_t>(v - a.lead) + 1u;
            unsigned long long mprev = lds_u64(am[0]);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const unsigned long long mw = lds_u64(am[j + 1]);
                ccnt += static_cast<uint32_t>(mw >> 62);
                if (EXACT) {
                    const uint32_t id4 = *reinterpret_cast<ldsw_cu16 *>(static_cast<uintptr_t>((am[j + 1] >> 2) + cid_bias));
                    A += lds_u32(id4);
                    T += A;
                }
                const bool hit = (mprev >> kx[K + j]) & 1ull;
                const unsigned long long m = __ballot(hit);
                if (m != 0) {
                    const uint32_t q_s = q_n;
                    if (hit)
                        (ring + q_s)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] =
                            uint2{(tri[j] << 12) | gi[j], v32 + j};
                    q_n = q_s + static_cast<uint32_t>(__popcll(m));
                    if (q_n >= 64u) process_batch();
                }
                mprev = mw;
            }
            cnt32 += ccnt;
            if (EXACT) {
                tot_s1 += A;
                tot_s2 += A * (e0 + static_cast<uint32_t>(P)) - T;
            }
        }
        tot_cnt += cnt32;
        cnt32 = 0;
      }
      while (q_n != 0) process_batch();
      consume_pending();
      drain();
      tot_cnt += cnt32;
      cnt32 = 0;
    }
    // block-level reduction of {count, S1, S2}
    const unsigned long long c = gw_wave_sum(tot_cnt), x1 = gw_wave_sum(tot_s1), x2 = gw_wave_sum(tot_s2);
    __syncthreads();
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(smem);
    if (lane == 0) { scratch[wave_in_wg * 3] = c; scratch[wave_in_wg * 3 + 1] = x1; scratch[wave_in_wg * 3 + 2] = x2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r0 = 0, r1 = 0, r2 = 0;
        for (int w = 0; w < static_cast<int>((blockDim.x + 63) >> 6); ++w) { r0 += scratch[w * 3]; r1 += scratch[w * 3 + 1]; r2 += scratch[w * 3 + 2]; }
        if (r0 | r1 | r2) { atomicAdd(a.result, r0); atomicAdd(a.result + 1, r1); atomicAdd(a.result + 2, r2); }
    }
}

hipError_t launch_gram2w_scan(const Gram2WDev &dev, const GramArgs &a, bool exact, uint32_t blocks, hipStream_t stream) {
    const uint32_t lds = exact ? dev.lds_exact : dev.lds_count;
    hipError_t e;
    if (exact) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram2w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))) != hipSuccess) return e;
        hipLaunchKernelGGL((gram2w_kernel<true>), dim3(blocks), dim3(1024), lds, stream, dev, a);
    } else {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram2w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))) != hipSuccess) return e;
        hipLaunchKernelGGL((gram2w_kernel<false>), dim3(blocks), dim3(1024), lds, stream, dev, a);
    }
    return hipGetLastError();
}

}  // namespace daac
