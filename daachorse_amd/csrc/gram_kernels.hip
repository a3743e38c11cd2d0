// GRAM engine kernels (gfx950): count + checksum of the find_overlapping stream without a state
// chain.  See gram.hpp for the method.  One wavefront streams 1 KiB per step with fully coalesced
// 16-byte non-temporal loads (one chunk ahead per lane; sixteen waves per CU keep enough bytes in flight);
// every position is independent, so there is no halo and no warm-up:
//
//   per position p (all in LDS): class of the byte, CID[last K classes] -> COMBO[id] = count and
//   h32 sum of every pattern of length <= K ending here, B_{K+1} bit -> "a longer pattern may start
//   K bytes back".  Positions whose B bit is set are rare enough (11 % on the 100k-word automaton)
//   that handling them in place would waste most lanes, so the wave COMPACTS them: a wave ballot
//   gives every hit a slot in a 128-entry stack in LDS, and whenever 64 are queued all 64 lanes take
//   one each: rank directory -> depth-(K+1) state -> its 8-byte {child bitmap, own h32 sum} record,
//   the ONLY HBM/L2 access of the fast pass, consumed one batch later so its latency hides behind
//   LDS work.  Branches that go on past depth K+1 are rarer still (0.6 %); the state they reach and the
//   class of the byte after next go to the wave's slab in HBM (ballot-compacted, no atomics) and are
//   finished 64 at a time by a goto-only trie walk whenever the slab fills up and at the end of the
//   wave's work, without touching the haystack again unless a branch survives two more levels.
//
// Roofline: HBM bytes of haystack (1 B read per byte); integer/bit work only, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

typedef uint32_t g_u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kGroup = 4;     // positions whose LDS reads are issued together (lgkmcnt tracks at most 15 reads)
constexpr uint32_t kRing = 128;  // entries of a wave's hit stack in LDS (at most 63 left over + 64 new)
constexpr int kPrefetch = 1;  // haystack chunks in flight per lane beyond the current one
constexpr uint32_t kOffBbits = 1024;  // LDS layout: 256 classes as u32, then the B bitmap (api_upload.hip)
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;

// 24-bit multiply-adds, spelled out: left to itself hipcc turns some `__umul24(a, b) + c` of the gram
// indices into the quarter-rate v_mad_u64_u32.
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b_uniform, uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t mad_i24(uint32_t a, int32_t b_uniform, uint32_t c) {
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}

__device__ __forceinline__ uint32_t lshl_or(uint32_t a, uint32_t shift, uint32_t c) {  // (a << shift) | c in one instruction
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(shift), "v"(c));
    return d;
}

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

// P = positions (bytes) a lane takes per step: 16, or 32 for automata without short patterns (their step is so
// cheap that the per-step work — neighbour exchange, prefetch bookkeeping — and the load latency show)
template <int K, bool HAS_SHORT, int TPB, bool DENSE, bool RANK_LDS, int P>
__global__ __launch_bounds__(TPB) void gram_count_kernel(const GramDev g, const GramArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gram_copy(smem, g.cls32, 1024);
    gram_copy(smem + kOffBbits, g.bbits, g.off_cid - kOffBbits);
    if (HAS_SHORT) {
        gram_copy(smem + g.off_cid, g.cid, g.off_combo - g.off_cid);
        gram_copy(smem + g.off_combo, g.combo, g.off_brank - g.off_combo);
    }
    if (RANK_LDS) {
        gram_copy(smem + g.off_brank, g.brank, g.off_bsuper - g.off_brank);
        gram_copy(smem + g.off_bsuper, g.bsuper, g.off_scratch - g.off_bsuper);
    }
    __syncthreads();
    // classes and bitmap words are read through absolute LDS addresses (the dynamic segment starts where the
    // static one ends; this kernel has none): a byte select + shift is the whole address of a class
    // (spelled as the literal 0 so that it folds into the instructions; checked once per workgroup)
    constexpr uint32_t lds0 = 0;
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<lds_cu32 *>(static_cast<uintptr_t>(lds0 + byte * 4u)); };
    const uint16_t *l_cid = reinterpret_cast<const uint16_t *>(smem + g.off_cid);
    const uint2 *l_combo = reinterpret_cast<const uint2 *>(smem + g.off_combo);
    const uint32_t *l_bbits = reinterpret_cast<const uint32_t *>(smem + kOffBbits);
    // the rank directory of B is only touched on hits: small automata keep it in L2 so that two
    // workgroups fit one CU's LDS
    // (compile-time choice: a run-time select would turn these into generic pointers and flat loads)
    const uint8_t *l_brank = reinterpret_cast<const uint8_t *>(smem + g.off_brank);
    const uint32_t *l_bsuper = reinterpret_cast<const uint32_t *>(smem + g.off_bsuper);

    const uint32_t lane = threadIdx.x & 63;
    // the wave's index, as a scalar: everything derived from it (regions, step bounds, fill levels of the hit stack
    // and the slab) then lives in SGPRs and the loops branch on SCC instead of masking lanes
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C = g.C;
    const uint32_t PK = K == 3 ? g.CCC : g.CC;   // C^K
    const int32_t negPK = -static_cast<int32_t>(PK);
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    // this wave's slab of pending walkers: {low 32 bits of the virtual position p of the last byte of a
    // (K+1)-gram, the depth-(K+2) state reached on the byte at p + 1 | class of the byte at p + 2 << 27};
    // all entries of a slab share the bits of p above 2^32 (slab_hi): it is emptied before they change
    uint2 *__restrict__ slab =
        a.wq + (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg) * a.wq_slab;
    uint32_t wq_n = 0, slab_hi = 0;  // wave-uniform

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
        return (p >= a.lead && p < a.vlen) ? cls_of(hay[p]) : 0u;
    };
    // offset within its level of the depth-(K+1) state whose gram index is `ib` (its B bit is set):
    // popcount directory = u32 per 256 bits + u8 per 64 bits + the bits below inside the 64-bit pair
    auto deep_rank = [&](uint32_t ib) -> uint32_t {
        const uint32_t w = ib >> 5;
        const uint2 pair = *reinterpret_cast<const uint2 *>(l_bbits + (w & ~1u));
        const bool odd = w & 1u;
        const uint32_t word = odd ? pair.y : pair.x;
        const uint32_t below = (odd ? __popc(pair.x) : 0u) + __popc(word & ((1u << (ib & 31u)) - 1u));
        return RANK_LDS ? l_bsuper[w >> 3] + l_brank[w >> 1] + below : g.bsuper[w >> 3] + g.brank[w >> 1] + below;
    };
    // Finishes the queued branches, 64 per round.  The entry names the depth-(K+2) state the branch has
    // reached and the class of the next byte, so the haystack is only read again for the few branches
    // that survive yet another level (2e-4 of the positions on the 100k-word automaton).
    auto drain = [&]() {
        for (uint32_t i = lane; i < wq_n; i += 64) {
            const uint2 e = slab[i];
            uint64_t vnext = ((static_cast<uint64_t>(slab_hi) << 32) | e.x) + 2;  // the state consumed the byte before vnext
            uint4 r = g.drec[e.y & 0x07ffffffu];  // {cmap, first_child, own_cnt, own_hsum}
            uint32_t kn = e.y >> 27;
            // bytes ahead of the walk, four per read: on text made of dictionary words a walker lives for several
            // levels, and every separate byte read is one more uncoalesced request to the memory pipeline
            uint32_t ahead = 0, n_ahead = 0;
            for (;;) {
                tot_cnt += r.z;                   // its own patterns end at vnext - lead
                tot_s1 += r.w;
                tot_s2 += r.w * static_cast<uint32_t>(vnext - a.lead);
                if (((r.x >> kn) & 1u) == 0) break;
                r = g.drec[r.y + __popc(r.x & ((1u << kn) - 1u))];
                ++vnext;
                if (r.y >> 31) {
                    // a tail record (gram.cpp, round 6): what is left below this state is one path of r.y & 15 edges with one pattern end, at its
                    // node (r.y >> 4) & 15; {h of that pattern, -, path bytes} against the next eight text bytes in one step
                    unsigned long long text;
                    if (vnext >= a.lead && vnext + 8 <= a.vlen) {
                        __builtin_memcpy(&text, hay + vnext, 8);
                    } else {
                        text = 0;
                        for (int b = 7; b >= 0; --b) text = (text << 8) | ((vnext + b >= a.lead && vnext + b < a.vlen) ? hay[vnext + b] : g.unused_byte);
                    }
                    const unsigned long long diff = ((static_cast<unsigned long long>(r.w) << 32) | r.z) ^ text;
                    const uint32_t edges = r.y & 15u, at = (r.y >> 4) & 15u;
                    uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
                    same = same < edges ? same : edges;
                    if (at <= same) {
                        tot_cnt += 1;
                        tot_s1 += r.x;
                        tot_s2 += r.x * static_cast<uint32_t>(vnext + at - a.lead);
                    }
                    break;
                }
                if (n_ahead == 0) {
                    if (vnext >= a.lead && vnext + 4 <= a.vlen) {
                        __builtin_memcpy(&ahead, hay + vnext, 4);  // one (unaligned) dword
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

    // ---- the hit ring: entry = {gram index | classes of the next two bytes << 20 / << 25, low 32 bits of the position}
    // (a stack: batches are taken from the top, so no wrap-around arithmetic; order does not matter)
    uint2 *ring = reinterpret_cast<uint2 *>(smem + g.off_scratch) + wave_in_wg * kRing;
    uint32_t q_n = 0;                           // wave-uniform
    uint4 pend = uint4{0u, 0u, 0u, 0u};         // record read for the previous batch, not yet consumed
    uint32_t pend_item = 0, pend_pos = 0, pend_rank = 0;
    bool pend_valid = false;                    // wave-uniform
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;                   // {cmap, own h32 sum, first child, -}; zero for idle lanes
        tot_cnt += r.y != 0;
        tot_s1 += r.y;
        tot_s2 += r.y * (pend_pos - a.lead + 1u);  // end = position - lead + 1 (mod 2^32)
        const uint32_t k1 = (pend_item >> 20) & 31u;
        const bool go = (r.x >> k1) & 1u;
        const unsigned long long m = __ballot(go);
        if (m != 0) {  // the branch goes on past depth K+1 -> queue a walker (same slab epoch: see the step loop)
            if (go)
                (slab + wq_n)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0))] =
                    uint2{pend_pos, (r.z + __popc(r.x & ((1u << k1) - 1u))) | ((pend_item >> 25) << 27)};
            wq_n += __popcll(m);
        }
    };
    // takes up to 64 queued hits, one per lane: issue their record reads, retire the previous batch
    auto process_batch = [&]() {
        // a wave with records in flight runs at raised priority until it has retired them at the top of its next
        // step: the four waves of a SIMD stop marching in phase (measured +3-4 %; either polarity works, a static
        // per-wave priority does not)
        __builtin_amdgcn_s_setprio(2);
        consume_pending();
        const uint32_t n = q_n < 64u ? q_n : 64u;
        q_n -= n;
        pend = uint4{0u, 0u, 0u, 0u};
        pend_item = 0;
        if (lane < n) {
            const uint2 it = ring[q_n + lane];
            pend_item = it.x;
            pend_pos = it.y;
            pend_rank = deep_rank(it.x & 0xfffffu);
            pend = g.dhit4[pend_rank];
        }
        pend_valid = true;
    };

    // Regions (multiples of 1 KiB) never straddle a multiple of 4 GiB.  The wave works through its regions one
    // 4 GiB epoch at a time and retires everything queued before moving on, so that every position in the hit
    // stack, the pending batch and the walker slab shares its bits above 2^32 (slab_hi).
    uint64_t region = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    while (region < a.nregions) {
      slab_hi = static_cast<uint32_t>((region * a.region_bytes) >> 32);
      for (; region < a.nregions && static_cast<uint32_t>((region * a.region_bytes) >> 32) == slab_hi; region += nwaves) {
        const uint64_t rbase = region * a.region_bytes;
        const uint64_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        // classes of the K bytes before the region, oldest in the low byte
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (rbase >= static_cast<uint64_t>(K - i) ? class_at(rbase - (K - i)) : 0u) << (8 * i);

        constexpr int Q = P / 16;               // 16-byte loads per lane and step
        constexpr uint64_t SB = 64ull * P;      // bytes a wave takes per step
        uint4 pf[kPrefetch + 1][Q];
#pragma unroll
        for (int i = 0; i <= kPrefetch; ++i)
#pragma unroll
            for (int h = 0; h < Q; ++h)
                pf[i][h] = (rbase + SB * i < rend) ? load_chunk(rbase + SB * i + lane * P + 16 * h) : uint4{ub4, ub4, ub4, ub4};

        for (uint64_t sb = rbase; sb < rend; sb += SB) {
            // a walker is queued when its hit is retired, and a step can retire every hit it makes (64 * P) plus what
            // the step before left in the stack (< 64) and in flight (64)
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint64_t v = sb + lane * P;
            const uint32_t v32 = static_cast<uint32_t>(v);
            uint4 cur[Q];
#pragma unroll
            for (int h = 0; h < Q; ++h) cur[h] = pf[0][h];
#pragma unroll
            for (int i = 0; i < kPrefetch; ++i)
#pragma unroll
                for (int h = 0; h < Q; ++h) pf[i][h] = pf[i + 1][h];
            consume_pending();  // before the next chunk is requested: loads retire in order, the batch's records must not queue behind it
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int h = 0; h < Q; ++h)
                pf[kPrefetch][h] = (sb + SB * (kPrefetch + 1) < rend) ? load_chunk(v + SB * (kPrefetch + 1) + 16 * h) : uint4{ub4, ub4, ub4, ub4};

            // the two bytes after this wave's share (lane 63 needs their classes): lane 0's part of the chunk that
            // is already in flight or, on the last step of a region, two wave-uniform loads
            uint32_t after2;
            if (sb + SB < rend) {
                after2 = __builtin_amdgcn_readfirstlane(pf[0][0].x);
            } else {
                const uint64_t p0 = sb + SB, p1 = sb + SB + 1;
                after2 = ((p0 >= a.lead && p0 < a.vlen) ? hay[p0] : g.unused_byte) | (((p1 >= a.lead && p1 < a.vlen) ? hay[p1] : g.unused_byte) << 8);
            }

            // ---- byte classes of this lane's P positions plus K to the left and 2 to the right ----
            uint32_t kx[K + P + 2];
            {
                uint32_t w[4 * Q];
#pragma unroll
                for (int h = 0; h < Q; ++h) { w[4 * h] = cur[h].x; w[4 * h + 1] = cur[h].y; w[4 * h + 2] = cur[h].z; w[4 * h + 3] = cur[h].w; }
#pragma unroll
                for (int b = 0; b < P; ++b) kx[K + b] = cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);  // this lane's last K classes, oldest low
            uint32_t left = __shfl_up(pk, 1, 64);
            if (lane == 0) left = carry;
            carry = __shfl(pk, 63, 64);
#pragma unroll
            for (int i = 0; i < K; ++i) kx[i] = (left >> (8 * i)) & 0xffu;
            uint32_t right = __shfl_down(kx[K] | (kx[K + 1] << 5), 1, 64);  // the two classes after this lane's P
            uint32_t right63 = cls_of(after2 & 0xffu) | (cls_of((after2 >> 8) & 0xffu) << 5);  // same address in every lane: a broadcast
            asm volatile("" : "+v"(right63));  // keep the two reads out of a lane-63-only branch: the step stays one basic block
            right = lane == 63 ? right63 : right;
            kx[K + P] = right & 31u;
            kx[K + P + 1] = right >> 5;

            // ---- per group of kGroup positions: LDS work of every position independently (all reads of
            // the group in flight before the first is consumed), then the B hits are queued ---------------
            uint32_t ccnt = 0, A = 0, T = 0;       // T = sum over positions of running A (prefix trick for h * end)
            const uint32_t e0 = static_cast<uint32_t>(v - a.lead) + 1u;  // end of this lane's position 0
            // gram indices as a chain: the (K+1)-gram ending at j is the K-gram ending at j - 1 shifted by one
            // class, and dropping its oldest class leaves the K-gram ending at j (two 24-bit mads per position)
            uint32_t wprev = mad_u24(kx[0], C, kx[1]);                               // K classes ending before position 0
            if (K == 3) wprev = mad_u24(wprev, C, kx[2]);
#pragma unroll
            for (int grp = 0; grp < P / kGroup; ++grp) {
                uint32_t iW[kGroup], iB[kGroup], bw[kGroup], id[kGroup];
                uint2 co[kGroup];
#pragma unroll
                for (int jj = 0; jj < kGroup; ++jj) {
                    const int j = grp * kGroup + jj;
                    iB[jj] = mad_u24(wprev, C, kx[j + K]);                           // K+1 classes ending at j
                    wprev = mad_i24(kx[j], negPK, iB[jj]);
                    iW[jj] = wprev;                                                   // K classes ending at j
                }
#pragma unroll
                for (int jj = 0; jj < kGroup; ++jj) {
                    if (HAS_SHORT) id[jj] = l_cid[iW[jj]];
                    bw[jj] = *reinterpret_cast<lds_cu32 *>(static_cast<uintptr_t>(lds0 + kOffBbits + ((iB[jj] >> 3) & ~3u)));
                }
                if (HAS_SHORT) {
#pragma unroll
                    for (int jj = 0; jj < kGroup; ++jj) co[jj] = l_combo[id[jj]];
#pragma unroll
                    for (int jj = 0; jj < kGroup; ++jj) {
                        ccnt += co[jj].x;
                        A += co[jj].y;
                        T += A;
                    }
                }
                // a hit is queued with the classes of the next two bytes; the slot comes from the wave ballot
                auto queue_hit = [&](int jj, bool hit) {
                    const int j = grp * kGroup + jj;
                    const unsigned long long m = __ballot(hit);
                    if (m != 0) {  // wave-uniform
                        const uint32_t q_s = q_n;  // wave-uniform (a scalar register: everything it is computed from is)
                        if (hit) {
                            const uint32_t nx = lshl_or(kx[K + j + 2], 5, kx[K + j + 1]);
                            (ring + q_s)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] =
                                uint2{lshl_or(nx, 20, iB[jj]), v32 + j};
                        }
                        q_n = q_s + static_cast<uint32_t>(__popcll(m));
                        if (q_n >= 64u) process_batch();
                    }
                };
                if (DENSE) {  // B hits on most steps anyway: no point in testing the group first
#pragma unroll
                    for (int jj = 0; jj < kGroup; ++jj) queue_hit(jj, __builtin_amdgcn_ubfe(bw[jj], iB[jj], 1) != 0);
                } else {
                    uint32_t hits = 0;
#pragma unroll
                    for (int jj = 0; jj < kGroup; ++jj) hits |= __builtin_amdgcn_ubfe(bw[jj], iB[jj], 1) << jj;  // bit (iB & 31) of the word
                    if (__any(hits != 0)) {  // one branch per group when nothing hits (sparse automata)
#pragma unroll
                        for (int jj = 0; jj < kGroup; ++jj) queue_hit(jj, (hits >> jj) & 1u);
                    }
                }
            }
            // sum_j hs_j * (e0 + j):  A * (e0 + P) - T
            tot_cnt += ccnt;
            tot_s1 += A;
            tot_s2 += A * (e0 + static_cast<uint32_t>(P)) - T;
        }
      }
      while (q_n != 0) process_batch();
      consume_pending();
      drain();
    }
    gram_reduce(tot_cnt, tot_s1, tot_s2, reinterpret_cast<unsigned long long *>(smem), a.result);
}

template <int K, bool S, int TPB, bool DENSE, bool RL, int P>
static hipError_t launch_p(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    if (dev.lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram_count_kernel<K, S, TPB, DENSE, RL, P>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dev.lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gram_count_kernel<K, S, TPB, DENSE, RL, P>), dim3(blocks), dim3(threads), dev.lds_bytes, stream, dev, a);
    return hipGetLastError();
}
template <int K, bool S, int TPB, bool DENSE, bool RL>
static hipError_t launch_rl(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    // 32 positions per lane only where the step is light: no short patterns, 1024-thread workgroups
    if (!S && TPB == 1024 && a.ppl == 32) return launch_p<K, S, TPB, DENSE, RL, (!S && TPB == 1024) ? 32 : 16>(dev, a, blocks, threads, stream);
    return launch_p<K, S, TPB, DENSE, RL, 16>(dev, a, blocks, threads, stream);
}
template <int K, bool S, int TPB, bool DENSE>
static hipError_t launch_pipe(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    return dev.rank_in_lds ? launch_rl<K, S, TPB, DENSE, true>(dev, a, blocks, threads, stream)
                           : launch_rl<K, S, TPB, DENSE, false>(dev, a, blocks, threads, stream);
}
template <int K, bool S, int TPB>
static hipError_t launch_tpb(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    return a.dense ? launch_pipe<K, S, TPB, true>(dev, a, blocks, threads, stream)
                   : launch_pipe<K, S, TPB, false>(dev, a, blocks, threads, stream);
}

// Three register budgets: 1024-thread workgroups (128 VGPRs, 4 waves/SIMD), 768 (168 VGPRs,
// 3 waves/SIMD) and <= 512 (256 VGPRs, 2 waves/SIMD) with one workgroup per CU.
template <int K, bool S>
static hipError_t launch_one(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    if (threads > 768) return launch_tpb<K, S, 1024>(dev, a, blocks, threads, stream);
    if (threads > 512) return launch_tpb<K, S, 768>(dev, a, blocks, threads, stream);
    return launch_tpb<K, S, 512>(dev, a, blocks, threads, stream);
}

hipError_t launch_gram_scan(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    if (dev.K == 3) return dev.has_short ? launch_one<3, true>(dev, a, blocks, threads, stream) : launch_one<3, false>(dev, a, blocks, threads, stream);
    return dev.has_short ? launch_one<2, true>(dev, a, blocks, threads, stream) : launch_one<2, false>(dev, a, blocks, threads, stream);
}

}  // namespace daac
