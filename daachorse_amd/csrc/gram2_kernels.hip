// GRAM engine kernels, second table set (gfx950): count (+ checksum) of the find_overlapping stream with ONE random
// LDS lookup per position.  See gram2.hpp for the tables and gram_kernels.hip for the method the two share (no state
// chain, coalesced 16-byte non-temporal haystack loads, B hits compacted by wave ballot into an LDS stack and taken 64
// at a time, walkers in per-wave HBM slabs).  What changed, and why (tools/micro/pipes_bench.hip on the MI355X):
//
//   * a wave-wide random LDS lookup costs 7.5 (b32) - 9 (u16) CU cycles, conflict or not, and the CU has one LDS pipe
//     against four SIMDs: lookups per position are what the step is priced in.  v1: class (u32), CID (u16), COMBO (b64),
//     B word (b32) = 3 random + 1 table read, two index chains (K-gram and (K+1)-gram).  Here: class (u8), M word (b32) —
//     continuation bits AND the number of short patterns in one word, one index chain; CID/H are read only when the
//     caller wants the checksum — which is what this kernel serves since round 4: `.count()` alone runs on gram3_kernels.hip (one
//     random lookup per position), and the count-only variants of this kernel, which AUTO never selected, are gone;
//   * the neighbour exchange goes through DPP wave shifts (VALU) instead of ds_bpermute (6.4 LDS cycles each);
//   * the byte address of the M word comes straight out of the mad chain: the class is scaled and the table base added
//     by one v_lshl_add, so neither lookup needs address arithmetic of its own.
//
// Roofline: HBM bytes of haystack (1 B read per byte); integer/bit work only, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef uint32_t g2_u32x4_t __attribute__((ext_vector_type(4)));
// settings of the round-2 A/B runs, frozen at the measured best
#define DAAC_G2_TAILS 1
#define DAAC_G2_DRAIN_W 2
#define G2_GROUP 8
#define G2_GROUPED_BALLOT 1
constexpr int kGroup2 = G2_GROUP;   // positions whose LDS reads are issued together
constexpr uint32_t kRing2 = 128;    // entries of a wave's hit stack in LDS (at most 63 left over + 64 new)
constexpr uint32_t kMaskBits = 0x3fffffffu;
typedef __attribute__((address_space(3))) const uint32_t lds2_cu32;
typedef __attribute__((address_space(3))) const uint16_t lds2_cu16;
typedef __attribute__((address_space(3))) const uint8_t lds2_cu8;

// Keeps a sum the way it is written: without it the compiler re-associates the three-term index sums into
// mul + mad + shift + add3 (four instructions instead of v_lshl_add + two v_mad_u32_u24).  Operands of the arithmetic
// itself stay visible to the compiler: a class that came out of ds_read_u8 is known to be zero-extended (as an inline-asm
// operand it was masked again with 0xff, one instruction per byte).
__device__ __forceinline__ uint32_t pin(uint32_t x) {
    asm("" : "+v"(x));
    return x;
}
// lane i <- lane i - 1 of `v`; lane 0 keeps `lane0` (DPP wave_shr:1, bound_ctrl off keeps the old destination)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t lane0) {
    uint32_t d = lane0;
    asm volatile("s_nop 1\nv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
// lane i <- lane i + 1 of `v`; lane 63 keeps `lane63`
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t lane63) {
    uint32_t d = lane63;
    asm volatile("s_nop 1\nv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}

__device__ __forceinline__ unsigned long long g2_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void g2_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ void g2_reduce(unsigned long long cnt, uint32_t s1, uint32_t s2, unsigned long long *scratch, unsigned long long *result) {
    const unsigned long long c = g2_wave_sum(cnt), x1 = g2_wave_sum(s1), x2 = g2_wave_sum(s2);
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

}  // namespace

// Count + checksum (CID/H staged and read).  K = context length; S16 = directory entries are u16;
// DENSE = queue the hits position by position without testing the group of four first
template <int K, bool S16, bool DENSE, int TPB>
__global__ __launch_bounds__(TPB) void gram2_kernel(const Gram2Dev g, const GramArgs a) {
    constexpr int P = 16;  // positions (bytes) a lane takes per step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t offM = g.off_m_exact, offS = g.off_s_exact;
    const uint32_t offRing = g.off_ring_exact;
    g2_copy(smem, g.cls, 256);
    g2_copy(smem + offM, g.m, g.m_bytes);
    g2_copy(smem + offS, g.sdir, g.s_bytes);
    g2_copy(smem + kGram2OffH, g.hsum, g.h_bytes);
    g2_copy(smem + g.off_cid, g.cid4, g.cid_bytes);
    __syncthreads();
    // tables are read through absolute LDS addresses (this kernel has no static LDS: the dynamic segment starts at 0)
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<lds2_cu8 *>(static_cast<uintptr_t>(byte)); };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<lds2_cu32 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C4 = g.C * 4u, CC4 = g.C * g.C * 4u;
    const uint32_t cid_bias = g.off_cid - (offM >> 1);  // CID entry of the K-gram whose M word is at `am`: (am >> 1) + cid_bias
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    // this wave's slab of pending walkers: {low 32 bits of the virtual position p of the last byte of a (K+1)-gram,
    // the depth-(K+2) state reached on the byte at p + 1 | class of the byte at p + 2 << 27}; all entries of a slab share
    // the bits of p above 2^32 (slab_hi): the slab is emptied before they change
    uint2 *__restrict__ slab = a.wq + (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg) * a.wq_slab;
    uint32_t wq_n = 0, slab_hi = 0;  // wave-uniform

    unsigned long long tot_cnt = 0;
    uint32_t cnt32 = 0;  // matches of the current region (a region is far too short to overflow 32 bits)
    uint32_t tot_s1 = 0, tot_s2 = 0;

    auto load_chunk = [&](uint64_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const g2_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const g2_u32x4_t *>(hay + v));
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
    auto class_at = [&](uint64_t p) -> uint32_t { return (p >= a.lead && p < a.vlen) ? cls_of(hay[p]) : 0u; };

    // Finishes the queued branches, 64 per round (gram_kernels.hip: drain).  (A version in which a lane whose branch has ended
    // takes the next entry at once, one record per lane and turn, was no faster on word-soup text and 11 % slower on uniform
    // text: the drain is not where the time goes, profiles/r02_emit_experiments.txt.)
    auto drain = [&]() {
        const uint4 *__restrict__ recs = g.drec;
        // W branches per lane at a time, each taken through its first record and first step (where, with the single paths folded
        // into tail records, nearly every branch ends); what is left of a branch walks on by itself.  (W = 1, 2, 4, 6 measured
        // the same: the drain is not bound by the latency of a round.)
        constexpr int W = DAAC_G2_DRAIN_W;
        for (uint32_t base = 0; base < wq_n; base += 64u * W) {
            uint4 r[W];
            uint64_t vnext[W];
            uint32_t kn[W];
            unsigned long long ahead[W];
            bool go[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t i = base + 64u * w + lane;
                uint2 e = uint2{0u, 0u};
                if (i < wq_n) e = slab[i];
                vnext[w] = ((static_cast<uint64_t>(slab_hi) << 32) | e.x) + 2;  // the state consumed the byte before vnext
                kn[w] = e.y >> 27;
                r[w] = uint4{0u, 0u, 0u, 0u};  // (an idle slot: counts nothing, leads nowhere)
                if (i < wq_n) r[w] = recs[e.y & 0x07ffffffu];  // {cmap, first_child, own_cnt, own_hsum}
            }
            auto read_ahead = [&](uint64_t v) -> unsigned long long {
                unsigned long long x;
                if (v >= a.lead && v + 8 <= a.vlen) {
                    __builtin_memcpy(&x, hay + v, 8);
                } else {
                    x = 0;
                    for (int b = 7; b >= 0; --b) x = (x << 8) | ((v + b >= a.lead && v + b < a.vlen) ? hay[v + b] : g.unused_byte);
                }
                return x;
            };
            // the first step of each: the walker stands on a state reached by the byte before vnext, `kn` is the class of the byte AT
            // vnext.  Only a branch that goes on asks for more — its next record and, with it, the eight bytes from vnext on (from
            // HBM: the scan's own reads of the text are non-temporal); asking for all of them regardless cost 9 % on uniform text.
#pragma unroll
            for (int w = 0; w < W; ++w) {
                cnt32 += r[w].z;
                tot_s1 += r[w].w;
                tot_s2 += r[w].w * static_cast<uint32_t>(vnext[w] - a.lead);
                go[w] = ((r[w].x >> kn[w]) & 1u) != 0;
                ahead[w] = 0;
                if (go[w]) {
                    r[w] = recs[r[w].y + __popc(r[w].x & ((1u << kn[w]) - 1u))];
                    ++vnext[w];
                    ahead[w] = read_ahead(vnext[w]);
                }
            }
            // ... and whatever is left of each
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (!go[w]) continue;
                uint4 rr = r[w];
                uint64_t vn = vnext[w];
                unsigned long long ah = ahead[w];
                uint32_t n_ahead = 8, k = 0;
                for (;;) {
                    if (rr.y >> 31) {
                        // a tail record (gram2.cpp, round 6): one path of rr.y & 15 edges with one pattern end, at its node (rr.y >> 4) & 15;
                        // {h of that pattern, -, path bytes} against the next eight text bytes in one step
                        if (n_ahead < 8u) ah = read_ahead(vn);
                        const unsigned long long diff = ((static_cast<unsigned long long>(rr.w) << 32) | rr.z) ^ ah;
                        const uint32_t edges = rr.y & 15u, at = (rr.y >> 4) & 15u;
                        uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
                        same = same < edges ? same : edges;
                        if (at <= same) {
                            cnt32 += 1u;
                            tot_s1 += rr.x;
                            tot_s2 += rr.x * static_cast<uint32_t>(vn + at - a.lead);
                        }
                        break;
                    }
                    k = cls_of(static_cast<uint32_t>(ah) & 0xffu);
                    cnt32 += rr.z;
                    tot_s1 += rr.w;
                    tot_s2 += rr.w * static_cast<uint32_t>(vn - a.lead);
                    if (((rr.x >> k) & 1u) == 0) break;
                    rr = recs[rr.y + __popc(rr.x & ((1u << k) - 1u))];
                    ++vn;
                    if (n_ahead <= 1u) { ah = read_ahead(vn); n_ahead = 8; }
                    else { ah >>= 8; --n_ahead; }
                }
            }
        }
        wq_n = 0;
    };

    // ---- the hit ring: entry = {LDS address of the M word of the K-gram before the hit (17 bits) | class of the hit byte
    // << 17 | classes of the next two bytes << 22 / << 27, low 32 bits of the position of the hit byte}
    uint2 *ring = reinterpret_cast<uint2 *>(smem + offRing) + wave_in_wg * kRing2;
    uint32_t q_n = 0;             // wave-uniform
    uint4 pend = uint4{0u, 0u, 0u, 0u};  // record read for the previous batch, not yet consumed
    uint32_t pend_item = 0, pend_pos = 0, pend_rank = 0;
    bool pend_valid = false;      // wave-uniform
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;     // {cmap, own h32 sum, first_child, -}; zero for idle lanes
        cnt32 += r.y != 0;
        tot_s1 += r.y;
        tot_s2 += r.y * (pend_pos - a.lead + 1u);  // end = position - lead + 1 (mod 2^32)
        const uint32_t k1 = (pend_item >> 22) & 31u;  // (a class >= 1 where there is an edge to follow; class 0: bit 0 is not an edge)
        const bool go = k1 != 0 && ((r.x >> k1) & 1u);
        const unsigned long long m = __ballot(go);
        if (m != 0) {  // the branch goes on past depth K+1 -> queue a walker
            if (go)
                (slab + wq_n)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0))] =
                    uint2{pend_pos, (r.z + __popc(r.x & ((1u << k1) - 2u))) | ((pend_item >> 27) << 27)};
            wq_n += __popcll(m);
        }
    };
    // rank of continuation bit d of the M word at LDS address `am` among all set bits = offset of the depth-(K+1) state
    auto deep_rank = [&](uint32_t am, uint32_t d) -> uint32_t {
        const uint32_t rel = am - offM;
        // one 16-byte read of the group (left to itself the compiler splits it into b96 + b32: two LDS instructions)
        g2_u32x4_t q;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(offM + (rel & ~15u)) : "memory");
        const uint32_t idx = (rel >> 2) & 3u;
        const uint32_t base = S16 ? *reinterpret_cast<const uint16_t *>(smem + offS + ((rel >> 4) << 1))
                                  : *reinterpret_cast<const uint32_t *>(smem + offS + ((rel >> 4) << 2));
        const uint32_t own = idx == 0 ? q.x : idx == 1 ? q.y : idx == 2 ? q.z : q.w;
        uint32_t below = __popc(own & kMaskBits & ((1u << d) - 1u));
        below += idx > 0 ? __popc(q.x & kMaskBits) : 0u;
        below += idx > 1 ? __popc(q.y & kMaskBits) : 0u;
        below += idx > 2 ? __popc(q.z & kMaskBits) : 0u;
        return base + below;
    };
    auto process_batch = [&]() {
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
            pend_rank = deep_rank(it.x & 0x1ffffu, (it.x >> 17) & 31u);
            pend = g.dhit4[pend_rank];
        }
        pend_valid = true;
    };

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

        constexpr uint64_t SB = 64ull * P;  // bytes a wave takes per step
        uint4 pf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) pf[i] = (rbase + SB * i < rend) ? load_chunk(rbase + SB * i + lane * P) : uint4{ub4, ub4, ub4, ub4};

        for (uint64_t sb = rbase; sb < rend; sb += SB) {
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint64_t v = sb + lane * P;
            const uint32_t v32 = static_cast<uint32_t>(v);
            const uint4 cur = pf[0];
            pf[0] = pf[1];
            consume_pending();  // before the next chunk is requested: loads retire in order
            __builtin_amdgcn_s_setprio(0);
            pf[1] = (sb + SB * 2 < rend) ? load_chunk(v + SB * 2) : uint4{ub4, ub4, ub4, ub4};

            // the two bytes after this wave's share (lane 63 needs their classes)
            uint32_t after2;
            if (sb + SB < rend) {
                after2 = __builtin_amdgcn_readfirstlane(pf[0].x);
            } else {
                const uint64_t p0 = sb + SB, p1 = sb + SB + 1;
                after2 = ((p0 >= a.lead && p0 < a.vlen) ? hay[p0] : g.unused_byte) | (((p1 >= a.lead && p1 < a.vlen) ? hay[p1] : g.unused_byte) << 8);
            }

            // ---- byte classes of this lane's P positions plus K to the left and 2 to the right ----
            uint32_t kx[K + P + 2];
            {
                const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                // pinned: with several users, some of which shift the upper bits out, the compiler loads the byte "any-extended"
                // and masks it again for every other user (one v_and per byte)
                for (int b = 0; b < P; ++b) {
                    kx[K + b] = pin(cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu));
                    __builtin_assume(kx[K + b] < 32u);  // (no v_and 0xffffff in front of the 24-bit mads)
                }
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);  // this lane's last K classes, oldest low
            uint32_t left;
            if (g.xlane_dpp) {
                left = wave_shr1(pk, carry);
            } else {
                left = __shfl_up(pk, 1, 64);
                if (lane == 0) left = carry;
            }
            carry = __builtin_amdgcn_readlane(pk, 63);
#pragma unroll
            for (int i = 0; i < K; ++i) { kx[i] = (left >> (8 * i)) & 0xffu; __builtin_assume(kx[i] < 32u); }
            uint32_t right63 = cls_of(after2 & 0xffu) | (cls_of((after2 >> 8) & 0xffu) << 5);  // same address in every lane: a broadcast
            asm volatile("" : "+v"(right63));  // keep the two reads out of a lane-63-only branch: the step stays one basic block
            uint32_t right;                    // the two classes after this lane's P
            if (g.xlane_dpp) {
                right = wave_shl1(kx[K] | (kx[K + 1] << 5), right63);
            } else {
                right = __shfl_down(kx[K] | (kx[K + 1] << 5), 1, 64);
                right = lane == 63 ? right63 : right;
            }
            kx[K + P] = right & 31u;
            kx[K + P + 1] = right >> 5;

            // ---- LDS address of the M word of the K-gram ending at j, j = -1 .. P-1: offM + 4 * (K-gram) straight out of the
            // mad chain (the class of the last byte is scaled and the table base added by one v_lshl_add) ----
            uint32_t am[P + 1];
#pragma unroll
            for (int j = -1; j < P; ++j) {
                uint32_t x = pin((kx[K + j] << 2) + offM);                        // 4 c_j + offM              (v_lshl_add_u32)
                x = __umul24(kx[K + j - 1], C4) + x;                              // + 4 C c_(j-1)             (v_mad_u32_u24)
                if (K == 3) x = __umul24(kx[K + j - 2], CC4) + pin(x);            // + 4 C^2 c_(j-2)           (v_mad_u32_u24)
                am[j + 1] = x;
            }
            // the classes of the bytes at j, j+1, j+2 as the queue entry wants them (5 bits each from bit 17 up), rolling from
            // the right; whatever is above the third class is shifted out when the entry is put together
            uint32_t tri[P];
            tri[P - 1] = (((kx[K + P + 1] << 5) | kx[K + P]) << 5) | kx[K + P - 1];
#pragma unroll
            for (int j = P - 2; j >= 0; --j) tri[j] = (tri[j + 1] << 5) | kx[K + j];

            uint32_t ccnt = 0, A = 0, T = 0;
            const uint32_t e0 = static_cast<uint32_t>(v - a.lead) + 1u;  // end of this lane's position 0
            uint32_t mprev = lds_u32(am[0]);  // M word of the K-gram ending just before this lane's share
#pragma unroll
            for (int grp = 0; grp < P / kGroup2; ++grp) {
                uint32_t mw[kGroup2], mb[kGroup2], id4[kGroup2], hs[kGroup2];
#pragma unroll
                for (int jj = 0; jj < kGroup2; ++jj) {
                    const int j = grp * kGroup2 + jj;
                    mw[jj] = lds_u32(am[j + 1]);
                    id4[jj] = *reinterpret_cast<lds2_cu16 *>(static_cast<uintptr_t>((am[j + 1] >> 1) + cid_bias));
                }
#pragma unroll
                for (int jj = 0; jj < kGroup2; ++jj) hs[jj] = lds_u32(id4[jj]);  // H sits at a fixed offset that the ids include
#pragma unroll
                for (int jj = 0; jj < kGroup2; ++jj) {
                    mb[jj] = jj == 0 ? mprev : mw[jj - 1];  // the word whose continuation bits the byte at j is tested against
                    ccnt += mw[jj] >> 30;
                    A += hs[jj];
                    T += A;
                }
                mprev = mw[kGroup2 - 1];
                auto queue_hit = [&](int jj, bool hit) {
                    const int j = grp * kGroup2 + jj;
                    const unsigned long long m = __ballot(hit);
                    if (m != 0) {  // wave-uniform
                        const uint32_t q_s = q_n;
                        if (hit)
                            (ring + q_s)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] =
                                uint2{(tri[j] << 17) | am[j], v32 + j};
                        q_n = q_s + static_cast<uint32_t>(__popcll(m));
                        if (q_n >= 64u) process_batch();
                    }
                };
                if (DENSE && G2_GROUPED_BALLOT) {
                    // the four ballots first, then one round of scalar bookkeeping for the group: a ballot is a VALU result
                    // read by the scalar unit, and waiting for it once per position column left both pipes idle
                    bool hit[kGroup2];
                    unsigned long long bm[kGroup2];
                    uint32_t tot = 0;
#pragma unroll
                    for (int jj = 0; jj < kGroup2; ++jj) {
                        hit[jj] = __builtin_amdgcn_ubfe(mb[jj], kx[K + grp * kGroup2 + jj], 1) != 0;
                        bm[jj] = __ballot(hit[jj]);
                        tot += static_cast<uint32_t>(__popcll(bm[jj]));
                    }
                    if (q_n + tot <= kRing2) {  // the stack takes the whole group (always, unless nearly every lane hits)
                        uint32_t q_s = q_n;
#pragma unroll
                        for (int jj = 0; jj < kGroup2; ++jj) {
                            const int j = grp * kGroup2 + jj;
                            if (hit[jj])
                                (ring + q_s)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(bm[jj] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bm[jj]), 0u))] =
                                    uint2{(tri[j] << 17) | am[j], v32 + j};
                            q_s += static_cast<uint32_t>(__popcll(bm[jj]));
                        }
                        q_n = q_s;
                        if (q_n >= 64u) process_batch();
                        if (q_n >= 64u) process_batch();
                    } else {
#pragma unroll
                        for (int jj = 0; jj < kGroup2; ++jj) queue_hit(jj, hit[jj]);
                    }
                } else if (DENSE) {
#pragma unroll
                    for (int jj = 0; jj < kGroup2; ++jj) queue_hit(jj, __builtin_amdgcn_ubfe(mb[jj], kx[K + grp * kGroup2 + jj], 1) != 0);
                } else {
                    uint32_t hits = 0;
#pragma unroll
                    for (int jj = 0; jj < kGroup2; ++jj) hits |= __builtin_amdgcn_ubfe(mb[jj], kx[K + grp * kGroup2 + jj], 1) << jj;
                    if (__any(hits != 0)) {
#pragma unroll
                        for (int jj = 0; jj < kGroup2; ++jj) queue_hit(jj, (hits >> jj) & 1u);
                    }
                }
            }
            cnt32 += ccnt;
            tot_s1 += A;
            tot_s2 += A * (e0 + static_cast<uint32_t>(P)) - T;  // sum_j hs_j * (e0 + j)
        }
        tot_cnt += cnt32;  // per region: 32 bits cannot overflow within one
        cnt32 = 0;
      }
      while (q_n != 0) process_batch();
      consume_pending();
      drain();
      tot_cnt += cnt32;
      cnt32 = 0;
    }
    g2_reduce(tot_cnt, tot_s1, tot_s2, reinterpret_cast<unsigned long long *>(smem), a.result);
}

template <int K, bool S16, bool DENSE>
static hipError_t launch2_tpb(const Gram2Dev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t stream) {
    // two register budgets: 1024-thread workgroups (128 VGPRs) and <= 512 (256 VGPRs)
    if (threads > 512) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram2_kernel<K, S16, DENSE, 1024>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gram2_kernel<K, S16, DENSE, 1024>), dim3(blocks), dim3(threads), lds, stream, dev, a);
    } else {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram2_kernel<K, S16, DENSE, 512>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gram2_kernel<K, S16, DENSE, 512>), dim3(blocks), dim3(threads), lds, stream, dev, a);
    }
    return hipGetLastError();
}
template <int K>
static hipError_t launch2_k(const Gram2Dev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t stream) {
    if (dev.s16)
        return a.dense ? launch2_tpb<K, true, true>(dev, a, blocks, threads, lds, stream) : launch2_tpb<K, true, false>(dev, a, blocks, threads, lds, stream);
    return a.dense ? launch2_tpb<K, false, true>(dev, a, blocks, threads, lds, stream) : launch2_tpb<K, false, false>(dev, a, blocks, threads, lds, stream);
}

uint32_t gram2_lds_bytes(const Gram2Dev &dev, bool) { return dev.lds_exact; }

// count + checksum over the second table set (`.count()` alone: launch_gram3_scan); needs dev.exact_ok
hipError_t launch_gram2_scan(const Gram2Dev &dev, const GramArgs &a, bool, uint32_t blocks, uint32_t threads, hipStream_t stream) {
    if (!dev.exact_ok) return hipErrorInvalidValue;
    const uint32_t lds = dev.lds_exact;
    return dev.K == 3 ? launch2_k<3>(dev, a, blocks, threads, lds, stream) : launch2_k<2>(dev, a, blocks, threads, lds, stream);
}

}  // namespace daac
