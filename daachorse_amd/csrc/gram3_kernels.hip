// GRAM engine, `.count()` kernel with lane-local hit masks (gfx950).  Tables: gram2.hpp (M words, rank directory, hit and walk
// records); method: gram_kernels.hip (no state chain; an occurrence longer than K is found from its start: continuation bit of
// the (K+1)-gram, rank, one 8-byte record from L2, queued goto-only walk for the few branches that go on).
//
// What this kernel changes against gram2_kernels.hip, and why (profiles/r02_rocprofv3_pmc_sq.txt: 20.4 VALU instructions per
// haystack byte, of which ~8 were the producer side of the hit queue — ballot, two mbcnt, entry, address, store for EVERY
// position column, because with an 11 % hit rate some lane of 64 hits in every column — and one more the rolling class
// triples every queue entry carried):
//
//   * the main path only collects a per-lane HIT MASK (v_bfe + v_lshl_or per position); nothing else is kept per position;
//   * after the P positions of a step the set bits are queued "iteration-major": turn t takes the t-th set bit of every lane
//     that has one (one compare = the ballot, two mbcnt, ffbl, clear, add, store), 5-6 turns per 16 positions instead of 16
//     columns;
//   * a queue entry is just the LDS address of the hit byte in the wave's TEXT SLOT: the step's kilobyte of raw haystack is
//     written to LDS as it arrives (one ds_write_b128 per lane and chunk; slots are self-contained: the last four bytes of the
//     previous step in front, the first sixteen of the next behind), and the consumer — which runs 64 hits wide, all lanes busy
//     — re-derives the context, the hit class and the two classes behind from there;
//   * the short-pattern counts roll through one register (v_alignbit, two bits per position) and are summed by popcounts once
//     per 16 positions;
//   * TAIL: single paths of the trie are folded into one record FROM THE HIT RECORD ON (gram2.hpp: dhit_t / drec_t): two thirds
//     of the depth-(K+1) states of a word list have one path below them, and such a hit is settled by one 16-byte record and
//     one compare with the eight text bytes behind it (taken from the slot) — no walker, no second and third record.  Branches
//     that do go on take their text with them into the slab.  For text made of dictionary words, where walkers were half of
//     the kernel's time (profiles/r03_gram3_decomposition.txt).
//
// Roofline: HBM bytes of haystack (1 B read per byte); integer/bit work only, no MFMA.
// (The timing-only builds that cut one stage out to price it — profiles/r03_gram3_decomposition.txt — are a patch on top of this file:
// tools/variants/gram3_decomposition.patch, applied by tools/mkvar2.sh.  Nothing of them is in the shipped source.)
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef uint32_t g3_u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t g3_u32x2_t __attribute__((ext_vector_type(2)));
constexpr uint32_t kRing3 = 128;     // entries of a wave's hit queue (FIFO; at most 63 left over + 64 new)
constexpr uint32_t kOffM3 = 256;     // LDS offset of M (classes at 0)
constexpr uint32_t kProbePercent = 3;  // density probe: TAIL when more than 3 % of the sampled positions start a walker
typedef __attribute__((address_space(3))) const uint32_t lds3_cu32;
typedef __attribute__((address_space(3))) uint32_t lds3_u32;
typedef __attribute__((address_space(3))) uint16_t lds3_u16;
typedef __attribute__((address_space(3))) const uint16_t lds3_cu16;
typedef __attribute__((address_space(3))) const uint8_t lds3_cu8;
typedef __attribute__((address_space(3))) g3_u32x4_t lds3_u32x4;

__device__ __forceinline__ uint32_t pin3(uint32_t x) {
    asm("" : "+v"(x));
    return x;
}
// lane i <- lane i - 1 of `v`; lane 0 keeps `lane0`
__device__ __forceinline__ uint32_t wave_shr1_3(uint32_t v, uint32_t lane0) {
    uint32_t d = lane0;
    asm volatile("s_nop 1\nv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ unsigned long long g3_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void g3_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ void g3_reduce(unsigned long long cnt, unsigned long long *scratch, unsigned long long *result) {
    const unsigned long long c = g3_wave_sum(cnt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) scratch[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r0 = 0;
        for (int w = 0; w < static_cast<int>((blockDim.x + 63) >> 6); ++w) r0 += scratch[w];
        if (r0) atomicAdd(result, r0);
    }
}

}  // namespace

// K = context length; Q = 16-byte chunks a lane takes per step (P = 16 Q positions); RFULL = one directory entry per M word
// (else one per four words, u16 or u32 by S16); TAIL = tail records from the hit record on, walker slab entries are 16 bytes and carry their text
template <int K, int Q, bool S16, bool RFULL, bool TAIL>
__device__ __forceinline__ void gram3_body(const Gram2Dev &g, const GramArgs &a, const Gram3Lds &L, char *smem) {
    constexpr int P = 16 * Q;
    constexpr uint32_t SB = 64u * P;          // bytes a wave takes per step
    constexpr uint32_t SLOT = SB + 32u;       // [12,16) the four bytes before the step | [16, 16 + SB) the step | 16 bytes of the next
    const uint32_t offM = kOffM3, offS = L.off_s;
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<lds3_cu8 *>(static_cast<uintptr_t>(byte)); };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<lds3_cu32 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C4 = g.C * 4u, CC4 = g.C * g.C * 4u;
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    // per-wave LDS: two text slots, then the hit queue
    const uint32_t tb = L.off_wave + wave_in_wg * L.wave_stride;   // wave-uniform
    const uint32_t ringb = tb + 2u * SLOT;
    // this wave's slab of pending walkers (gram2_kernels.hip); TAIL: 16-byte entries {pos, state, the seven text bytes from pos + 2 on}
    const uint64_t slab_index = (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg) * a.wq_slab;
    uint2 *__restrict__ slab2 = a.wq + (TAIL ? 2 * slab_index : slab_index);
    uint4 *__restrict__ slab4 = reinterpret_cast<uint4 *>(slab2);
    uint32_t wq_n = 0, slab_hi = 0;  // wave-uniform

    unsigned long long tot_cnt = 0;
    uint32_t cnt32 = 0;  // matches of the current region (a region is far too short to overflow 32 bits)

    auto load_chunk = [&](uint64_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const g3_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const g3_u32x4_t *>(hay + v));
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
    auto raw_at = [&](uint64_t p) -> uint32_t { return (p >= a.lead && p < a.vlen) ? hay[p] : g.unused_byte; };

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
    // the rest of a subtree that is one path {1 << 31 | edges | word ends << 4, -, path bytes 0-3, path bytes 4-7} against the text
    auto tail_count = [&](const uint4 &rr, unsigned long long text) -> uint32_t {
        const uint32_t edges = rr.x & 15u;
        const unsigned long long path = (static_cast<unsigned long long>(rr.w) << 32) | rr.z;
        const unsigned long long diff = path ^ text;
        uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
        same = same < edges ? same : edges;
        return __popc((rr.x >> 4) & ((2u << same) - 1u) & 0x1ffu);
    };
    // Finishes the queued branches, 64 per round (gram2_kernels.hip: drain)
    auto drain = [&]() {
        const uint4 *__restrict__ recs = TAIL ? g.drec_t : g.drec_c;
        constexpr int W = 2;
        for (uint32_t base = 0; base < wq_n; base += 64u * W) {
            uint4 r[W];
            uint64_t vnext[W];
            uint32_t kn[W], n_ah[W] = {0u, 0u};
            unsigned long long ahead[W];
            bool go[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t i = base + 64u * w + lane;
                uint2 e = uint2{0u, 0u};
                ahead[w] = 0;
                if (i < wq_n) {
                    if (TAIL) {
                        const uint4 e4 = slab4[i];
                        e = uint2{e4.x, e4.y};
                        ahead[w] = (static_cast<unsigned long long>(e4.w & 0xffffffu) << 32) | e4.z;  // the bytes from vnext on (e4.w >> 24 of them)
                        n_ah[w] = e4.w >> 24;
                    } else {
                        e = slab2[i];
                    }
                }
                vnext[w] = ((static_cast<uint64_t>(slab_hi) << 32) | e.x) + 2;  // the state consumed the byte before vnext
                kn[w] = e.y >> 27;
                if (!TAIL) n_ah[w] = 0u;
                r[w] = uint4{0u, 0u, 0u, 0u};  // (an idle slot: counts nothing, leads nowhere)
                if (i < wq_n) r[w] = recs[e.y & 0x07ffffffu];  // {cmap, first_child, own_cnt, -} or a tail record
            }
            // the first step of each: the walker stands on a state reached by the byte before vnext, `kn` is the class of the byte
            // AT vnext.  Only a branch that goes on asks for more.
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (r[w].x >> 31) {  // a tail record (TAIL: its path is no longer than the bytes at hand unless it has eight edges)
                    if (n_ah[w] < (r[w].x & 15u)) ahead[w] = read_ahead(vnext[w]);
                    cnt32 += tail_count(r[w], ahead[w]);
                    go[w] = false;
                    continue;
                }
                cnt32 += r[w].z;
                go[w] = ((r[w].x >> kn[w]) & 1u) != 0;
                if (go[w]) {
                    r[w] = recs[r[w].y + __popc(r[w].x & ((1u << kn[w]) - 1u))];
                    ++vnext[w];
                    if (TAIL) { ahead[w] >>= 8; n_ah[w] -= 1u; }
                    else { ahead[w] = read_ahead(vnext[w]); n_ah[w] = 8; }
                }
            }
            // ... and whatever is left of each
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (!go[w]) continue;
                uint4 rr = r[w];
                uint64_t vn = vnext[w];
                unsigned long long ah = ahead[w];
                uint32_t n_ahead = n_ah[w], k = 0;
                for (;;) {
                    if (rr.x >> 31) {
                        if (n_ahead < (rr.x & 15u)) { ah = read_ahead(vn); n_ahead = 8; }
                        cnt32 += tail_count(rr, ah);
                        break;
                    }
                    if (n_ahead == 0u) { ah = read_ahead(vn); n_ahead = 8; }
                    k = cls_of(static_cast<uint32_t>(ah) & 0xffu);
                    cnt32 += rr.z;
                    if (((rr.x >> k) & 1u) == 0) break;
                    rr = recs[rr.y + __popc(rr.x & ((1u << k) - 1u))];
                    ++vn;
                    ah >>= 8;
                    --n_ahead;
                }
            }
        }
        wq_n = 0;
    };

    // ---- the hit queue: entry = LDS address of the hit byte in one of the wave's two text slots
    // (measured and dropped, profiles/r03_gram3_ab.txt: two or three batches taken side by side so that they share the chain of
    // dependent LDS round trips, +1 % / -6 %; two batches of records in flight instead of one, -2 %; the text request at the
    // end of the step instead of its top, -1 %: what a batch costs is its instructions and its walkers, not its waiting)
    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running; entries live at (index & (kRing3 - 1))
    uint32_t posbias0 = 0, posbias1 = 0;  // per slot: (low 32 bits of the virtual position of a byte) - (its LDS address)
    uint4 pend = uint4{0u, 0u, 0u, 0u};  // record read for the previous batch, not yet consumed; zero for idle lanes
                                         // {cmap | ends-a-pattern, first_child, -, -}; TAIL: or a tail record
    uint32_t pend_pos = 0, pend_k = 0; // position of the hit byte; classes of the two bytes behind it (k1 | k2 << 8)
    uint32_t pend_t0 = 0, pend_t1 = 0; // TAIL: the eight bytes from position + 1 on
    bool pend_valid = false;           // wave-uniform
    // A branch that goes on past the hit's state asks for the record of the NEXT state at once (p2, looked at one batch later) instead
    // of going through the slab: most branches end there — with TAIL nine in ten such records are tail records and settle the branch with
    // one compare — and only what still goes on is queued for the drain.  (On word soup the slab round trip of every walker was 2 GB
    // written and read back per 4 GiB: profiles/r03_hbm_traffic.json.)
    uint4 p2 = uint4{0u, 0u, 0u, 0u};    // record of the state below the hit's; zero for idle lanes
    uint32_t p2_pos = 0, p2_state = 0;   // position of the hit byte; the state asked for | class of the byte at position + 2 << 27
    uint32_t p2_t0 = 0, p2_t1 = 0;       // the seven bytes from position + 2 on
    bool p2_live = false, p2_any = false;        // per lane / wave-uniform
    auto push_walker = [&](bool go, uint32_t pos, uint32_t st_k, uint32_t t0, uint32_t t1n) {
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go) {
                const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                if (TAIL) slab4[at] = uint4{pos, st_k, t0, t1n};   // t1n: text bytes 4-6 | number of valid text bytes << 24
                else slab2[at] = uint2{pos, st_k};
            }
            wq_n += __popcll(m);
        }
    };
    auto finish_second = [&]() {
        if (!TAIL || !p2_any) return;
        p2_any = false;
        const uint4 r = p2;
        bool again = false;           // still going on: to the slab (or a tail of eight edges: the drain has the eighth byte fetched)
        uint32_t st_k = 0, pos = p2_pos, t0 = p2_t0, t1n = (p2_t1 & 0xffffffu) | (7u << 24);
        if (p2_live) {
            if (TAIL && (r.x >> 31)) {
                if ((r.x & 15u) <= 7u) cnt32 += tail_count(r, (static_cast<unsigned long long>(p2_t1) << 32) | p2_t0);
                else { again = true; st_k = p2_state; }  // eight edges, seven bytes at hand: the drain looks at the record again with the eighth byte fetched
            } else {
                cnt32 += r.z;
                const uint32_t k = p2_state >> 27;
                if ((r.x >> k) & 1u) {   // (class 0: bit 0 of a walk record's child map is never set)
                    again = true;
                    const uint32_t k3 = cls_of((p2_t0 >> 8) & 0xffu);
                    st_k = (r.y + __popc(r.x & ((1u << k) - 1u))) | (k3 << 27);
                    pos = p2_pos + 1u;
                    if (TAIL) {
                        t0 = __builtin_amdgcn_alignbyte(p2_t1, p2_t0, 1u);
                        t1n = ((p2_t1 >> 8) & 0xffffu) | (6u << 24);
                    }
                }
            }
        }
        p2_live = false;
        push_walker(again, pos, st_k, t0, t1n);
    };
    auto consume_pending = [&]() {
        finish_second();
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;
        bool go;
        const uint32_t k1 = pend_k & 0xffu;  // (class 0: bit 0 is not an edge)
        if (TAIL && (r.x >> 31)) {  // one path below the hit: settled here
            cnt32 += tail_count(r, (static_cast<unsigned long long>(pend_t1) << 32) | pend_t0);
            go = false;
        } else {
            cnt32 += r.x & 1u;
            go = k1 != 0 && ((r.x >> k1) & 1u);
        }
        const uint32_t child = r.y + __popc(r.x & ((1u << k1) - 2u));
        if (!TAIL) {
            // (without tail records the second stage does not pay: asked for at once the next record cost uniform text 5 %,
            // profiles/r03_gram3_ab.txt — there one hit in seventeen goes on at all)
            push_walker(go, pend_pos, child | (((pend_k >> 8) & 0xffu) << 27), 0u, 0u);
            return;
        }
        p2 = uint4{0u, 0u, 0u, 0u};
        p2_live = go;
        p2_any = __ballot(go) != 0;
        if (go) {
            p2 = g.drec_t[child];
            p2_pos = pend_pos;
            p2_state = child | (((pend_k >> 8) & 0xffu) << 27);
            p2_t0 = __builtin_amdgcn_alignbyte(pend_t1, pend_t0, 1u);  // text from position + 2 on
            p2_t1 = pend_t1 >> 8;
        }
    };
    auto process_batch = [&](uint32_t n) {  // n <= 64 entries from the head of the queue
        __builtin_amdgcn_s_setprio(2);
        consume_pending();
        pend = uint4{0u, 0u, 0u, 0u};
        pend_k = 0;
        if (lane < n) {
            const uint32_t e = lds_u32(ringb + (((q_head + lane) & (kRing3 - 1u)) << 2));
            pend_pos = e + ((e - tb) >= SLOT ? posbias1 : posbias0);
            const uint32_t t3 = e - 3u;
            const uint32_t a0 = t3 & ~3u, sh = t3 & 3u;
            // the dwords around the hit byte (slots are self-contained: never outside [slot + 12, slot + SLOT))
            const uint32_t d0 = lds_u32(a0), d1 = lds_u32(a0 + 4u), d2 = lds_u32(a0 + 8u);
            const uint32_t x_lo = __builtin_amdgcn_alignbyte(d1, d0, sh);  // bytes p-3 .. p
            const uint32_t x_hi = __builtin_amdgcn_alignbyte(d2, d1, sh);  // bytes p+1 .. p+4
            if (TAIL) {
                const uint32_t d3 = lds_u32(a0 + 12u);
                pend_t0 = x_hi;
                pend_t1 = __builtin_amdgcn_alignbyte(d3, d2, sh);          // bytes p+5 .. p+8
            }
            const uint32_t c0 = cls_of(x_lo & 0xffu), c1 = cls_of((x_lo >> 8) & 0xffu), c2 = cls_of((x_lo >> 16) & 0xffu), d = cls_of(x_lo >> 24);
            const uint32_t k1 = cls_of(x_hi & 0xffu), k2 = cls_of((x_hi >> 8) & 0xffu);
            pend_k = k1 | (k2 << 8);
            uint32_t am = (c2 << 2) + offM;
            am = __umul24(c1, C4) + am;
            if (K == 3) am = __umul24(c0, CC4) + am; else (void)c0;
            // rank of continuation bit d of that M word among all set bits = offset of the depth-(K+1) state
            uint32_t rank;
            if (RFULL) {
                const uint32_t own = lds_u32(am);
                const uint32_t base = *reinterpret_cast<lds3_cu16 *>(static_cast<uintptr_t>(offS + ((am - offM) >> 1)));
                rank = base + __popc(own & ((1u << d) - 1u));
            } else {
                const uint32_t rel = am - offM, grp = offM + (rel & ~15u);
                const uint32_t own = lds_u32(am), qx = lds_u32(grp), qy = lds_u32(grp + 4u), qz = lds_u32(grp + 8u);
                const uint32_t idx = (rel >> 2) & 3u;
                const uint32_t base = S16 ? *reinterpret_cast<lds3_cu16 *>(static_cast<uintptr_t>(offS + ((rel >> 4) << 1)))
                                          : *reinterpret_cast<lds3_cu32 *>(static_cast<uintptr_t>(offS + ((rel >> 4) << 2)));
                uint32_t below = __popc(own & ((1u << d) - 1u));
                below += idx > 0 ? __popc(qx & 0x3fffffffu) : 0u;
                below += idx > 1 ? __popc(qy & 0x3fffffffu) : 0u;
                below += idx > 2 ? __popc(qz & 0x3fffffffu) : 0u;
                rank = base + below;
            }
            if (TAIL) {
                pend = g.dhit_t[rank];
            } else {
                const uint2 h = g.dhit_c[rank];
                pend = uint4{h.x, h.y, 0u, 0u};
            }
        }
        q_head += n;
        pend_valid = true;
    };

    uint32_t sl = 0;          // slot of the current step (wave-uniform)
    uint32_t carry_in = 0;    // queued entries that belong to the step before the current one
    uint64_t region = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    while (region < a.nregions) {
      slab_hi = static_cast<uint32_t>((region * a.region_bytes) >> 32);
      for (; region < a.nregions && static_cast<uint32_t>((region * a.region_bytes) >> 32) == slab_hi; region += nwaves) {
        const uint64_t rbase = region * a.region_bytes;
        const uint64_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        // classes of the K bytes before the region, oldest in the low byte; the four raw bytes before it
        uint32_t carry = 0, tail4 = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (rbase >= static_cast<uint64_t>(K - i) ? cls_of(raw_at(rbase - (K - i))) : 0u) << (8 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) tail4 |= (rbase >= static_cast<uint64_t>(4 - i) ? raw_at(rbase - (4 - i)) : static_cast<uint32_t>(g.unused_byte)) << (8 * i);
        tail4 = __builtin_amdgcn_readfirstlane(tail4);
        uint32_t mcarry;  // M word of the K-gram ending just before the region
        {
            uint32_t x = (((carry >> (8 * (K - 1))) & 0xffu) << 2) + offM;
            x += __umul24((carry >> (8 * (K - 2))) & 0xffu, C4);
            if (K == 3) x += __umul24(carry & 0xffu, CC4);
            mcarry = __builtin_amdgcn_readfirstlane(lds_u32(x));
        }

        // the chunks of the step at s0; past the region's end only lane 0's first chunk (it feeds the last step's trailer)
        auto fetch = [&](uint64_t s0, uint4 (&out)[Q]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) out[q] = uint4{ub4, ub4, ub4, ub4};
            if (s0 < rend) {
#pragma unroll
                for (int q = 0; q < Q; ++q) out[q] = load_chunk(s0 + lane * P + 16u * q);
            } else if (lane == 0 && s0 < rend + SB) {
                out[0] = load_chunk(s0);
            }
        };
        uint4 pf0[Q], pf1[Q];
        fetch(rbase, pf0);
        fetch(rbase + SB, pf1);

        for (uint64_t sb = rbase; sb < rend; sb += SB) {
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint64_t v = sb + lane * P;
            uint4 cur[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) { cur[q] = pf0[q]; pf0[q] = pf1[q]; }
            consume_pending();  // before the next chunk is requested: loads retire in order
            __builtin_amdgcn_s_setprio(0);
            fetch(sb + 2ull * SB, pf1);

            // ---- this step's text into its slot (whatever was queued from the step before last has been consumed) ----
            const uint32_t slot = tb + sl * SLOT;                 // wave-uniform
            const uint32_t my_text = slot + 16u + lane * P;       // LDS address of this lane's first byte
            {
                const uint32_t bias = static_cast<uint32_t>(sb) - (slot + 16u);
                if (sl) posbias1 = bias; else posbias0 = bias;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                *reinterpret_cast<lds3_u32x4 *>(static_cast<uintptr_t>(my_text + 16u * q)) = g3_u32x4_t{cur[q].x, cur[q].y, cur[q].z, cur[q].w};
            if (lane == 0) {
                *reinterpret_cast<lds3_u32 *>(static_cast<uintptr_t>(slot + 12u)) = tail4;
                *reinterpret_cast<lds3_u32x4 *>(static_cast<uintptr_t>(slot + 16u + SB)) = g3_u32x4_t{pf0[0].x, pf0[0].y, pf0[0].z, pf0[0].w};
            }
            tail4 = __builtin_amdgcn_readlane(cur[Q - 1].w, 63);

            // ---- byte classes of this lane's P positions plus K to the left ----
            uint32_t kx[K + P];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t w[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    kx[K + 16 * q + b] = pin3(cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu));
                    __builtin_assume(kx[K + 16 * q + b] < 32u);
                }
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);  // this lane's last K classes, oldest low
            const uint32_t left = wave_shr1_3(pk, carry);
            carry = __builtin_amdgcn_readlane(pk, 63);
#pragma unroll
            for (int i = 0; i < K; ++i) { kx[i] = (left >> (8 * i)) & 0xffu; __builtin_assume(kx[i] < 32u); }

            // ---- M words of the K-grams ending at j = 0 .. P-1 (the one ending at -1 comes from the lane to the left) ----
            uint32_t H = 0, ccnt = 0, roll = 0, mprev = 0;
#pragma unroll
            for (int grp = 0; grp < P / 8; ++grp) {
                uint32_t mw[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = grp * 8 + jj;
                    uint32_t x = pin3((kx[K + j] << 2) + offM);                       // 4 c_j + offM              (v_lshl_add_u32)
                    x = __umul24(kx[K + j - 1], C4) + x;                              // + 4 C c_(j-1)             (v_mad_u32_u24)
                    if (K == 3) x = __umul24(kx[K + j - 2], CC4) + pin3(x);           // + 4 C^2 c_(j-2)           (v_mad_u32_u24)
                    mw[jj] = lds_u32(x);
                }
                if (grp == 0) {
                    // this lane's position -1 is the left neighbour's position P-1: its M word arrives by DPP once the
                    // neighbour's last group is known; until then the hit bit of position 0 is deferred (below)
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = grp * 8 + jj;
                    roll = __builtin_amdgcn_alignbit(roll, mw[jj], 30);               // two count bits per position
                    if (j > 0) H |= __builtin_amdgcn_ubfe(jj == 0 ? mprev : mw[jj - 1], kx[K + j], 1) << j;
                }
                mprev = mw[7];
                if ((grp & 1) == 1 || grp == P / 8 - 1) {  // 16 positions rolled in: sum the two-bit fields
                    ccnt += __popc(roll & 0x55555555u) + 2u * __popc(roll & 0xaaaaaaaau);
                    roll = 0;
                }
            }
            {   // position 0 against the M word of the K-gram ending just before this lane's share
                const uint32_t mleft = wave_shr1_3(mprev, mcarry);
                mcarry = __builtin_amdgcn_readlane(mprev, 63);
                H |= __builtin_amdgcn_ubfe(mleft, kx[K], 1);
            }
            cnt32 += ccnt;

            // ---- queue the hits, one per lane and turn ----
            bool did_batch = false;
            for (;;) {
                const bool has = H != 0;
                const unsigned long long m = __ballot(has);
                if (m == 0) break;
                if (has) {
                    const uint32_t b = static_cast<uint32_t>(__builtin_ctz(H));
                    H &= H - 1u;
                    const uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                    *reinterpret_cast<lds3_u32 *>(static_cast<uintptr_t>(ringb + ((at & (kRing3 - 1u)) << 2))) = my_text + b;
                }
                q_tail += static_cast<uint32_t>(__popcll(m));
                if (q_tail - q_head >= 64u) { process_batch(64u); did_batch = true; }
            }
            // whatever was queued a step ago must be gone before its slot is written again
            if (carry_in != 0 && !did_batch) process_batch(q_tail - q_head);
            carry_in = q_tail - q_head;
            sl ^= 1u;
        }
        tot_cnt += cnt32;  // per region: 32 bits cannot overflow within one
        cnt32 = 0;
      }
      if (q_tail != q_head) process_batch(q_tail - q_head);
      carry_in = 0;
      consume_pending();
      finish_second();
      drain();
      tot_cnt += cnt32;
      cnt32 = 0;
    }
    g3_reduce(tot_cnt, reinterpret_cast<unsigned long long *>(smem), a.result);
}


// One kernel, both variants: the workgroup stages the tables, decides TAIL (a.sel_want: 0 / 1, or 2 = by its own density probe)
// and runs the body compiled for that choice.
//
// Density probe: every thread of the workgroup looks at one position (spread over the haystack, the same ones in every
// workgroup) the way the consumer looks at a hit, with the tables it has just staged; one record gather from L2 for the few
// that hit.  Text of dictionary words sends a tenth of its positions below depth K + 1, random text over the dictionary's
// letters well under one in a hundred: TAIL when more than kProbePercent of the samples do.  (Two launches with a probe kernel
// in front cost 60-90 us per scan; a run-time TAIL flag inside one body 10 % of the kernel: profiles/r03_gram3_ab.txt.)
template <int K, int Q, bool S16, bool RFULL, int TPB>
__global__ __launch_bounds__(TPB) void gram3_kernel(const Gram2Dev g, const GramArgs a, const Gram3Lds L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t offM = kOffM3, offS = L.off_s;
    g3_copy(smem, g.cls, 256);
    g3_copy(smem + offM, g.m, g.m_bytes);
    if (RFULL) g3_copy(smem + offS, g.rfull, g.rfull_bytes); else g3_copy(smem + offS, g.sdir, g.s_bytes);
    uint32_t *votes = reinterpret_cast<uint32_t *>(smem + L.off_wave);  // (the first wave's text slot: not in use yet)
    if (threadIdx.x == 0) *votes = 0;
    __syncthreads();
    // tables are read through absolute LDS addresses (this kernel has no static LDS: the dynamic segment starts at 0)
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    bool tail = a.sel_want == 1u;
    if (a.sel_want == 2u) {
        const uint64_t span = a.vlen > a.lead + 16 ? a.vlen - a.lead - 8 : 0;
        bool go = false;
        if (span != 0) {
            unsigned long long h = (static_cast<unsigned long long>(threadIdx.x) + 1) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            const uint64_t p = a.lead + 3 + (h % (span - 3));  // hit byte; p - 3 .. p + 1 lie inside the haystack
            const uint8_t *t = a.hay_al + p;
            const uint8_t *cl = reinterpret_cast<const uint8_t *>(smem);
            const uint32_t *m = reinterpret_cast<const uint32_t *>(smem + offM);
            const uint32_t c0 = cl[t[-3]], c1 = cl[t[-2]], c2 = cl[t[-1]], d = cl[t[0]], k1 = cl[t[1]];
            const uint32_t ctx = K == 3 ? (c0 * g.C + c1) * g.C + c2 : c1 * g.C + c2;
            const uint32_t w = m[ctx];
            if (d != 0 && ((w >> d) & 1u)) {
                uint32_t rank = __popc(w & ((1u << d) - 1u));
                if (RFULL) {
                    rank += reinterpret_cast<const uint16_t *>(smem + offS)[ctx];
                } else {
                    for (uint32_t j = ctx & ~3u; j < ctx; ++j) rank += __popc(m[j] & 0x3fffffffu);
                    rank += S16 ? reinterpret_cast<const uint16_t *>(smem + offS)[ctx >> 2] : reinterpret_cast<const uint32_t *>(smem + offS)[ctx >> 2];
                }
                const uint2 r = g.dhit_c[rank];
                go = k1 != 0 && ((r.x >> k1) & 1u);
            }
        }
        const unsigned long long bm = __ballot(go);
        if ((threadIdx.x & 63) == 0 && bm != 0) atomicAdd(votes, static_cast<uint32_t>(__popcll(bm)));
        __syncthreads();
        tail = *votes * 100u > static_cast<uint32_t>(TPB) * kProbePercent;
        __syncthreads();
    }
    if (tail) gram3_body<K, Q, S16, RFULL, true>(g, a, L, smem);
    else gram3_body<K, Q, S16, RFULL, false>(g, a, L, smem);
}

template <int K, int Q, bool S16, bool RFULL, int TPB>
static hipError_t launch3_inst(const Gram2Dev &dev, const GramArgs &a, const Gram3Lds &L, uint32_t blocks, hipStream_t stream) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram3_kernel<K, Q, S16, RFULL, TPB>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L.lds_bytes));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gram3_kernel<K, Q, S16, RFULL, TPB>), dim3(blocks), dim3(TPB), L.lds_bytes, stream, dev, a, L);
    return hipGetLastError();
}
template <int K, int Q, int TPB>
static hipError_t launch3_k(const Gram2Dev &dev, const GramArgs &a, const Gram3Lds &L, uint32_t blocks, hipStream_t stream) {
    if (L.rfull) return launch3_inst<K, Q, true, true, TPB>(dev, a, L, blocks, stream);
    if (dev.s16) return launch3_inst<K, Q, true, false, TPB>(dev, a, L, blocks, stream);
    return launch3_inst<K, Q, false, false, TPB>(dev, a, L, blocks, stream);
}

// LDS plan of a gram3 launch of `waves` waves per workgroup with `ppl` positions per lane and step.  Returns false when the
// tables and the per-wave areas do not fit.
bool gram3_plan(const Gram2Dev &dev, uint32_t ppl, uint32_t waves, bool want_rfull, uint32_t lds_limit, Gram3Lds &L) {
    L = Gram3Lds{};
    const uint32_t slot = 64u * ppl + 32u;
    L.wave_stride = 2u * slot + kRing3 * 4u;
    const uint32_t per_wg = waves * L.wave_stride;
    const uint32_t off_s = kOffM3 + dev.m_bytes;
    if (want_rfull && dev.rfull != nullptr && off_s + dev.rfull_bytes + per_wg <= lds_limit) {
        L.rfull = 1;
        L.off_wave = off_s + dev.rfull_bytes;
    } else {
        L.rfull = 0;
        L.off_wave = off_s + dev.s_bytes;
    }
    L.off_s = off_s;
    L.lds_bytes = L.off_wave + per_wg;
    L.threads = waves * 64u;
    return L.lds_bytes <= lds_limit;
}

// a.sel_want: 0 = plain records, 1 = tail records from the hit record on, 2 = every workgroup decides by its density probe
hipError_t launch_gram3_scan(const Gram2Dev &dev, const GramArgs &a, const Gram3Lds &L, uint32_t blocks, hipStream_t stream) {
    if (a.ppl == 32 && L.threads == 1024)
        return dev.K == 3 ? launch3_k<3, 2, 1024>(dev, a, L, blocks, stream) : launch3_k<2, 2, 1024>(dev, a, L, blocks, stream);
    if (a.ppl == 32)
        return dev.K == 3 ? launch3_k<3, 2, 512>(dev, a, L, blocks, stream) : launch3_k<2, 2, 512>(dev, a, L, blocks, stream);
    return dev.K == 3 ? launch3_k<3, 1, 1024>(dev, a, L, blocks, stream) : launch3_k<2, 1, 1024>(dev, a, L, blocks, stream);
}

}  // namespace daac
