// GRAM engine, `.count()` kernel of rounds 5-6 (gfx950): find_overlapping_iter(haystack).count() of a Standard bytewise automaton
// (reference loop: src/bytewise/iter.rs:133-176 over src/bytewise.rs:1063-1088) with ONE LDS lookup per haystack byte in the main
// path and six per hit.  Tables: gram4.hpp (M words, per-word rank directory, hit and walk records, "no pattern" as the last
// class).  Method: gram_kernels.hip / gram3_kernels.hip — no state chain; an occurrence of at most K bytes is a function of the
// last K classes (two count bits of the M word), a longer one is found from its start: continuation bit of its (K+1)-gram, the
// bit's rank, one record from L2, and a queued goto-only walk for the few branches that go on.
//
// What changed against gram3_kernels.hip and why (profiles/r04_pmc_sq.txt: 21.5 VALU and 3.85 wave-wide LDS accesses per
// haystack byte — at the rates of tools/micro/pipes_bench.hip both pipes were full at 1.3 TB/s):
//   * byte classes by arithmetic where the dictionary's bytes are one range (cfg3: a-z): class = min(byte - lo, C - 1), two
//     VALU and no LDS access — one lookup per byte less in the main path, six per hit less in the consumer;
//   * the rank of a hit from the PER-WORD directory (two LDS reads and one popcount instead of five reads and four): the
//     directory fits beside M once the class table is gone and a step is 16 positions per lane;
//   * the consumer builds one context index and scales it twice (M word, directory entry) instead of carrying an LDS address
//     through three multiply-adds, keeps the four text bytes behind a hit as they are (their classes are only worked out
//     for a record that can go on) and leaves the first child of a branch that goes on to the drain, which runs 64 such
//     branches wide where the batch that found them had three or four;
//   * hit records carry "ends a pattern" in bit 30 and no class needs the `!= 0` test in front of its child bit.
// Round 6: (a) a filter in front of rank + gather (gram4_filter.hpp; the FILT body below): 83 % of uniform text's hits neither end a
// pattern nor go on — a batch of hits is probed in one Bloom word per (K+1)-gram, what passes is re-compacted by a forward permute and
// ranked / gathered 64 at a time; the Bloom array lies where the per-word directory would, so workgroups whose text is made of
// dictionary words (their own probe) keep that directory and the TAIL body.  1 400 -> 1 500 GB/s on cfg3 (profiles/r06_gram4_decomposition.txt).
// (b) the hit queues hold 16-bit offsets into the wave's text slot (4 KB more for the Bloom array).  (c) walker positions are offsets
// from a 2 GiB epoch base: they no longer wrap when a walker crosses a multiple of 4 GiB.
// Roofline: HBM bytes of haystack (1 B read per byte); integer/bit work only, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "device_tables.hpp"
#include "gram4_filter.hpp"

namespace daac {

namespace {

typedef uint32_t g4_u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kRing4 = 128;        // entries of a wave's hit queue (FIFO; at most 63 left over + 64 new), u16 each: the hit byte's offset in the wave's text slot
constexpr uint32_t kProbePercent4 = 3;  // density probe: TAIL when more than 3 % of the sampled positions start a walker
typedef __attribute__((address_space(3))) const uint32_t lds4_cu32;
typedef __attribute__((address_space(3))) uint32_t lds4_u32;
typedef __attribute__((address_space(3))) const uint16_t lds4_cu16;
typedef __attribute__((address_space(3))) uint16_t lds4_u16;
typedef __attribute__((address_space(3))) const uint8_t lds4_cu8;
typedef __attribute__((address_space(3))) g4_u32x4_t lds4_u32x4;

__device__ __forceinline__ uint32_t pin4(uint32_t x) {
    asm("" : "+v"(x));
    return x;
}
// a * b + c on the 24-bit multiplier (left to itself the compiler takes v_mad_u64_u32 here: a register pair, a move and an s_nop each)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
// lane i <- lane i - 1 of `v`; lane 0 keeps `lane0`
__device__ __forceinline__ uint32_t wave_shr1_4(uint32_t v, uint32_t lane0) {
    uint32_t d = lane0;
    asm volatile("s_nop 1\nv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ unsigned long long g4_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void g4_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ void g4_reduce(unsigned long long cnt, unsigned long long *scratch, unsigned long long *result) {
    const unsigned long long c = g4_wave_sum(cnt);
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

// K = context length; Q = 16-byte chunks a lane takes per step (P = 16 Q positions); ARITH = classes by min(byte - lo, C - 1) (else the
// 256-byte table in LDS); DIR = 0: one u16 directory entry per M word, 1 / 2: one u16 / u32 entry per four words; TAIL = tail records
// from the hit record on and a second pending stage (text made of dictionary words); FILT = a batch of hits goes through the LDS
// filter of gram4_filter.hpp first and only what passes — collected 64 at a time — is ranked and asks the L2 for its record (round 6;
// coarse directory, plain records: text made of dictionary words passes the filter anyway and keeps the TAIL body)
template <int K, int Q, bool ARITH, int DIR, bool TAIL, bool FILT>
__device__ __forceinline__ void gram4_body(const Gram4Dev &g, const GramArgs &a, const Gram4Lds &L, char *smem) {
    static_assert(!(FILT && TAIL), "the filter runs in front of the plain records");
    static_assert(!(FILT && DIR == 0), "the filter's Bloom array lies where the per-word directory would");
    constexpr int P = 16 * Q;
    constexpr int GS = 8;
    constexpr uint32_t SB = 64u * P;          // bytes a wave takes per step
    constexpr uint32_t SLOT = SB + 32u;       // [12,16) the four bytes before the step | [16, 16 + SB) the step | 16 bytes of the next
    // ONE text slot per wave: what is left in the hit queue at the end of a step (fewer than 64 entries) has its text taken out of the
    // slot into registers before the next step's text goes in (`derive` below).  Two slots would not leave the per-word rank directory
    // room beside M at 32 positions per lane (cfg3: 79 + 39 KB of tables, 16 x 2.6 KB of slots and queues).
    const uint32_t offM = L.off_m, offS = L.off_s, offC = L.off_cls;
    const uint32_t lo = g.lo, OTH = g.C - 1u, C = g.C;
    auto cls_of = [&](uint32_t byte) -> uint32_t {
        if (ARITH) {
            const uint32_t u = byte - lo;
            return u < OTH ? u : OTH;
        }
        return *reinterpret_cast<lds4_cu8 *>(static_cast<uintptr_t>(offC + byte));
    };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<lds4_cu32 *>(static_cast<uintptr_t>(addr)); };
    auto below = [&](uint32_t k) -> uint32_t { return (1u << k) - 1u; };  // k <= 29

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C4 = C * 4u, CC4 = C * C * 4u;
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    // per-wave LDS: the text slot; the hit queues of all waves come first (each starts at a multiple of its size)
    const uint32_t tb = L.off_wave + wave_in_wg * L.wave_stride;   // wave-uniform
    const uint32_t ringb = wave_in_wg * (kRing4 * 2u);
    const uint32_t ringb_v = pin4(ringb);   // (the same in a vector register, for the queue store's v_and_or_b32: one scalar / literal operand per VOP3)
    // this wave's slab of pending walkers, 16-byte entries.  plain: {position of the hit byte, hit record x, hit record y, the four text
    // bytes behind the hit}; TAIL: {position, state | class << 27, text bytes from position + 2 on, three more | how many << 24}
    const uint64_t slab_index = (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg) * a.wq_slab;
    uint4 *__restrict__ slab4 = reinterpret_cast<uint4 *>(a.wq) + slab_index;
    // Positions in the queues, the pending stages and the slab are 32-bit offsets from `epoch_base`, a multiple of 2 GiB: a region (a power
    // of two of at most 1 GiB) lies inside one epoch, an offset stays below 2^31 + the longest pattern however far a walker moves on, and
    // the slab is emptied before the epoch changes.  (Until round 6 the epoch was 4 GiB and the offset the position's low word: a walker
    // that walked across a multiple of 4 GiB wrapped to 0 and read its text 4 GiB too early.)
    uint32_t wq_n = 0, slab_hi = 0;  // wave-uniform; slab_hi = epoch_base >> 31
    uint64_t epoch_base = 0;
    const uint32_t drain_early = wave_in_wg * 64u;   // the waves of a workgroup drain at different fill levels (they fill at the same pace: all
                                                     // sixteen waiting on the same round trips at once left the CU idle)

    unsigned long long tot_cnt = 0;
    uint32_t cnt32 = 0;  // matches of the current region (a region is far too short to overflow 32 bits)

    auto load_chunk = [&](uint64_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const g4_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const g4_u32x4_t *>(hay + v));
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
    // Finishes the queued branches.  One pass moves every branch ONE state on, 64 branches wide, and writes those that still go on
    // back to the front of the slab (compacted by ballot); passes repeat until nothing is left.  (Round 4's drain followed every
    // branch to its end inside one loop: the wave then runs as long as its deepest branch — on random text nine branches in ten end
    // at their first record, and the few that do not kept 64 lanes busy for five or six more rounds: 12 % of the kernel.)
    auto drain = [&]() {
        const uint4 *__restrict__ recs = TAIL ? g.drec_t : g.drec_c;
        uint32_t n_in = wq_n;
        bool raw = !TAIL;   // plain: the first pass reads what the batches left: {position, hit record x, hit record y, four text bytes}
        while (n_in != 0) {
            uint32_t n_out = 0;
            for (uint32_t base = 0; base < n_in; base += 64u) {
                const uint32_t i = base + lane;
                const bool live = i < n_in;
                uint4 e = uint4{0u, 0u, 0u, 0u};
                if (live) e = slab4[i];
                uint32_t state, n_ahead;
                unsigned long long ah;
                if (raw) {
                    const uint32_t k1 = cls_of(e.w & 0xffu);
                    state = e.z + __popc(e.y & below(k1));
                    ah = e.w >> 8;
                    n_ahead = 3;
                } else {
                    state = e.y & 0x07ffffffu;
                    ah = (static_cast<unsigned long long>(e.w & 0xffffffu) << 32) | e.z;  // the bytes from vnext on (e.w >> 24 of them)
                    n_ahead = e.w >> 24;
                }
                uint4 rr = uint4{0u, 0u, 0u, 0u};
                if (live) rr = recs[state];  // {cmap, first_child, own_cnt, -} or a tail record
                const uint64_t vn = epoch_base + e.x + 2;  // the state consumed the byte before vn
                bool cont = false;
                uint32_t next_state = 0;
                if (rr.x >> 31) {   // the rest is one path (idle lanes: a zero record, nothing happens)
                    if (n_ahead < (rr.x & 15u)) ah = read_ahead(vn);
                    cnt32 += tail_count(rr, ah);
                } else {
                    if (live && n_ahead == 0u) { ah = read_ahead(vn); n_ahead = 8; }
                    const uint32_t k = cls_of(static_cast<uint32_t>(ah) & 0xffu);
                    cnt32 += rr.z;
                    cont = ((rr.x >> k) & 1u) != 0;   // (the class of no pattern has no bit in a child map)
                    next_state = rr.y + __popc(rr.x & below(k));
                }
                const unsigned long long m = __ballot(cont);
                if (m != 0) {
                    if (cont) {
                        const uint32_t at = n_out + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                        ah >>= 8;
                        slab4[at] = uint4{e.x + 1u, next_state, static_cast<uint32_t>(ah), static_cast<uint32_t>(ah >> 32) | ((n_ahead - 1u) << 24)};
                    }
                    n_out += static_cast<uint32_t>(__popcll(m));
                }
            }
            n_in = n_out;
            raw = false;
        }
        wq_n = 0;
    };

    // ---- the hit queue: entry = offset of the hit byte from the start of the wave's text slot (u16)
    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running; entries live at (index & (kRing4 - 1))
    uint32_t posbias = 0;              // (virtual position of a byte - epoch_base) - (its LDS address), of the text in the slot
    uint32_t st_n = 0;                 // wave-uniform: lanes [0, st_n) hold an entry of the NEXT batch already taken out of queue and slot
    uint32_t pend_lo = 0;              // the bytes p-3 .. p of the next batch's entry (p = its hit byte); pend_pos / pend_t0 / pend_t1 go with it
    uint4 pend = uint4{0u, 0u, 0u, 0u};  // record read for the previous batch, not yet consumed; zero for idle lanes
                                         // {cmap | ends-a-pattern << 30, first_child, -, -}; TAIL: or a tail record
    uint32_t pend_pos = 0;             // position of the hit byte
    uint32_t pend_t0 = 0, pend_t1 = 0; // the four bytes from position + 1 on; TAIL: and the four behind them
    bool pend_valid = false;           // wave-uniform
    // TAIL: a branch that goes on past the hit's state asks for the record of the NEXT state at once (p2, looked at one batch later)
    uint4 p2 = uint4{0u, 0u, 0u, 0u};    // record of the state below the hit's; zero for idle lanes
    uint32_t p2_pos = 0, p2_state = 0;   // position of the hit byte; the state asked for | class of the byte at position + 2 << 27
    uint32_t p2_t0 = 0, p2_t1 = 0;       // the seven bytes from position + 2 on
    bool p2_live = false, p2_any = false;        // per lane / wave-uniform
    // FILT: what has passed the filter waits in lanes [0, sb_n) — position, the four bytes up to the hit byte, the four behind it — until 64
    // are together; the batch whose records are in flight then has its position and text in fp_pos / fp_t0 (pend_* belong to the batch
    // that is going through the filter)
    uint32_t sb_n = 0;                 // wave-uniform
    uint32_t sb_pos = 0, sb_lo = 0, sb_t0 = 0;
    uint32_t fp_pos = 0, fp_t0 = 0;
    const uint32_t offB = L.off_b, bloomW = g.bloom_words;
    auto push_walker = [&](bool go, const uint4 &entry) {
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go) {
                const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                slab4[at] = entry;
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
            if (r.x >> 31) {
                if ((r.x & 15u) <= 7u) cnt32 += tail_count(r, (static_cast<unsigned long long>(p2_t1) << 32) | p2_t0);
                else { again = true; st_k = p2_state; }  // eight edges, seven bytes at hand: the drain looks at the record again with the eighth byte fetched
            } else {
                cnt32 += r.z;
                const uint32_t k = p2_state >> 27;
                if ((r.x >> k) & 1u) {
                    again = true;
                    const uint32_t k3 = cls_of((p2_t0 >> 8) & 0xffu);
                    st_k = (r.y + __popc(r.x & below(k))) | (k3 << 27);
                    pos = p2_pos + 1u;
                    t0 = __builtin_amdgcn_alignbyte(p2_t1, p2_t0, 1u);
                    t1n = ((p2_t1 >> 8) & 0xffffu) | (6u << 24);
                }
            }
        }
        p2_live = false;
        push_walker(again, uint4{pos, st_k, t0, t1n});
    };
    auto consume_pending = [&]() {
        finish_second();
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;
        if (FILT) {
            const uint32_t k1 = cls_of(fp_t0 & 0xffu);
            cnt32 += (r.x >> kGram4EndsBitDev) & 1u;
            push_walker(__builtin_amdgcn_ubfe(r.x, k1, 1) != 0, uint4{fp_pos, r.x, r.y, fp_t0});
            return;
        }
        const uint32_t k1 = cls_of(pend_t0 & 0xffu);
        if (!TAIL) {
            cnt32 += (r.x >> kGram4EndsBitDev) & 1u;
            // (the first child of a branch that goes on is worked out in the drain, 64 branches wide: here three or four lanes of 64 go on)
            push_walker(__builtin_amdgcn_ubfe(r.x, k1, 1) != 0, uint4{pend_pos, r.x, r.y, pend_t0});
            return;
        }
        bool go;
        if (r.x >> 31) {  // one path below the hit: settled here
            cnt32 += tail_count(r, (static_cast<unsigned long long>(pend_t1) << 32) | pend_t0);
            go = false;
        } else {
            cnt32 += (r.x >> kGram4EndsBitDev) & 1u;
            go = ((r.x >> k1) & 1u) != 0;
        }
        p2 = uint4{0u, 0u, 0u, 0u};
        p2_live = go;
        p2_any = __ballot(go) != 0;
        if (go) {
            const uint32_t child = r.y + __popc(r.x & below(k1));
            const uint32_t k2 = cls_of((pend_t0 >> 8) & 0xffu);
            p2 = g.drec_t[child];
            p2_pos = pend_pos;
            p2_state = child | (k2 << 27);
            p2_t0 = __builtin_amdgcn_alignbyte(pend_t1, pend_t0, 1u);  // text from position + 2 on
            p2_t1 = pend_t1 >> 8;
        }
    };
    // Takes `cnt` entries from the head of the queue into lanes [first, first + cnt): position, the four bytes up to the hit byte, the
    // four (TAIL: eight) behind it.  After this the entries no longer refer to the slot.
    auto derive = [&](uint32_t first, uint32_t cnt) {
        if (lane - first < cnt) {
            const uint32_t e = tb + *reinterpret_cast<lds4_cu16 *>(static_cast<uintptr_t>(ringb | (((q_head + lane - first) << 1) & (kRing4 * 2u - 2u))));
            pend_pos = e + posbias;
            const uint32_t t3 = e - 3u;
            const uint32_t a0 = t3 & ~3u, sh = t3 & 3u;
            // the dwords around the hit byte (the slot is self-contained: never outside [slot + 12, slot + SLOT))
            const uint32_t d0 = lds_u32(a0), d1 = lds_u32(a0 + 4u), d2 = lds_u32(a0 + 8u);
            pend_lo = __builtin_amdgcn_alignbyte(d1, d0, sh);              // bytes p-3 .. p
            pend_t0 = __builtin_amdgcn_alignbyte(d2, d1, sh);              // bytes p+1 .. p+4
            if (TAIL) {
                const uint32_t d3 = lds_u32(a0 + 12u);
                pend_t1 = __builtin_amdgcn_alignbyte(d3, d2, sh);          // bytes p+5 .. p+8
            }
        }
        q_head += cnt;
    };
    // rank of the hit whose bytes p-3 .. p are x_lo -> its record asked for (into `pend`)
    auto rank_and_ask = [&](uint32_t x_lo) {
        const uint32_t c1 = cls_of((x_lo >> 8) & 0xffu), c2 = cls_of((x_lo >> 16) & 0xffu), d = cls_of(x_lo >> 24);
        uint32_t idx = __umul24(c1, C) + c2;
        if (K == 3) idx = __umul24(cls_of(x_lo & 0xffu), C * C) + idx;
        // rank of continuation bit d of that M word among all set bits = offset of the depth-(K+1) state
        const uint32_t am = (idx << 2) + offM;
        const uint32_t own = lds_u32(am);
        uint32_t rank;
        if (DIR == 0) {
            const uint32_t base = *reinterpret_cast<lds4_cu16 *>(static_cast<uintptr_t>((idx << 1) + offS));
            rank = base + __popc(own & below(d));
        } else {
            const uint32_t grp = offM + ((idx & ~3u) << 2);
            const uint32_t qx = lds_u32(grp), qy = lds_u32(grp + 4u), qz = lds_u32(grp + 8u);
            const uint32_t sub = idx & 3u;
            const uint32_t base = DIR == 1 ? *reinterpret_cast<lds4_cu16 *>(static_cast<uintptr_t>(offS + ((idx >> 2) << 1)))
                                           : *reinterpret_cast<lds4_cu32 *>(static_cast<uintptr_t>(offS + ((idx >> 2) << 2)));
            uint32_t under = __popc(own & below(d));
            under += sub > 0 ? __popc(qx & 0x3fffffffu) : 0u;
            under += sub > 1 ? __popc(qy & 0x3fffffffu) : 0u;
            under += sub > 2 ? __popc(qz & 0x3fffffffu) : 0u;
            rank = base + under;
        }
        if (TAIL) {
            pend = g.dhit_t[rank];
        } else {
            const uint2 h = g.dhit_c[rank];
            pend = uint4{h.x, h.y, 0u, 0u};
        }
    };
    // FILT, second stage: the n <= 64 survivors in lanes [0, n) of sb_* are ranked and ask for their records
    auto survivors_ask = [&](uint32_t n) {
        consume_pending();
        pend = uint4{0u, 0u, 0u, 0u};
        if (lane < n) rank_and_ask(sb_lo);
        fp_pos = sb_pos;
        fp_t0 = lane < n ? sb_t0 : 0u;   // (an idle lane: nothing of it may look like a branch that goes on)
        pend_valid = true;
    };
    auto process_batch = [&](uint32_t n) {  // n <= 64 entries: the st_n already taken out + the head of the queue
        __builtin_amdgcn_s_setprio(2);
        if (FILT) {
            derive(st_n, n - st_n);
            st_n = 0;
            // first stage: the probe of gram4_filter.hpp on the raw bytes p-K .. p and p + 1 — one word, the GO key's two bits or the ENDS key's four
            const G4Probe pr = g4f_probe(K == 3 ? pend_lo : pend_lo >> 8, pend_t0 & 0xffu, bloomW);
            const uint32_t fw = lds_u32(offB + (pr.word << 2));
            const bool pass = lane < n && ((fw & pr.go) == pr.go || (fw & pr.ends) == pr.ends);
            const unsigned long long pm = __ballot(pass);
            if (pm != 0) {
                // survivors move to lanes sb_n, sb_n + 1, .. (mod 64) — one forward permute per word, the others fill the lanes in between —
                const uint32_t s = static_cast<uint32_t>(__popcll(pm));
                const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0));
                const uint32_t tgt = ((pass ? sb_n + r : sb_n + s + lane - r) & 63u) << 2;
                const uint32_t r_pos = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>(tgt), static_cast<int>(pend_pos)));
                const uint32_t r_lo = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>(tgt), static_cast<int>(pend_lo)));
                const uint32_t r_t0 = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>(tgt), static_cast<int>(pend_t0)));
                const bool mine = lane - sb_n < s;   // lanes [sb_n, sb_n + s) below 64: they join the waiting ones
                sb_pos = mine ? r_pos : sb_pos;
                sb_lo = mine ? r_lo : sb_lo;
                sb_t0 = mine ? r_t0 : sb_t0;
                if (sb_n + s >= 64u) {   // 64 together: ranked and asked for; what went beyond lane 63 arrived in lanes 0 .. and waits on
                    survivors_ask(64u);
                    sb_pos = r_pos; sb_lo = r_lo; sb_t0 = r_t0;
                    sb_n = sb_n + s - 64u;
                } else {
                    sb_n += s;
                }
            }
            return;
        }
        consume_pending();
        derive(st_n, n - st_n);
        st_n = 0;
        pend = uint4{0u, 0u, 0u, 0u};
        if (lane < n) {
            rank_and_ask(pend_lo);
        } else {
            pend_t0 = 0;   // (an idle lane: nothing of it may look like a branch that goes on)
            pend_t1 = 0;
        }
        pend_valid = true;
    };

    uint64_t region = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    while (region < a.nregions) {
      slab_hi = static_cast<uint32_t>((region * a.region_bytes) >> 31);
      epoch_base = static_cast<uint64_t>(slab_hi) << 31;
      for (; region < a.nregions && static_cast<uint32_t>((region * a.region_bytes) >> 31) == slab_hi; region += nwaves) {
        const uint64_t rbase = region * a.region_bytes;
        const uint64_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        // classes of the K bytes before the region, oldest in the low byte; the four raw bytes before it
        uint32_t carry = 0, tail4 = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (rbase >= static_cast<uint64_t>(K - i) ? cls_of(raw_at(rbase - (K - i))) : OTH) << (8 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) tail4 |= (rbase >= static_cast<uint64_t>(4 - i) ? raw_at(rbase - (4 - i)) : static_cast<uint32_t>(g.unused_byte)) << (8 * i);
        tail4 = __builtin_amdgcn_readfirstlane(tail4);
        carry = __builtin_amdgcn_readfirstlane(carry);
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
            if (wq_n + 64u * P + 128u + drain_early > a.wq_slab) drain();
            uint4 cur[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) { cur[q] = pf0[q]; pf0[q] = pf1[q]; }
            consume_pending();  // before the next chunk is requested: loads retire in order
            __builtin_amdgcn_s_setprio(0);
            fetch(sb + 2ull * SB, pf1);

            // ---- what the last step left in the queue leaves the slot; then this step's text goes in ----
            if (q_tail != q_head) {
                const uint32_t n_left = q_tail - q_head;
                derive(st_n, n_left);
                st_n += n_left;
            }
            const uint32_t slot = tb;                             // wave-uniform
            const uint32_t my_text = slot + 16u + lane * P;       // LDS address of this lane's first byte
            posbias = static_cast<uint32_t>(sb - epoch_base) - (slot + 16u);
#pragma unroll
            for (int q = 0; q < Q; ++q)
                *reinterpret_cast<lds4_u32x4 *>(static_cast<uintptr_t>(my_text + 16u * q)) = g4_u32x4_t{cur[q].x, cur[q].y, cur[q].z, cur[q].w};
            if (lane == 0) {
                *reinterpret_cast<lds4_u32 *>(static_cast<uintptr_t>(slot + 12u)) = tail4;
                *reinterpret_cast<lds4_u32x4 *>(static_cast<uintptr_t>(slot + 16u + SB)) = g4_u32x4_t{pf0[0].x, pf0[0].y, pf0[0].z, pf0[0].w};
            }
            tail4 = __builtin_amdgcn_readlane(cur[Q - 1].w, 63);

            // ---- byte classes of this lane's P positions plus K to the left ----
            uint32_t kx[K + P];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t w[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    kx[K + 16 * q + b] = pin4(cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu));
                    __builtin_assume(kx[K + 16 * q + b] < 32u);
                }
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);  // this lane's last K classes, oldest low
            const uint32_t left = wave_shr1_4(pk, carry);
            carry = __builtin_amdgcn_readlane(pk, 63);
#pragma unroll
            for (int i = 0; i < K; ++i) { kx[i] = (left >> (8 * i)) & 0xffu; __builtin_assume(kx[i] < 32u); }

            // ---- M words of the K-grams ending at j = 0 .. P-1 (the one ending at -1 comes from the lane to the left) ----
            uint32_t H = 0, ccnt = 0, roll = 0, mprev = 0;
#pragma unroll
            for (int grp = 0; grp < P / GS; ++grp) {
                uint32_t mw[GS];
#pragma unroll
                for (int jj = 0; jj < GS; ++jj) {
                    const int j = grp * GS + jj;
                    uint32_t x = pin4((kx[K + j] << 2) + offM);                       // 4 c_j + offM              (v_lshl_add_u32)
                    x = mad24(kx[K + j - 1], C4, x);                                  // + 4 C c_(j-1)             (v_mad_u32_u24)
                    if (K == 3) x = mad24(kx[K + j - 2], CC4, x);                     // + 4 C^2 c_(j-2)           (v_mad_u32_u24)
                    mw[jj] = lds_u32(x);
                }
#pragma unroll
                for (int jj = 0; jj < GS; ++jj) {
                    const int j = grp * GS + jj;
                    roll = __builtin_amdgcn_alignbit(roll, mw[jj], 30);               // two count bits per position
                    // the hit bit of position j enters at the top: after the step's P - 1 shifts it sits at bit 32 - P + j
                    if (j > 0) H = __builtin_amdgcn_alignbit((jj == 0 ? mprev : mw[jj - 1]) >> kx[K + j], H, 1);   // (v_lshrrev_b32, v_alignbit_b32)
                    if ((j & 15) == 15) {  // 16 positions rolled in: sum the two-bit fields
                        ccnt += __popc(roll & 0x55555555u) + 2u * __popc(roll & 0xaaaaaaaau);
                        roll = 0;
                    }
                }
                mprev = mw[GS - 1];
            }
            {   // position 0 against the M word of the K-gram ending just before this lane's share
                const uint32_t mleft = wave_shr1_4(mprev, mcarry);
                mcarry = __builtin_amdgcn_readlane(mprev, 63);
                H |= __builtin_amdgcn_ubfe(mleft, kx[K], 1) << (32 - P);
            }
            cnt32 += ccnt;

            // ---- queue the hits, one per lane and turn ----
            const uint32_t text_adj = my_text - (32u - P) - tb;   // (queue entries are offsets in the wave's slot: 16 bits)
            for (;;) {
                const bool has = H != 0;
                const unsigned long long m = __ballot(has);
                if (m == 0) break;
                // every lane computes (an idle lane's entry is never stored); the ring of a wave starts at a multiple of its size
                uint32_t b;
                asm("v_ffbl_b32 %0, %1" : "=v"(b) : "v"(H));   // (all lanes: -1 where there is no bit)
                const uint32_t entry = text_adj + b;
                H &= H - 1u;
                // slot = ((q_tail + lanes with a hit below this one) * 2 & ring mask) | ring base, in four instructions: the two mbcnt, one
                // add-and-shift with q_tail as the scalar operand, one and-or (left to itself the compiler moves q_tail into a register for
                // mbcnt's accumulator — both of mbcnt_lo's other operands are scalar already — and splits the and-or: six; this loop turns
                // 9.5 times per step)
                const uint32_t below_me = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                uint32_t at2, slot_addr;
                asm("v_add_lshl_u32 %0, %1, %2, 1" : "=v"(at2) : "v"(below_me), "s"(q_tail));
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(slot_addr) : "v"(at2), "s"(kRing4 * 2u - 2u), "v"(ringb_v));   // (VOP3 takes no literal on gfx950: the mask is a scalar register)
                if (has) *reinterpret_cast<lds4_u16 *>(static_cast<uintptr_t>(slot_addr)) = static_cast<uint16_t>(entry);
                q_tail += static_cast<uint32_t>(__popcll(m));
                if (st_n + q_tail - q_head >= 64u) process_batch(64u);
            }
        }
        tot_cnt += cnt32;  // per region: 32 bits cannot overflow within one
        cnt32 = 0;
      }
      if (st_n + q_tail - q_head != 0) process_batch(st_n + q_tail - q_head);
      if (FILT && sb_n != 0) { survivors_ask(sb_n); sb_n = 0; }
      consume_pending();
      finish_second();
      drain();
      tot_cnt += cnt32;
      cnt32 = 0;
    }
    g4_reduce(tot_cnt, reinterpret_cast<unsigned long long *>(smem), a.result);
}

// One kernel, the variants inside: the workgroup stages M, decides TAIL (a.sel_want: 0 / 1, or 2 = by its own density probe), stages the
// directory that goes with the choice and runs the body compiled for it (gram3_kernels.hip: two launches with a probe kernel in front
// cost 60-90 us per scan, a run-time TAIL flag inside one body 10 % of the kernel).  FILT: a workgroup whose text is not made of
// dictionary words takes [coarse directory | Bloom array] in the place of the per-word directory and runs the body with the filter.
template <int K, int Q, bool ARITH, int DIR, int TPB, bool FILT>
__global__ __launch_bounds__(TPB) void gram4_kernel(const Gram4Dev g, const GramArgs a, const Gram4Lds L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DIRC = DIR == 0 ? 1 : DIR;   // the coarse directory that goes with the filter (a per-word directory exists only beside u16 entries)
    if (!ARITH) g4_copy(smem + L.off_cls, g.cls, 256);
    g4_copy(smem + L.off_m, g.m, g.m_bytes);
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
            const uint32_t OTH = g.C - 1u;
            auto cl = [&](uint32_t b) -> uint32_t {
                if (ARITH) { const uint32_t u = b - g.lo; return u < OTH ? u : OTH; }
                return reinterpret_cast<const uint8_t *>(smem + L.off_cls)[b];
            };
            const uint32_t *m = reinterpret_cast<const uint32_t *>(smem + L.off_m);
            const uint32_t c0 = cl(t[-3]), c1 = cl(t[-2]), c2 = cl(t[-1]), d = cl(t[0]), k1 = cl(t[1]);
            const uint32_t ctx = K == 3 ? (c0 * g.C + c1) * g.C + c2 : c1 * g.C + c2;
            const uint32_t w = m[ctx];
            if ((w >> d) & 1u & (d < OTH ? 1u : 0u)) {
                // (the directory is not staged yet — which one depends on this probe —: its entry comes from device memory)
                uint32_t rank = __popc(w & ((1u << d) - 1u));
                for (uint32_t j = ctx & ~3u; j < ctx; ++j) rank += __popc(m[j] & 0x3fffffffu);
                rank += g.s16 ? reinterpret_cast<const uint16_t *>(g.sdir)[ctx >> 2] : reinterpret_cast<const uint32_t *>(g.sdir)[ctx >> 2];
                const uint2 r = g.dhit_c[rank];
                go = ((r.x >> k1) & 1u) != 0;
            }
        }
        const unsigned long long bm = __ballot(go);
        if ((threadIdx.x & 63) == 0 && bm != 0) atomicAdd(votes, static_cast<uint32_t>(__popcll(bm)));
        __syncthreads();
        tail = *votes * 100u > static_cast<uint32_t>(TPB) * kProbePercent4;
        __syncthreads();
    }
    if (FILT && !tail) {
        g4_copy(smem + L.off_s, g.sdir, L.s_bytes_f);
        g4_copy(smem + L.off_b, g.bloom, g.bloom_words * 4u);
    } else {
        g4_copy(smem + L.off_s, DIR == 0 ? static_cast<const void *>(g.rfull) : g.sdir, L.s_bytes);
    }
    __syncthreads();
    if (tail) gram4_body<K, Q, ARITH, DIR, true, false>(g, a, L, smem);
    else if constexpr (FILT) gram4_body<K, Q, ARITH, DIRC, false, true>(g, a, L, smem);
    else gram4_body<K, Q, ARITH, DIR, false, false>(g, a, L, smem);
}

template <int K, int Q, bool ARITH, int DIR, int TPB, bool FILT>
static hipError_t launch4_inst(const Gram4Dev &dev, const GramArgs &a, const Gram4Lds &L, uint32_t blocks, hipStream_t stream) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram4_kernel<K, Q, ARITH, DIR, TPB, FILT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L.lds_bytes));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gram4_kernel<K, Q, ARITH, DIR, TPB, FILT>), dim3(blocks), dim3(TPB), L.lds_bytes, stream, dev, a, L);
    return hipGetLastError();
}
template <int K, int Q, int TPB, bool ARITH>
static hipError_t launch4_d(const Gram4Dev &dev, const GramArgs &a, const Gram4Lds &L, uint32_t blocks, hipStream_t stream) {
    if (L.filter) {
        if (L.dir == 0) return launch4_inst<K, Q, ARITH, 0, TPB, true>(dev, a, L, blocks, stream);
        if (L.dir == 1) return launch4_inst<K, Q, ARITH, 1, TPB, true>(dev, a, L, blocks, stream);
        return launch4_inst<K, Q, ARITH, 2, TPB, true>(dev, a, L, blocks, stream);
    }
    if (L.dir == 0) return launch4_inst<K, Q, ARITH, 0, TPB, false>(dev, a, L, blocks, stream);
    if (L.dir == 1) return launch4_inst<K, Q, ARITH, 1, TPB, false>(dev, a, L, blocks, stream);
    return launch4_inst<K, Q, ARITH, 2, TPB, false>(dev, a, L, blocks, stream);
}
template <int K, int Q, int TPB>
static hipError_t launch4_k(const Gram4Dev &dev, const GramArgs &a, const Gram4Lds &L, uint32_t blocks, hipStream_t stream) {
    return L.arith ? launch4_d<K, Q, TPB, true>(dev, a, L, blocks, stream) : launch4_d<K, Q, TPB, false>(dev, a, L, blocks, stream);
}

// LDS plan of a gram4 launch of `waves` waves per workgroup with `ppl` positions per lane and step:
// [hit queues | one text slot per wave | class table (256 B, when the classes are not arithmetic) | rank directory | M].
// `rfull`: the per-word directory (false: one entry per four words).  `want_filter`: the directory's place is made large enough for
// [coarse directory | Bloom array] as well, for the workgroups that run the body with the filter (gram4_filter.hpp) — not taken when the
// array was not built or does not fit this shape.  Returns false when the directory asked for is not there or the tables and the
// per-wave areas do not fit — it never hands back another shape than the one asked for.
bool gram4_plan(const Gram4Dev &dev, uint32_t ppl, uint32_t waves, bool rfull, bool want_arith, bool want_filter, uint32_t lds_limit, Gram4Lds &L) {
    L = Gram4Lds{};
    if (rfull && dev.rfull == nullptr) return false;
    const uint32_t slot = 64u * ppl + 32u;
    L.wave_stride = slot;
    L.off_wave = waves * kRing4 * 2u;      // the hit queues sit at 0
    L.threads = waves * 64u;
    L.arith = (want_arith && dev.arith) ? 1u : 0u;
    const uint32_t per_wg = L.off_wave + waves * L.wave_stride;
    L.off_cls = per_wg;
    L.off_s = per_wg + (L.arith ? 0u : 256u);
    L.dir = rfull ? 0u : (dev.s16 ? 1u : 2u);
    L.s_bytes = rfull ? dev.rfull_bytes : dev.s_bytes;
    uint32_t region = L.s_bytes;
    if (want_filter && dev.bloom != nullptr) {
        const uint32_t with = dev.s_bytes + dev.bloom_words * 4u;
        if (L.off_s + std::max(region, with) + dev.m_bytes <= lds_limit) {
            L.filter = 1u;
            L.s_bytes_f = dev.s_bytes;
            L.off_b = L.off_s + dev.s_bytes;
            region = std::max(region, with);
        }
    }
    L.off_m = L.off_s + region;
    L.lds_bytes = L.off_m + dev.m_bytes;
    return L.lds_bytes <= lds_limit;
}
uint32_t gram4_filter_room(uint32_t m_bytes, uint32_t sdir_bytes, bool arith, uint32_t lds_limit) {
    const uint32_t fixed = 16u * kRing4 * 2u + 16u * (64u * 32u + 32u) + (arith ? 0u : 256u) + sdir_bytes + m_bytes;
    return fixed < lds_limit ? (lds_limit - fixed) & ~15u : 0u;
}

// a.sel_want: 0 = plain records, 1 = tail records from the hit record on, 2 = every workgroup decides by its density probe
hipError_t launch_gram4_scan(const Gram4Dev &dev, const GramArgs &a, const Gram4Lds &L, uint32_t blocks, hipStream_t stream) {
    if (a.ppl == 32 && L.threads == 1024)
        return dev.K == 3 ? launch4_k<3, 2, 1024>(dev, a, L, blocks, stream) : launch4_k<2, 2, 1024>(dev, a, L, blocks, stream);
    if (a.ppl == 32)
        return dev.K == 3 ? launch4_k<3, 2, 512>(dev, a, L, blocks, stream) : launch4_k<2, 2, 512>(dev, a, L, blocks, stream);
    if (L.threads == 512)
        return dev.K == 3 ? launch4_k<3, 1, 512>(dev, a, L, blocks, stream) : launch4_k<2, 1, 512>(dev, a, L, blocks, stream);
    return dev.K == 3 ? launch4_k<3, 1, 1024>(dev, a, L, blocks, stream) : launch4_k<2, 1, 1024>(dev, a, L, blocks, stream);
}

}  // namespace daac
