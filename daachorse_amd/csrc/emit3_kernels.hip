// GRAM engine, tuple emission with detection done ONCE (gfx950): the find_overlapping match stream as (start, end, value) tuples in
// the reference's order (bytewise/iter.rs:133-176: by end, longest first; lib.rs:286-320: Match), written to device memory.
//
// gram2_emit_kernels.hip ran the detection twice (COUNT, then WRITE) and gave every lane 16 consecutive end positions, so that a
// store instruction's 64 lanes wrote to 64 different 128-byte lines (profiles/r03_emit.txt: 1.92 + 6.95 ms per GiB of cfg3, 6.3 ms
// of it the stores).  Here:
//
//   DETECT  = the `.count()` kernel's main path and hit path (gram3_kernels.hip: one M word per position, lane-local hit masks,
//             iteration-major queueing, consumer 64 hits wide, goto-only walkers) over the emission tables (gram2.hpp: ME words carry
//             one flag per short pattern length; hit and walk records carry values).  It leaves
//               * ANN[v]: one byte per position = class | flags << 5 (which patterns of 1 .. K bytes end here) — everything EXPAND
//                 needs to know about the short matches of a position and the two positions before it;
//               * per tile of 1024 positions the number of short tuples (plain store by the owning wave) and of deep matches that END
//                 in it (atomic adds: a deep match is found from its START, by whichever wave owns that byte);
//               * every deep match as a 16-byte record {last byte, length | copy, value, tile} in a chunked list: a wave appends to
//                 its own open chunk through an LDS cursor and takes a new chunk (one global atomic) when the open one is half full.
//             No prologue and no carry lists: a match that starts in one region and ends in another is logged by the wave that found it.
//   (host)    tile totals -> exclusive scans (tuple offsets, record offsets); BIN: records grouped by the tile they end in.
//   EXPAND  = a pure expansion of (ANN, bins): the tile's stream goes through LDS so that lane l of column i looks at position
//             64 i + l; per column one wave scan of the per-position tuple counts; a store instruction then writes the tuples of 64
//             CONSECUTIVE positions = neighbouring slots of a handful of lines (the L2 merges them).  Deep matches set a length bit
//             per position (LDS), which gives each its rank; those the bits cannot carry (longer than K + 16 bytes, further copies of
//             a duplicate pattern) are "extras", at most 64 per tile, placed by comparison among themselves.
//
// Roofline: DETECT reads 1 B and writes 1 B (+ records) per haystack byte; EXPAND reads that byte and writes 16 (24) B per tuple:
// HBM write bytes bound the pair (cfg3: 9.5 B of tuples per haystack byte).  Integer / bit work only, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef uint32_t e3_u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kRingE3 = 128;        // entries of a wave's hit queue (FIFO; at most 63 left over + 64 new)
constexpr uint32_t kOffME3 = 256;        // LDS offset of ME (classes at 0)
constexpr uint32_t kContBitsE3 = 0x1ffffffeu;  // continuation bits of an ME word (classes 1 .. 28)
typedef __attribute__((address_space(3))) const uint32_t ldsx_cu32;
typedef __attribute__((address_space(3))) uint32_t ldsx_u32;
typedef __attribute__((address_space(3))) const uint16_t ldsx_cu16;
typedef __attribute__((address_space(3))) const uint8_t ldsx_cu8;
typedef __attribute__((address_space(3))) e3_u32x4_t ldsx_u32x4;

__device__ __forceinline__ uint32_t pinx(uint32_t x) {
    asm("" : "+v"(x));
    return x;
}
// lane i <- lane i - 1 of `v`; lane 0 keeps `lane0`
__device__ __forceinline__ uint32_t wave_shr1_x(uint32_t v, uint32_t lane0) {
    uint32_t d = lane0;
    asm volatile("s_nop 1\nv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ void e3_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
// inclusive scan over the 64 lanes on the VALU (DPP row shifts + row broadcasts); all lanes must be active
__device__ __forceinline__ uint32_t wave_incl_scan_x(uint32_t x) {
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}
// One tuple.  daac_match (24 bytes: start, end, value) takes a 16-byte and an 8-byte store; daac_match16 (the crate's own Match
// fields, src/lib.rs:287-291: end, length, value) ONE 16-byte store.  Plain stores: it is the L2 that puts the lines together.
template <bool F16>
__device__ __forceinline__ void put_tuple_x(void *out, unsigned long long slot, unsigned long long end, uint32_t len, uint32_t value) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (F16) {
        const u32x4 t = {static_cast<uint32_t>(end), static_cast<uint32_t>(end >> 32), len, value};
        *reinterpret_cast<u32x4 *>(static_cast<char *>(out) + slot * 16ull) = t;
    } else {
        char *dst = static_cast<char *>(out) + slot * 24ull;
        const u64x2 se = {end - len, end};
        const u32x2 vp = {value, 0u};
        *reinterpret_cast<u64x2 *>(dst) = se;
        *reinterpret_cast<u32x2 *>(dst + 16) = vp;
    }
}

}  // namespace

// ================================================================================================================ DETECT
// K = context length; S16 = rank directory entries are u16.  32 positions per lane and step (2 KiB per wave-step = two tiles).
template <int K, bool S16>
__global__ __launch_bounds__(1024) void emit3_detect_kernel(const Gram2EmitDev g, const Emit3Args a, const Gram3Lds L) {
    constexpr int Q = 2;
    constexpr int P = 16 * Q;
    constexpr uint32_t SB = 64u * P;          // bytes a wave takes per step
    constexpr uint32_t SLOT = SB + 32u;       // [0,4) slot 0 only: the wave's record cursor | [12,16) the four bytes before the step | [16, 16 + SB) the step | 16 bytes of the next
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t offM = kOffME3, offS = L.off_s;
    e3_copy(smem, g.cls, 256);
    e3_copy(smem + offM, g.me, g.m_bytes);
    e3_copy(smem + offS, g.sdir, g.s_bytes);
    __syncthreads();
    // tables are read through absolute LDS addresses (this kernel has no static LDS: the dynamic segment starts at 0)
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<ldsx_cu8 *>(static_cast<uintptr_t>(byte)); };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsx_cu32 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C4 = g.C * 4u, CC4 = g.C * g.C * 4u;
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    // per-wave LDS: two text slots, then the hit queue
    const uint32_t tb = L.off_wave + wave_in_wg * L.wave_stride;   // wave-uniform
    const uint32_t ringb = tb + 2u * SLOT;
    uint32_t *cursor = reinterpret_cast<uint32_t *>(smem + tb);    // records in the wave's open chunk (the first bytes of a slot are padding)
    uint2 *__restrict__ slab = a.wq + static_cast<uint64_t>(wave_global) * a.wq_slab;
    uint32_t wq_n = 0;  // wave-uniform

    // ---- the record list: this wave's open chunk ----
    uint32_t chunk = 0;       // wave-uniform
    bool chunk_ok = false;    // wave-uniform: the chunk lies inside the list (else the records are only counted: the caller reruns with a longer list)
    auto take_chunk = [&]() {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(a.chunk_next, 1u);
        chunk = __builtin_amdgcn_readfirstlane(c);
        chunk_ok = chunk < a.chunk_cap;
    };
    // (wave-uniform places only) the open chunk is closed once it is half full: whatever is logged before the next checkpoint then fits
    auto rec_checkpoint = [&]() {
        const uint32_t n = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile ldsx_u32 *>(static_cast<uintptr_t>(tb)));
        if (n > kEmit3Chunk / 2u) {
            if (lane == 0) {
                if (chunk_ok) a.chunk_fill[chunk] = n < kEmit3Chunk ? n : kEmit3Chunk;
                *reinterpret_cast<volatile ldsx_u32 *>(static_cast<uintptr_t>(tb)) = 0u;
            }
            take_chunk();
        }
    };
    if (lane == 0) *reinterpret_cast<volatile ldsx_u32 *>(static_cast<uintptr_t>(tb)) = 0u;
    take_chunk();

    // a deep match: `p` = virtual position of its last byte, `len` its length; `copy` > 0: a further copy of a pattern registered more than once
    // `counted`: the caller has added the match to its tile's count already (one atomic per tile and batch instead of one per match:
    // device-scope atomics run at ~15 G/s on this chip whatever their addresses, profiles/r04_emit3_experiments.txt)
    auto log_deep = [&](uint32_t p, uint32_t len, uint32_t value, uint32_t copy, bool counted) {
        if (p < a.emit_from) return;
        const uint32_t slot = atomicAdd(cursor, 1u);
        if (!counted) atomicAdd(&a.tile_deep[p >> 10], 1u);
        if (slot >= kEmit3Chunk) { atomicOr(a.fail, 2u); return; }
        if (chunk_ok) a.recs[static_cast<uint64_t>(chunk) * kEmit3Chunk + slot] = uint4{p, len | (copy << 24), value, a.tile0 + (p >> 10)};
    };
    // a state that ends a pattern: its own match, then the further copies of a duplicate (erec.w / ehit4.w: count << 24)
    auto log_state = [&](uint32_t p, uint32_t len, uint32_t value, uint32_t ncopies, uint32_t state, bool counted) {
        log_deep(p, len, value, 0u, counted);
        if (ncopies != 0) {
            const uint32_t off = g.dupo[state];
            for (uint32_t k = 0; k < ncopies; ++k) log_deep(p, len, g.dupv[off + k], k + 1u, false);
        }
    };

    auto load_chunk = [&](uint32_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const e3_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const e3_u32x4_t *>(hay + v));
        uint4 r{q.x, q.y, q.z, q.w};
        if (v < a.lead || v + 16 > a.vlen) {  // first / last chunk of the window only
            uint32_t w[4] = {r.x, r.y, r.z, r.w};
            for (int b = 0; b < 16; ++b) {
                const uint32_t p = v + b;
                if (p < a.lead || p >= a.vlen) w[b >> 2] = (w[b >> 2] & ~(0xffu << (8 * (b & 3)))) | (g.unused_byte << (8 * (b & 3)));
            }
            r = uint4{w[0], w[1], w[2], w[3]};
        }
        return r;
    };
    auto raw_at = [&](uint32_t p) -> uint32_t { return (p >= a.lead && p < a.vlen) ? hay[p] : g.unused_byte; };
    auto read_ahead = [&](uint32_t v) -> unsigned long long {
        unsigned long long x;
        if (v >= a.lead && v + 8 <= a.vlen) {
            __builtin_memcpy(&x, hay + v, 8);
        } else {
            x = 0;
            for (int b = 7; b >= 0; --b) x = (x << 8) | ((v + b >= a.lead && v + b < a.vlen) ? hay[v + b] : g.unused_byte);
        }
        return x;
    };

    // walkers: {byte position p of the last byte of a (K+1)-gram, the depth-(K+2) state reached on the byte at p + 1 | class of
    // the byte at p + 2 << 27}.  What one lane stored to the slab is read back by another lane: the stores have to be out first.
    auto drain = [&]() {
        if (wq_n != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t base = 0; base < wq_n; base += 64u) {
            rec_checkpoint();
            const uint32_t i = base + lane;
            if (i < wq_n) {
                const uint2 e = slab[i];
                uint32_t vnext = e.x + 2u;   // the state consumed the byte before vnext
                uint32_t state = e.y & 0x07ffffffu;
                uint4 r = g.erec[state];     // {cmap | own, first_child, own_value, depth | further copies << 24}
                uint32_t kn = e.y >> 27;
                unsigned long long ahead = 0;
                uint32_t n_ahead = 0;
                for (;;) {
                    if (r.x & 1u) log_state(vnext - 1u, r.w & 0xffffffu, r.z, r.w >> 24, state, false);
                    if (((r.x >> kn) & 1u) == 0 || kn == 0) break;
                    state = r.y + __popc(r.x & ((1u << kn) - 2u));
                    r = g.erec[state];
                    ++vnext;
                    if (n_ahead == 0) { ahead = read_ahead(vnext); n_ahead = 8; }
                    kn = cls_of(static_cast<uint32_t>(ahead) & 0xffu);
                    ahead >>= 8;
                    --n_ahead;
                }
            }
        }
        wq_n = 0;
    };

    // ---- the hit queue: entry = LDS address of the hit byte in one of the wave's two text slots (gram3_kernels.hip) ----
    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running; entries live at (index & (kRingE3 - 1))
    uint32_t posbias0 = 0, posbias1 = 0;  // per slot: (virtual position of a byte) - (its LDS address)
    uint4 pend = uint4{0u, 0u, 0u, 0u};  // hit record read for the previous batch, not yet consumed; zero for idle lanes
                                         // {cmap | own, own_value, first_child, further copies << 24}
    uint32_t pend_pos = 0, pend_k = 0, pend_rank = 0;  // position of the hit byte; classes of the two bytes behind it (k1 | k2 << 8); rank of the hit's state
    bool pend_valid = false;           // wave-uniform
    uint32_t acc_tile = 0xffffffffu, acc_cnt = 0;   // (lanes 0 .. 3: see consume_pending)
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;
        {   // the batch's own matches: counted per tile by one lane each (a batch comes from one or two steps: a handful of tiles)
            const bool own = (r.x & 1u) != 0 && pend_pos >= a.emit_from;
            const uint32_t my_tile = pend_pos >> 10;
            unsigned long long om = __ballot(own);
            while (om != 0) {
                const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(om));
                const uint32_t t0 = __builtin_amdgcn_readlane(my_tile, leader);
                const unsigned long long same = __ballot(own && my_tile == t0);
                // lanes 0 .. 3 of acc_tile / acc_cnt are a direct-mapped cache of tile counts (a step's batches keep hitting the same two
                // or three tiles): the count goes to memory when another tile takes the entry — one atomic per tile and step, not per batch
                if (lane == (t0 & 3u)) {
                    if (acc_tile != t0) {
                        if (acc_cnt != 0) atomicAdd(&a.tile_deep[acc_tile], acc_cnt);
                        acc_tile = t0;
                        acc_cnt = 0;
                    }
                    acc_cnt += static_cast<uint32_t>(__popcll(same));
                }
                om &= ~same;
            }
        }
        if (r.x & 1u) log_state(pend_pos, K + 1, r.y, r.w >> 24, g.level_start + pend_rank, true);
        const uint32_t k1 = pend_k & 0xffu;  // (class 0: bit 0 is not an edge)
        const bool go = k1 != 0 && ((r.x >> k1) & 1u);
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go)
                (slab + wq_n)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] =
                    uint2{pend_pos, (r.z + __popc(r.x & ((1u << k1) - 2u))) | ((pend_k >> 8) << 27)};
            wq_n += static_cast<uint32_t>(__popcll(m));
        }
    };
    auto process_batch = [&](uint32_t n) {  // n <= 64 entries from the head of the queue
        __builtin_amdgcn_s_setprio(2);
        rec_checkpoint();
        consume_pending();
        pend = uint4{0u, 0u, 0u, 0u};
        pend_k = 0;
        if (lane < n) {
            const uint32_t e = lds_u32(ringb + (((q_head + lane) & (kRingE3 - 1u)) << 2));
            pend_pos = e + ((e - tb) >= SLOT ? posbias1 : posbias0);
            const uint32_t t3 = e - 3u;
            const uint32_t a0 = t3 & ~3u, sh = t3 & 3u;
            // the dwords around the hit byte (slots are self-contained: never outside [slot + 12, slot + SLOT))
            const uint32_t d0 = lds_u32(a0), d1 = lds_u32(a0 + 4u), d2 = lds_u32(a0 + 8u);
            const uint32_t x_lo = __builtin_amdgcn_alignbyte(d1, d0, sh);  // bytes p-3 .. p
            const uint32_t x_hi = __builtin_amdgcn_alignbyte(d2, d1, sh);  // bytes p+1 .. p+4
            const uint32_t c0 = cls_of(x_lo & 0xffu), c1 = cls_of((x_lo >> 8) & 0xffu), c2 = cls_of((x_lo >> 16) & 0xffu), d = cls_of(x_lo >> 24);
            const uint32_t k1 = cls_of(x_hi & 0xffu), k2 = cls_of((x_hi >> 8) & 0xffu);
            pend_k = k1 | (k2 << 8);
            uint32_t am = (c2 << 2) + offM;
            am = __umul24(c1, C4) + am;
            if (K == 3) am = __umul24(c0, CC4) + am; else (void)c0;
            // rank of continuation bit d of that word among all set bits = offset of the depth-(K+1) state
            const uint32_t rel = am - offM, grp = offM + (rel & ~15u);
            const uint32_t own = lds_u32(am), qx = lds_u32(grp), qy = lds_u32(grp + 4u), qz = lds_u32(grp + 8u);
            const uint32_t idx = (rel >> 2) & 3u;
            const uint32_t base = S16 ? *reinterpret_cast<ldsx_cu16 *>(static_cast<uintptr_t>(offS + ((rel >> 4) << 1)))
                                      : *reinterpret_cast<ldsx_cu32 *>(static_cast<uintptr_t>(offS + ((rel >> 4) << 2)));
            uint32_t below = __popc(own & kContBitsE3 & ((1u << d) - 1u));
            below += idx > 0 ? __popc(qx & kContBitsE3) : 0u;
            below += idx > 1 ? __popc(qy & kContBitsE3) : 0u;
            below += idx > 2 ? __popc(qz & kContBitsE3) : 0u;
            pend_rank = base + below;
            pend = g.ehit4[pend_rank];
        }
        q_head += n;
        pend_valid = true;
    };

    uint32_t sl = 0;          // slot of the current step (wave-uniform)
    uint32_t carry_in = 0;    // queued entries that belong to the step before the current one
    for (uint32_t region = wave_global; region < a.nregions; region += nwaves) {
        const uint32_t rbase = region * a.region_bytes;
        const uint32_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        // classes of the K bytes before the region, oldest in the low byte; the four raw bytes before it
        uint32_t carry = 0, tail4 = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (rbase >= static_cast<uint32_t>(K - i) ? cls_of(raw_at(rbase - (K - i))) : 0u) << (8 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) tail4 |= (rbase >= static_cast<uint32_t>(4 - i) ? raw_at(rbase - (4 - i)) : static_cast<uint32_t>(g.unused_byte)) << (8 * i);
        tail4 = __builtin_amdgcn_readfirstlane(tail4);
        uint32_t mcarry;  // ME word of the K-gram ending just before the region
        {
            uint32_t x = (((carry >> (8 * (K - 1))) & 0xffu) << 2) + offM;
            x += __umul24((carry >> (8 * (K - 2))) & 0xffu, C4);
            if (K == 3) x += __umul24(carry & 0xffu, CC4);
            mcarry = __builtin_amdgcn_readfirstlane(lds_u32(x));
        }

        // the chunks of the step at s0; past the region's end only lane 0's first chunk (it feeds the last step's trailer)
        auto fetch = [&](uint32_t s0, uint4 (&out)[Q]) {
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

        for (uint32_t sb = rbase; sb < rend; sb += SB) {
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint32_t v = sb + lane * P;
            uint4 cur[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) { cur[q] = pf0[q]; pf0[q] = pf1[q]; }
            rec_checkpoint();
            consume_pending();  // before the next chunk is requested: loads retire in order
            __builtin_amdgcn_s_setprio(0);
            fetch(sb + 2u * SB, pf1);

            // ---- this step's text into its slot (whatever was queued from the step before last has been consumed) ----
            const uint32_t slot = tb + sl * SLOT;                 // wave-uniform
            const uint32_t my_text = slot + 16u + lane * P;       // LDS address of this lane's first byte
            {
                const uint32_t bias = sb - (slot + 16u);
                if (sl) posbias1 = bias; else posbias0 = bias;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                *reinterpret_cast<ldsx_u32x4 *>(static_cast<uintptr_t>(my_text + 16u * q)) = e3_u32x4_t{cur[q].x, cur[q].y, cur[q].z, cur[q].w};
            if (lane == 0) {
                *reinterpret_cast<ldsx_u32 *>(static_cast<uintptr_t>(slot + 12u)) = tail4;
                *reinterpret_cast<ldsx_u32x4 *>(static_cast<uintptr_t>(slot + 16u + SB)) = e3_u32x4_t{pf0[0].x, pf0[0].y, pf0[0].z, pf0[0].w};
            }
            tail4 = __builtin_amdgcn_readlane(cur[Q - 1].w, 63);

            // ---- byte classes of this lane's P positions plus K to the left ----
            uint32_t kx[K + P];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t w[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    kx[K + 16 * q + b] = pinx(cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu));
                    __builtin_assume(kx[K + 16 * q + b] < 32u);
                }
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);  // this lane's last K classes, oldest low
            const uint32_t left = wave_shr1_x(pk, carry);
            carry = __builtin_amdgcn_readlane(pk, 63);
#pragma unroll
            for (int i = 0; i < K; ++i) { kx[i] = (left >> (8 * i)) & 0xffu; __builtin_assume(kx[i] < 32u); }

            // ---- ME words of the K-grams ending at j = 0 .. P-1: hit bits into the lane's mask, flag bits + class into the stream ----
            uint32_t H = 0, mprev = 0;
            uint32_t annw[P / 4];
#pragma unroll
            for (int grp = 0; grp < P / 8; ++grp) {
                uint32_t mw[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = grp * 8 + jj;
                    uint32_t x = pinx((kx[K + j] << 2) + offM);                       // 4 c_j + offM              (v_lshl_add_u32)
                    x = __umul24(kx[K + j - 1], C4) + x;                              // + 4 C c_(j-1)             (v_mad_u32_u24)
                    if (K == 3) x = __umul24(kx[K + j - 2], CC4) + pinx(x);           // + 4 C^2 c_(j-2)           (v_mad_u32_u24)
                    mw[jj] = lds_u32(x);
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = grp * 8 + jj;
                    if (j > 0) H |= __builtin_amdgcn_ubfe(jj == 0 ? mprev : mw[jj - 1], kx[K + j], 1) << j;
                }
                mprev = mw[7];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j0 = grp * 8 + 4 * h;
                    // the top bytes of four words (flags in bits 5-7 of each) | the four classes
                    const uint32_t t01 = __builtin_amdgcn_perm(mw[4 * h + 1], mw[4 * h], 0x0c0c0703u);
                    const uint32_t t23 = __builtin_amdgcn_perm(mw[4 * h + 3], mw[4 * h + 2], 0x07030c0cu);
                    const uint32_t cpk = kx[K + j0] | (kx[K + j0 + 1] << 8) | (kx[K + j0 + 2] << 16) | (kx[K + j0 + 3] << 24);
                    annw[grp * 2 + h] = ((t01 | t23) & 0xe0e0e0e0u) | cpk;
                }
            }
            {   // position 0 against the ME word of the K-gram ending just before this lane's share
                const uint32_t mleft = wave_shr1_x(mprev, mcarry);
                mcarry = __builtin_amdgcn_readlane(mprev, 63);
                H |= __builtin_amdgcn_ubfe(mleft, kx[K], 1);
            }
            if (sb < a.emit_from) {   // (the first step(s) of a window: ends before emit_from belong to the window before)
#pragma unroll
                for (int i = 0; i < P / 4; ++i) {
                    const uint32_t p0 = v + 4u * i;
                    const uint32_t nb = a.emit_from > p0 ? (a.emit_from - p0 < 4u ? a.emit_from - p0 : 4u) : 0u;
                    annw[i] &= nb >= 4u ? 0x1f1f1f1fu : ~(0xe0e0e0e0u & ((1u << (8u * nb)) - 1u));
                }
            }
            uint32_t nshort = 0;
#pragma unroll
            for (int i = 0; i < P / 4; ++i) nshort += __popc(annw[i] & 0xe0e0e0e0u);
            {
                uint4 *dst = reinterpret_cast<uint4 *>(a.ann + v);
                {
                dst[0] = uint4{annw[0], annw[1], annw[2], annw[3]};
                dst[1] = uint4{annw[4], annw[5], annw[6], annw[7]};
                }
                const uint32_t incl = wave_incl_scan_x(nshort);
                const uint32_t s31 = __builtin_amdgcn_readlane(incl, 31), s63 = __builtin_amdgcn_readlane(incl, 63);
                if (lane == 0) {
                    a.tile_short[sb >> 10] = s31;
                    a.tile_short[(sb >> 10) + 1u] = s63 - s31;
                }
            }

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
                    *reinterpret_cast<ldsx_u32 *>(static_cast<uintptr_t>(ringb + ((at & (kRingE3 - 1u)) << 2))) = my_text + b;
                }
                q_tail += static_cast<uint32_t>(__popcll(m));
                if (q_tail - q_head >= 64u) { process_batch(64u); did_batch = true; }
            }
            // whatever was queued a step ago must be gone before its slot is written again
            if (carry_in != 0 && !did_batch) process_batch(q_tail - q_head);
            carry_in = q_tail - q_head;
            sl ^= 1u;
        }
    }
    if (q_tail != q_head) process_batch(q_tail - q_head);
    rec_checkpoint();
    consume_pending();
    drain();
    if (lane < 4u && acc_cnt != 0) atomicAdd(&a.tile_deep[acc_tile], acc_cnt);
    {   // close the open chunk
        const uint32_t n = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile ldsx_u32 *>(static_cast<uintptr_t>(tb)));
        if (lane == 0 && chunk_ok) a.chunk_fill[chunk] = n < kEmit3Chunk ? n : kEmit3Chunk;
    }
}

// ================================================================================================================ host-side glue kernels
__global__ __launch_bounds__(256) void emit3_combine_kernel(const uint32_t *__restrict__ tile_short, const uint32_t *__restrict__ tile_deep,
                                                            unsigned long long *__restrict__ total, unsigned long long *__restrict__ deep, uint64_t n) {
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t s = tile_short[i], d = tile_deep[i];
        total[i] = static_cast<unsigned long long>(s) + d;
        deep[i] = d;
    }
}
// records -> the bin of the tile they end in; `cursor` = the per-tile record counts, counted down.  One workgroup per chunk: the chunk's
// records are grouped by tile in LDS (a hash of the tile numbers: a chunk holds one wave's records, a few dozen tiles), so that one
// device-scope atomic per TILE AND CHUNK takes the slots — those atomics run at ~15-30 G/s whatever their addresses, and one per run of
// equal tiles in meeting order (8 M per GiB of cfg3) was all of this kernel's 0.26 ms (profiles/r04_emit3_experiments.txt).
__global__ __launch_bounds__(256) void emit3_bin_kernel(const uint4 *__restrict__ recs, const uint32_t *__restrict__ chunk_fill, const uint32_t *__restrict__ chunk_next,
                                                        uint32_t chunk_cap, const unsigned long long *__restrict__ bin_off, uint32_t *__restrict__ cursor,
                                                        uint4 *__restrict__ binned, uint32_t n1k, unsigned long long rec_limit) {
    constexpr uint32_t H = 2048, kEmptyKey = 0xffffffffu;   // slots of the hash (a chunk holds at most 1024 records)
    constexpr uint32_t PER = kEmit3Chunk / 256;
    __shared__ uint32_t keys[H], cnt[H], base[H];
    if (*chunk_next > chunk_cap || bin_off[n1k] > rec_limit) return;   // (a list that overflowed, or one the caller has no room / use for: launched without the host having looked)
    const uint32_t used = *chunk_next;
    for (uint32_t c = blockIdx.x; c < used; c += gridDim.x) {
        const uint32_t fill = chunk_fill[c];
        for (uint32_t h = threadIdx.x; h < H; h += 256) { keys[h] = kEmptyKey; cnt[h] = 0; }
        __syncthreads();
        uint4 r[PER];
        uint32_t slot[PER], rank[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * 256u + threadIdx.x;
            slot[k] = 0; rank[k] = 0;
            if (i < fill) {
                r[k] = recs[static_cast<uint64_t>(c) * kEmit3Chunk + i];
                uint32_t h = (r[k].w * 0x9E3779B1u) >> 21;
                for (;;) {
                    const uint32_t old = atomicCAS(&keys[h], kEmptyKey, r[k].w);
                    if (old == kEmptyKey || old == r[k].w) break;
                    h = (h + 1u) & (H - 1u);
                }
                slot[k] = h;
                rank[k] = atomicAdd(&cnt[h], 1u);
            }
        }
        __syncthreads();
        for (uint32_t h = threadIdx.x; h < H; h += 256) {
            const uint32_t n = cnt[h];
            if (n != 0) base[h] = atomicSub(&cursor[keys[h]], n) - n;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * 256u + threadIdx.x;
            if (i < fill) binned[bin_off[r[k].w] + base[slot[k]] + rank[k]] = r[k];
        }
        __syncthreads();
    }
}

// ================================================================================================================ EXPAND
// One wave per tile of 1024 positions; lane l owns the 16 consecutive positions 16 l .. 16 l + 15 (one 16-byte load of the stream).
// LDS per workgroup: V1 | V2 | per wave {1024 staged tuples, 1024 x u16 length bits of the deep matches, per lane its first slot and its
// flag bytes, 64 extras, counter}.
//
// Why the tuples go through LDS (profiles/r04_emit3_experiments.txt): a store instruction whose active lanes write 16 bytes each to slots
// that are merely NEAR each other is not merged by the memory pipeline — a version that stored straight from a column-transposed loop
// (lane = position, one wave scan per 64 positions) spent 4.4 of its 6.0 ms per GiB in those stores, no better than the 64 lines per
// instruction of gram2_emit_kernels.hip.  Here every tuple of the tile is STAGED as {value, position | length << 10} at its slot
// (ds_write_b64) — the short ones by the lane that owns the position, walking its 16 positions with a running slot, the deep ones by the
// lane that holds the record — and the tile is then copied out 64 CONSECUTIVE slots per store instruction: one contiguous kilobyte.
// A tile of more than 1024 tuples (dozens of deep matches per position) takes several passes over a window of slots.
// HAS1: the dictionary has one-byte patterns (else that flag bit is never set and the walk leaves it out)
// RAW (the PFX engine's tuples, K = 1): the "stream" is the haystack itself — a position's only flag is "this byte is a one-byte pattern" (a
// 256-byte table behind V1, which is indexed by byte), every longer match is a record; tuples are staged as 8-byte entries for both formats.
template <int K, bool F16, bool HAS1, bool RAW>
__global__ __launch_bounds__((F16 && !RAW) ? 512 : 256, (F16 && !RAW) ? 4 : 3) void emit3_expand_kernel(const Gram2EmitDev g, const Expand3Args a) {
    constexpr bool ST16 = F16 && !RAW;   // the staged entries are u16 {position | length} (values looked up when the tile goes out)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    e3_copy(smem, g.v1, g.v1_bytes);
    e3_copy(smem + g.v1_bytes, g.v2, g.v2_bytes);
    if (a.v3_in_lds) e3_copy(smem + g.v1_bytes + g.v2_bytes, g.v3c, g.v3c_bytes);
    __syncthreads();
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();   // (tables are also read through absolute LDS addresses)
    const uint32_t *v1 = reinterpret_cast<const uint32_t *>(smem);
    const uint32_t *v2 = reinterpret_cast<const uint32_t *>(smem + g.v1_bytes);
    const uint32_t *v3bm = reinterpret_cast<const uint32_t *>(smem + g.v1_bytes + g.v2_bytes);
    const uint16_t *v3dir = reinterpret_cast<const uint16_t *>(smem + g.v1_bytes + g.v2_bytes + g.v3c_dir);
    const uint32_t *v3val = reinterpret_cast<const uint32_t *>(smem + g.v1_bytes + g.v2_bytes + g.v3c_val);
    const bool v3l = a.v3_in_lds != 0;   // wave-uniform
    const uint32_t v3_at = g.v1_bytes + g.v2_bytes;   // LDS address of the rank structure (the dynamic segment starts at 0: no static LDS here)
    // value of the 3-byte pattern that is the 3-gram `idx` (the caller knows it is one)
    auto v3_of = [&](uint32_t idx) -> uint32_t {
        if (v3l) {
            const uint32_t w = v3bm[idx >> 5];
            return v3val[v3dir[idx >> 5] + __popc(w & ((1u << (idx & 31u)) - 1u))];
        }
        return g.v3[idx];
    };
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const uint32_t C = g.C, CC = g.C * g.C;
    constexpr uint32_t W = kEmit3Stage;   // slots per pass
    // F24: W staged tuples of 8 bytes.  F16: W u16 entries {position | length << 10} and the tile's 16 + 1024 stream bytes (the values are
    // looked up when the tile goes out).  Then one dump entry per lane (what a lane has NOT got to write goes there: no branch).
    constexpr uint32_t SW = ST16 ? (W + 64u) * 2u + 1040u : (W + 64u) * 8u;
    constexpr uint32_t WAVE_BYTES = ST16 ? kEmit3ExpandWave16 : kEmit3ExpandWave;
    char *wl = smem + a.off_wave + wave_in_wg * WAVE_BYTES;
    uint2 *stage = reinterpret_cast<uint2 *>(wl);                               // F24: W x {value, position in tile | length << 10}
    uint16_t *stage16 = reinterpret_cast<uint16_t *>(wl);                       // F16: W x {position in tile | length << 10} (0: a deep match's slot)
    uint8_t *annb = reinterpret_cast<uint8_t *>(wl + (W + 64u) * 2u);           // F16: [0,16) the stream bytes before the tile | the tile
    uint32_t *dm32 = reinterpret_cast<uint32_t *>(wl + SW);                     // per position a u16: bit (len - K - 1) per deep match
    uint16_t *dm16 = reinterpret_cast<uint16_t *>(wl + SW);
    uint32_t *lbase = reinterpret_cast<uint32_t *>(wl + SW + 2048);             // per lane: tile-relative slot of its first tuple
    uint4 *lflags = reinterpret_cast<uint4 *>(wl + SW + 2048 + 256);            // per lane: the flag bits of its 16 stream bytes
    uint4 *xs = reinterpret_cast<uint4 *>(wl + SW + 2048 + 256 + 1024);         // the tile's extras {position in tile, length | copy << 24, value, -}
    uint32_t *ctr = reinterpret_cast<uint32_t *>(wl + SW + 2048 + 256 + 1024 + kEmit3MaxExtras * 16);
    const uint32_t dump = W + lane;
    bool dirty = true;  // wave-uniform: dm holds bits of the previous tile

    // A tile's inputs are asked for AHEAD, and everything a tile reads has arrived before its first tuple is stored: loads retire in order
    // with the stores (one vmcnt for both), and a wave that asked for anything after ten kilobytes of its own stores waited for the stores
    // (profiles/r04_emit3_experiments.txt).  Two levels, because the records of a tile sit where its offsets say: the offsets of tile k + 2
    // and the stream bytes + first 64 records of tile k + 1 go out at the top of tile k.
    struct TileOff { unsigned long long tile_base, tile_end, bin0, bin1; };       // as loaded (per lane, all lanes the same)
    struct TileS { unsigned long long tile_base, bin0; uint32_t tile_n, n; };      // wave-uniform
    struct TileIn { uint4 annq; uint32_t prev2; uint4 rec0; };
    auto ask_off = [&](uint32_t t) -> TileOff {   // (no branches: a request under a condition is copied, and waited for, where the branches meet)
        t = t < a.ntiles ? t : a.ntiles - 1u;
        TileOff x;
        x.tile_base = a.tile_off[t]; x.tile_end = a.tile_off[t + 1];
        x.bin0 = a.bin_off[t]; x.bin1 = a.bin_off[t + 1];
        return x;
    };
    auto uniform64 = [&](unsigned long long v) -> unsigned long long {
        // (the builtin returns int: the low half goes through uint32_t, or a tuple index of 2^31 and more arrives sign-extended)
        return (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32)))) << 32) |
               static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v))));
    };
    auto to_s = [&](const TileOff &o) -> TileS {
        TileS x;
        x.tile_base = uniform64(o.tile_base); x.bin0 = uniform64(o.bin0);
        x.tile_n = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(o.tile_end - o.tile_base));
        x.n = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(o.bin1 - o.bin0));
        return x;
    };
    auto ask_in = [&](uint32_t t, const TileS &sx) -> TileIn {
        t = t < a.ntiles ? t : a.ntiles - 1u;
        const uint32_t v0 = t * kEmit3Tile;
        TileIn x;
        if (RAW) {   // (the haystack ends where it ends: a chunk beyond it is asked for at the last one's address and masked below)
            const uint32_t va = v0 + lane * 16u, last = (a.vlen - 1u) & ~15u;
            x.annq = *reinterpret_cast<const uint4 *>(a.ann + (va < last ? va : last));
            x.prev2 = 0u;
        } else {
        x.annq = *reinterpret_cast<const uint4 *>(a.ann + v0 + lane * 16u);
        x.prev2 = *reinterpret_cast<const uint16_t *>(a.ann + (v0 != 0 ? v0 - 2u : 0u));   // the two stream bytes before the tile (tile 0: not used)
        }
        x.rec0 = a.binned[sx.bin0 + (lane < sx.n ? lane : 0u)];   // (the list ends with one spare record: n may be 0 at its very end)
        return x;
    };
    const uint32_t ctr_at = a.off_wave + wave_in_wg * WAVE_BYTES + SW + 2048 + 256 + 1024 + kEmit3MaxExtras * 16;   // LDS address of `ctr`
    // (named sets of input registers, the loop unrolled by hand: with one set the compiler copied the freshly requested registers at the
    // loop's edge and waited for the request there)
    // (ts = this tile's offsets, ts1 = the next tile's (in: known; out: those of the tile after it), in / in_next = stream bytes and records)
    auto do_tile = [&](const uint32_t t, const TileS &ts, TileS &ts1, const TileIn &in, TileIn &in_next) {
        const uint32_t v0 = t * kEmit3Tile;
        const uint4 annq = in.annq;
        const uint32_t prev2 = t > 0 ? in.prev2 : 0u;
        const unsigned long long tile_base = ts.tile_base;
        const uint32_t tile_n = ts.tile_n;
        const unsigned long long bin0 = ts.bin0;
        const uint32_t n = ts.n;
        const bool deep = n != 0;
        const uint4 rec0 = in.rec0;
        in_next = ask_in(t + nwaves, ts1);
        const TileOff off2 = ask_off(t + 2u * nwaves);
        // everything asked for has arrived: the offsets go to scalar registers here, so that no register still counts as "being loaded"
        // when the stores go out (the compiler would wait for it — and with it for the stores — at the top of the next tile)
        auto arrived = [&]() {
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
            ts1 = to_s(off2);
        };
        if (tile_n == 0) { arrived(); return; }
        char *__restrict__ out = reinterpret_cast<char *>(a.out) + tile_base * (F16 ? 16ull : 24ull);
        const unsigned long long end0 = a.pos_base + v0;   // end of a match whose last byte is the tile's position 0
        uint32_t aw[4] = {annq.x, annq.y, annq.z, annq.w};
        const uint32_t rawq[4] = {annq.x, annq.y, annq.z, annq.w};   // RAW: the haystack bytes themselves
        if (RAW) {   // flag bytes from the table of one-byte patterns; positions outside [emit_from, vlen) have none
            const uint32_t p0 = v0 + lane * 16u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t f = 0;
                if (HAS1) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t p = p0 + 4u * k + b;
                        const uint32_t fl = *reinterpret_cast<ldsx_cu8 *>(static_cast<uintptr_t>(1024u + ((rawq[k] >> (8 * b)) & 0xffu)));
                        f |= ((p >= a.emit_from && p < a.vlen) ? fl : 0u) << (8 * b);
                    }
                }
                aw[k] = f;
            }
        }
        // classes of positions -2 .. 15 of this lane (5 bits each), flags of positions 0 .. 15 (3 bits each)
        const uint32_t left = wave_shr1_x(annq.w >> 16, prev2);
        if (ST16) {
            *reinterpret_cast<uint4 *>(annb + 16u + lane * 16u) = annq;
            if (lane == 0) *reinterpret_cast<uint16_t *>(annb + 14u) = static_cast<uint16_t>(prev2);
        }
        auto cls_at = [&](int j) -> uint32_t {   // j = -2 .. 15
            return j < 0 ? (left >> (8 * (j + 2))) & 31u : (aw[j >> 2] >> (8 * (j & 3))) & 31u;
        };
        auto flags_at = [&](int j) -> uint32_t { return (aw[j >> 2] >> (8 * (j & 3) + 5)) & 7u; };

        // ---- the tile's deep matches: a length bit per position; extras listed ----
        uint32_t xn = 0;
        if (deep || dirty) {
#pragma unroll
            for (int q = 0; q < 2; ++q) reinterpret_cast<uint4 *>(dm32)[lane + 64 * q] = uint4{0u, 0u, 0u, 0u};
        }
        dirty = deep;
        uint4 ex = uint4{0xffffffffu, 0u, 0u, 0u};   // this lane's extra
        if (deep) {
            if (lane == 0) *reinterpret_cast<ldsx_u32 *>(static_cast<uintptr_t>(ctr_at)) = 0u;
            for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                const uint32_t i = i0 + lane;
                if (i < n) {
                    const uint4 r = i0 == 0 ? rec0 : a.binned[bin0 + i];
                    const uint32_t p = (r.x - v0) & (kEmit3Tile - 1u), len = r.y & 0xffffffu, copy = r.y >> 24, lb = len - (K + 1);
                    if (lb < 16u && copy == 0u) {
                        atomicOr(&dm32[p >> 1], 1u << (lb + 16u * (p & 1u)));
                    } else {
                        const uint32_t idx = atomicAdd(ctr, 1u);
                        if (idx < kEmit3MaxExtras) xs[idx] = uint4{p, r.y, r.z, 0u};
                    }
                }
            }
            xn = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile ldsx_u32 *>(static_cast<uintptr_t>(ctr_at)));
            if (xn > kEmit3MaxExtras) {   // left to the other engines (the caller looks at the flag before it hands anything out)
                if (lane == 0) atomicOr(a.fail, 4u);
                arrived();
                return;
            }
            if (lane < xn) ex = xs[lane];
        }

        // ---- tuples per lane, one wave scan per tile ----
        uint32_t dmw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // this lane's 16 u16 of length bits
        unsigned long long xin = 0;                           // extras on this lane's 16 positions, four bits each
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt += __popc(aw[k] & 0xe0e0e0e0u);
        if (deep) {
            const uint4 d0 = reinterpret_cast<const uint4 *>(dm32)[lane * 2], d1 = reinterpret_cast<const uint4 *>(dm32)[lane * 2 + 1];
            dmw[0] = d0.x; dmw[1] = d0.y; dmw[2] = d0.z; dmw[3] = d0.w; dmw[4] = d1.x; dmw[5] = d1.y; dmw[6] = d1.z; dmw[7] = d1.w;
#pragma unroll
            for (int k = 0; k < 8; ++k) cnt += __popc(dmw[k]);
            for (uint32_t k = 0; k < xn; ++k) {
                const uint32_t ep = __builtin_amdgcn_readlane(ex.x, k);
                if ((ep >> 4) == lane) {
                    if (((xin >> (4u * (ep & 15u))) & 15u) == 15u) atomicOr(a.fail, 4u);
                    xin += 1ull << (4u * (ep & 15u));
                    ++cnt;
                }
            }
        }
        const uint32_t incl = wave_incl_scan_x(cnt);
        const uint32_t lanebase = incl - cnt;
        if (__builtin_amdgcn_readlane(incl, 63) != tile_n) {   // (DETECT's count of this tile and EXPAND's differ: a bug)
            if (lane == 0) atomicOr(a.fail, 16u);
            arrived();
            return;
        }
        if (deep) {
            lbase[lane] = lanebase;
            lflags[lane] = uint4{aw[0] & 0xe0e0e0e0u, aw[1] & 0xe0e0e0e0u, aw[2] & 0xe0e0e0e0u, aw[3] & 0xe0e0e0e0u};
        }
        uint32_t wbase = 0;
        do {   // one pass unless the tile holds more than W tuples (do-while: the wait in front of the stores is on every path out of the tile)
            // ---- the short matches: each lane walks its 16 positions; the deep ones of a position come first ----
            // (branch-free: a tuple this pass does not take — flag not set, slot outside the window — is written to the lane's dump entry;
            // an LDS store costs the same with any number of active lanes, and sixteen positions x three kinds of exec masks did not fit
            // the scalar registers.  A 3-byte pattern is staged with its 3-gram: its value, from L2, is looked up when the tile goes out.)
            // (Four positions — one stream dword — per turn of a loop that is NOT unrolled: unrolled sixteen-fold the compiler hoisted every
            // table lookup to the top and spilled a hundred registers.)
            uint32_t run = lanebase - wbase;
            if (ST16) {
                // (F16: an entry is {position | length << 10}; per kind of a position: the flag as a mask (v_bfe_i32), the slot or nothing
                // (v_bfi), the dump entry for nothing (v_min), the address, the entry, the store, the count)
                uint32_t q0 = aw[0], q1 = aw[1], q2 = aw[2], q3 = aw[3];
                uint32_t e0 = dmw[0], e1 = dmw[1], e2 = dmw[2], e3 = dmw[3], e4 = dmw[4], e5 = dmw[5], e6 = dmw[6], e7 = dmw[7];
                unsigned long long xq = xin;
                uint32_t pj = lane * 16u;
#pragma unroll 1
                for (int grp = 0; grp < 4; ++grp) {
                    const uint32_t dd[2] = {e0, e1};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (deep) run += __popc((dd[j >> 1] >> (16 * (j & 1))) & 0xffffu) + static_cast<uint32_t>((xq >> (4 * j)) & 15u);
                        if (K == 3) {
                            const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(q0), 8 * j + 7, 1));   // all ones: a 3-byte pattern ends here
                            stage16[min((run & m) | ~m, dump)] = static_cast<uint16_t>(pj | (j | (3u << 10)));
                            run -= m;
                        }
                        {
                            const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(q0), 8 * j + 6, 1));
                            stage16[min((run & m) | ~m, dump)] = static_cast<uint16_t>(pj | (j | (2u << 10)));
                            run -= m;
                        }
                        if (HAS1) {
                            const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(q0), 8 * j + 5, 1));
                            stage16[min((run & m) | ~m, dump)] = static_cast<uint16_t>(pj | (j | (1u << 10)));
                            run -= m;
                        }
                    }
                    q0 = q1; q1 = q2; q2 = q3;
                    e0 = e2; e1 = e3; e2 = e4; e3 = e5; e4 = e6; e5 = e7;
                    xq >>= 16;
                    pj += 4u;
                }
            } else {
                uint32_t prevw = left << 16;                  // the two classes before the lane's first position, in bytes 2 and 3
                uint32_t q0 = aw[0], q1 = aw[1], q2 = aw[2], q3 = aw[3];
                uint32_t r0 = rawq[0], r1 = rawq[1], r2 = rawq[2], r3 = rawq[3];
                uint32_t e0 = dmw[0], e1 = dmw[1], e2 = dmw[2], e3 = dmw[3], e4 = dmw[4], e5 = dmw[5], e6 = dmw[6], e7 = dmw[7];
                unsigned long long xq = xin;
                uint32_t pj = lane * 16u;
#pragma unroll 1
                for (int grp = 0; grp < 4; ++grp) {
                    const uint32_t dd[2] = {e0, e1};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t byte_j = (q0 >> (8 * j)) & 0xffu;
                        const uint32_t cj = RAW ? (r0 >> (8 * j)) & 0xffu : byte_j & 31u;
                        const uint32_t cm1 = (j >= 1 ? q0 >> (8 * (j - 1)) : prevw >> 24) & 31u;
                        const uint32_t cm2 = (j >= 2 ? q0 >> (8 * (j - 2)) : prevw >> (8 * (j + 2))) & 31u;
                        if (deep) run += __popc((dd[j >> 1] >> (16 * (j & 1))) & 0xffffu) + static_cast<uint32_t>((xq >> (4 * j)) & 15u);
                        if (K == 3) {
                            const uint32_t at = min((byte_j & 0x80u) ? run : 0xffffffffu, dump);
                            stage[at] = uint2{cm2 * CC + cm1 * C + cj, (pj + j) | (3u << 10)};
                            run += byte_j >> 7;
                        }
                        if (K >= 2) {
                            const uint32_t at = min((byte_j & 0x40u) ? run : 0xffffffffu, dump);
                            stage[at] = uint2{v2[cm1 * C + cj], (pj + j) | (2u << 10)};
                            run += (byte_j >> 6) & 1u;
                        }
                        if (HAS1) {
                            const uint32_t at = min((byte_j & 0x20u) ? run : 0xffffffffu, dump);
                            stage[at] = uint2{v1[cj], (pj + j) | (1u << 10)};
                            run += (byte_j >> 5) & 1u;
                        }
                    }
                    prevw = q0; q0 = q1; q1 = q2; q2 = q3;
                    r0 = r1; r1 = r2; r2 = r3;
                    e0 = e2; e1 = e3; e2 = e4; e3 = e5; e4 = e6; e5 = e7;
                    xq >>= 16;
                    pj += 4u;
                }
            }
            // everything this tile asked for at its top has arrived by now; waited for HERE, in front of the first store — the deep
            // matches below go straight to memory, and a wait behind them would be a wait for them
            if (wbase == 0) arrived(); else __builtin_amdgcn_s_waitcnt(0x0f70);
            // ---- the deep matches: first slot of the position + the longer ones at the same position ----
            if (deep) {
                // tuples of lane L that lie before its position j, extras left aside: deep ones (length bits) and short ones (flag bits)
                auto before_of = [&](uint32_t L, uint32_t j, uint32_t &here16) -> uint32_t {
                    const uint4 d0 = reinterpret_cast<const uint4 *>(dm32)[L * 2], d1 = reinterpret_cast<const uint4 *>(dm32)[L * 2 + 1];
                    const uint32_t dw[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    const uint4 fq = lflags[L];
                    const uint32_t fw[4] = {fq.x, fq.y, fq.z, fq.w};
                    uint32_t before = 0;
#pragma unroll
                    for (uint32_t w = 0; w < 8; ++w) {
                        const uint32_t keep = 2 * w + 1 < j ? 0xffffffffu : 2 * w < j ? 0xffffu : 0u;  // positions 2w, 2w+1 below j
                        before += __popc(dw[w] & keep);
                    }
#pragma unroll
                    for (uint32_t w = 0; w < 4; ++w) {
                        const uint32_t nb = j > 4 * w ? (j - 4 * w < 4u ? j - 4 * w : 4u) : 0u;       // bytes of this dword below j
                        before += __popc(fw[w] & (nb >= 4u ? 0xffffffffu : (1u << (8u * nb)) - 1u));
                    }
                    here16 = dm16[L * 16u + j];
                    return before;
                };
                // extras of the tile that come before the key: on an earlier position of the same lane, or on the same position and
                // longer / an earlier copy.  All lanes walk the list together.
                auto extras_before = [&](uint32_t p, uint32_t len, uint32_t copy) -> uint32_t {
                    uint32_t c = 0;
                    for (uint32_t k = 0; k < xn; ++k) {
                        const uint32_t ep = __builtin_amdgcn_readlane(ex.x, k), ey = __builtin_amdgcn_readlane(ex.y, k);
                        const uint32_t el = ey & 0xffffffu, ec = ey >> 24;
                        if ((ep >> 4) != (p >> 4)) continue;
                        c += (ep < p || (ep == p && (el > len || (el == len && ec < copy)))) ? 1u : 0u;
                    }
                    return c;
                };
                for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    uint4 r = rec0;
                    if (i0 != 0 && i < n) r = a.binned[bin0 + i];
                    const uint32_t p = (r.x - v0) & (kEmit3Tile - 1u), len = r.y & 0xffffffu, copy = r.y >> 24, lb = len - (K + 1);
                    const bool normal = i < n && lb < 16u && copy == 0u;
                    uint32_t slot = 0;
                    if (normal) {
                        uint32_t here16;
                        slot = lbase[p >> 4] + before_of(p >> 4, p & 15u, here16) + __popc(here16 >> (lb + 1u));
                    }
                    if (xn != 0) slot += extras_before(normal ? p : 0xfffffff0u, len, 0u);
                    slot -= wbase;
                    if (ST16) {   // a deep match goes straight to memory (one in fifty tuples of cfg3); its slot stays empty for the copy-out
                        stage16[min(normal ? slot : 0xffffffffu, dump)] = 0;
                        if (normal && slot < W) put_tuple_x<true>(out, wbase + slot, a.pos_base + r.x, len, r.z);
                    } else {
                        stage[min(normal ? slot : 0xffffffffu, dump)] = uint2{r.z, p | (len << 10)};
                    }
                }
                if (xn != 0) {  // the extras themselves, one per lane
                    const bool mine = lane < xn;
                    const uint32_t p = ex.x & (kEmit3Tile - 1u), el = ex.y & 0xffffffu, ec = ex.y >> 24, lb = el - (K + 1);
                    uint32_t slot = 0;
                    if (mine) {
                        uint32_t here16;
                        slot = lbase[p >> 4] + before_of(p >> 4, p & 15u, here16);
                        if (lb < 16u) slot += __popc(here16 >> (lb + 1u)) + ((here16 >> lb) & 1u);  // the longer ones and its own original
                    }
                    slot += extras_before(mine ? p : 0xfffffff0u, el, ec);
                    slot -= wbase;
                    if (ST16) {
                        stage16[min(mine ? slot : 0xffffffffu, dump)] = 0;
                        if (mine && slot < W) put_tuple_x<true>(out, wbase + slot, end0 + p, el, ex.z);
                    } else {
                        stage[min(mine ? slot : 0xffffffffu, dump)] = uint2{ex.z, p | (el << 10)};
                    }
                }
            }
            // ---- out: 64 consecutive slots per store instruction ----
            const uint32_t wn = tile_n - wbase < W ? tile_n - wbase : W;
            if (RAW && F16) {   // 16-byte tuples from the 8-byte entries: 64 consecutive slots per store instruction
                for (uint32_t c0 = 0; c0 < wn; c0 += 64u) {
                    const uint32_t s = c0 + lane;
                    if (s < wn) {
                        const uint2 e = stage[s];
                        put_tuple_x<true>(out, wbase + s, end0 + (e.y & 1023u), e.y >> 10, e.x);
                    }
                }
            } else if (ST16) {
                // Every slot of the window: its entry, the classes of the position and the two before it (the tile's stream bytes in LDS), the
                // value from V1 / V2 / the rank structure of V3 (LDS; V3 from L2 when the structure did not fit).  Five store instructions'
                // worth of slots at a time, each step for all five before the next step (no branch in between: the LDS round trips of the five
                // overlap; looked up slot by slot, a tile waited for forty round trips in a row).
                constexpr uint32_t NB = 4;
#pragma unroll 1
                for (uint32_t c0 = 0; c0 < wn; c0 += NB * 64u) {
                    uint32_t ent[NB], b0[NB], b1[NB], b2[NB], val[NB], w3[NB], d3[NB], i3[NB];
#pragma unroll
                    for (uint32_t k = 0; k < NB; ++k) ent[k] = stage16[c0 + k * 64u + lane];   // (past the window: stale entries, looked up and dropped)
#pragma unroll
                    for (uint32_t k = 0; k < NB; ++k) {
                        const uint32_t p = ent[k] & 1023u;
                        b0[k] = annb[16u + p]; b1[k] = annb[15u + p];
                        b2[k] = K == 3 ? annb[14u + p] : 0u;
                    }
#pragma unroll
                    for (uint32_t k = 0; k < NB; ++k) {
                        const uint32_t i2 = __umul24(b1[k] & 31u, C) + (b0[k] & 31u);
                        val[k] = *reinterpret_cast<ldsx_cu32 *>(static_cast<uintptr_t>(g.v1_bytes + i2 * 4u));
                        if (HAS1) b1[k] = *reinterpret_cast<ldsx_cu32 *>(static_cast<uintptr_t>((b0[k] & 31u) * 4u));
                        i3[k] = K == 3 ? __umul24(b2[k] & 31u, CC) + i2 : 0u;
                        if (K == 3 && v3l) {   // (absolute LDS addresses: a pointer that may be LDS or global compiles to flat loads)
                            w3[k] = *reinterpret_cast<ldsx_cu32 *>(static_cast<uintptr_t>(v3_at + (i3[k] >> 5) * 4u));
                            d3[k] = *reinterpret_cast<ldsx_cu16 *>(static_cast<uintptr_t>(v3_at + g.v3c_dir + (i3[k] >> 5) * 2u));
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < NB; ++k) {
                        const uint32_t kind = ent[k] >> 10;
                        if (HAS1) val[k] = kind == 1u ? b1[k] : val[k];
                        if (K == 3) {
                            if (v3l) {
                                const uint32_t v3v = *reinterpret_cast<ldsx_cu32 *>(static_cast<uintptr_t>(
                                    v3_at + g.v3c_val + (d3[k] + __popc(w3[k] & ((1u << (i3[k] & 31u)) - 1u))) * 4u));
                                val[k] = kind == 3u ? v3v : val[k];
                            } else if (kind == 3u && c0 + k * 64u + lane < wn) {
                                val[k] = g.v3[i3[k]];
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < NB; ++k) {
                        const uint32_t s = c0 + k * 64u + lane;
                        if (ent[k] != 0 && s < wn) put_tuple_x<true>(out, wbase + s, end0 + (ent[k] & 1023u), ent[k] >> 10, val[k]);
                    }
                }
            } else {
                // (K = 3: an entry of length 3 carries its 3-gram; the values come from L2 — every request of the window in flight before the
                // first value is written back into its staged tuple)
                if (K == 3) {
                    constexpr uint32_t NIT = W / 64u;
                    uint32_t val[NIT];
#pragma unroll
                    for (uint32_t k = 0; k < NIT; ++k) {
                        val[k] = 0;
                        if (k * 64u < wn) {
                            const uint32_t s = k * 64u + lane;
                            const uint2 e = stage[s];
                            if ((e.y >> 10) == 3u && s < wn) val[k] = v3_of(e.x);
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < NIT; ++k) {
                        if (k * 64u < wn) {
                            const uint32_t s = k * 64u + lane;
                            const uint32_t len = stage[s].y >> 10;
                            reinterpret_cast<uint32_t *>(stage)[2u * ((len == 3u && s < wn) ? s : dump)] = val[k];
                        }
                    }
                }
                // daac_match is 24 bytes: 128 slots = 192 units of 16 bytes; unit u of a block holds, by u mod 3, {start, end} of tuple 2u/3 |
                // {value, pad} of that tuple and {start} of the next | {end, value, pad} of tuple (2u + 1) / 3 ... one contiguous kilobyte per store
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const uint32_t units = (wn * 3u + 1u) / 2u;   // 16-byte units that hold the window's wn tuples (the last may be half used)
                for (uint32_t u0 = 0; u0 < units; u0 += 64u) {
                    const uint32_t u = u0 + lane;
                    if (u < units) {
                        const uint32_t m3 = u % 3u, t0 = (u / 3u) * 2u + (m3 == 2u ? 1u : 0u);   // the tuple the unit's first 8 bytes belong to
                        const uint2 e0 = stage[t0 < wn ? t0 : wn - 1u], e1 = stage[t0 + 1u < wn ? t0 + 1u : wn - 1u];
                        const unsigned long long end_a = end0 + (e0.y & 1023u), start_a = end_a - (e0.y >> 10);
                        const unsigned long long end_b = end0 + (e1.y & 1023u), start_b = end_b - (e1.y >> 10);
                        u32x4 q;
                        if (m3 == 0u) q = u32x4{static_cast<uint32_t>(start_a), static_cast<uint32_t>(start_a >> 32), static_cast<uint32_t>(end_a), static_cast<uint32_t>(end_a >> 32)};
                        else if (m3 == 1u) q = u32x4{e0.x, 0u, static_cast<uint32_t>(start_b), static_cast<uint32_t>(start_b >> 32)};
                        else q = u32x4{static_cast<uint32_t>(end_a), static_cast<uint32_t>(end_a >> 32), e0.x, 0u};
                        char *dst = out + static_cast<unsigned long long>(wbase) * 24ull + static_cast<unsigned long long>(u) * 16ull;
                        if (m3 == 1u && t0 + 1u >= wn) {   // the window's last tuple ends in the middle of this unit
                            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<u32x2 *>(dst) = u32x2{q.x, q.y};
                        } else {
                            *reinterpret_cast<u32x4 *>(dst) = q;
                        }
                    }
                }
            }
            wbase += W;
        } while (wbase < tile_n);
    };
    TileIn in_a = TileIn{}, in_b = TileIn{};
    if (a.ntiles == 0) return;
    // (staggered wave starts were measured and moved nothing: profiles/r04_emit3_experiments.txt, step 8)
    TileS ts0, ts1;
    {
        const TileOff o0 = ask_off(wave_global), o1 = ask_off(wave_global + nwaves);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        ts0 = to_s(o0); ts1 = to_s(o1);
        in_a = ask_in(wave_global, ts0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    for (uint32_t t = wave_global; t < a.ntiles; t += 2u * nwaves) {
        // (do_tile(t, offsets of t, [in] offsets of t + 1 [out] of t + 2, bytes of t, [out] bytes of t + 1))
        TileS tsa = ts1;
        do_tile(t, ts0, tsa, in_a, in_b);          // tsa: offsets of t + 2 nwaves
        if (t + nwaves >= a.ntiles) break;
        TileS tsb = tsa;
        do_tile(t + nwaves, ts1, tsb, in_b, in_a);  // tsb: offsets of t + 3 nwaves
        ts0 = tsa; ts1 = tsb;
    }
}

// ================================================================================================================ launchers
bool emit3_plan(const Gram2EmitDev &dev, uint32_t waves, uint32_t lds_limit, Gram3Lds &L) {
    L = Gram3Lds{};
    const uint32_t slot = 64u * 32u + 32u;
    L.wave_stride = 2u * slot + kRingE3 * 4u;
    L.off_s = kOffME3 + dev.m_bytes;
    L.off_wave = L.off_s + dev.s_bytes;
    L.lds_bytes = L.off_wave + waves * L.wave_stride;
    L.threads = waves * 64u;
    L.rfull = 0;
    return L.lds_bytes <= lds_limit;
}

template <int K, bool S16>
static hipError_t launch_detect_inst(const Gram2EmitDev &dev, const Emit3Args &a, const Gram3Lds &L, uint32_t blocks, hipStream_t stream) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(emit3_detect_kernel<K, S16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(L.lds_bytes));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((emit3_detect_kernel<K, S16>), dim3(blocks), dim3(L.threads), L.lds_bytes, stream, dev, a, L);
    return hipGetLastError();
}
hipError_t launch_emit3_detect(const Gram2EmitDev &dev, const Emit3Args &a, const Gram3Lds &L, uint32_t blocks, hipStream_t stream) {
    if (dev.K == 3) return dev.s16 ? launch_detect_inst<3, true>(dev, a, L, blocks, stream) : launch_detect_inst<3, false>(dev, a, L, blocks, stream);
    return dev.s16 ? launch_detect_inst<2, true>(dev, a, L, blocks, stream) : launch_detect_inst<2, false>(dev, a, L, blocks, stream);
}

hipError_t launch_emit3_combine(const uint32_t *tile_short, const uint32_t *tile_deep, unsigned long long *total, unsigned long long *deep, uint64_t n,
                                hipStream_t stream) {
    const uint32_t blocks = static_cast<uint32_t>(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
    hipLaunchKernelGGL(emit3_combine_kernel, dim3(blocks), dim3(256), 0, stream, tile_short, tile_deep, total, deep, n);
    return hipGetLastError();
}
hipError_t launch_emit3_bin(const uint4 *recs, const uint32_t *chunk_fill, const uint32_t *chunk_next, uint32_t chunk_cap, const unsigned long long *bin_off,
                            uint32_t *cursor, uint4 *binned, uint32_t n1k, unsigned long long rec_limit, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL(emit3_bin_kernel, dim3(blocks), dim3(256), 0, stream, recs, chunk_fill, chunk_next, chunk_cap, bin_off, cursor, binned, n1k, rec_limit);
    return hipGetLastError();
}

uint32_t emit3_expand_lds_bytes(const Gram2EmitDev &dev, uint32_t waves, bool f16, bool v3_in_lds) {
    return dev.v1_bytes + dev.v2_bytes + (v3_in_lds ? dev.v3c_bytes : 0u) + waves * (f16 ? kEmit3ExpandWave16 : kEmit3ExpandWave);
}

template <int K, bool F16, bool HAS1, bool RAW>
static hipError_t launch_expand_inst(const Gram2EmitDev &dev, const Expand3Args &a, uint32_t blocks, hipStream_t stream) {
    // 16-byte format: eight waves share the tables (two workgroups = sixteen waves per CU); 24-byte format and RAW: four (8-byte staged entries)
    constexpr uint32_t kWaves = (F16 && !RAW) ? 8 : 4;
    const uint32_t lds = emit3_expand_lds_bytes(dev, kWaves, F16 && !RAW, a.v3_in_lds != 0);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(emit3_expand_kernel<K, F16, HAS1, RAW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((emit3_expand_kernel<K, F16, HAS1, RAW>), dim3(blocks), dim3(kWaves * 64), lds, stream, dev, a);
    return hipGetLastError();
}
template <int K, bool F16>
static hipError_t launch_expand_k(const Gram2EmitDev &dev, const Expand3Args &a, uint32_t blocks, hipStream_t stream) {
    return a.has_len1 ? launch_expand_inst<K, F16, true, false>(dev, a, blocks, stream) : launch_expand_inst<K, F16, false, false>(dev, a, blocks, stream);
}
hipError_t launch_emit3_expand(const Gram2EmitDev &dev, const Expand3Args &a, bool f16, uint32_t blocks, hipStream_t stream) {
    if (dev.K == 3) return f16 ? launch_expand_k<3, true>(dev, a, blocks, stream) : launch_expand_k<3, false>(dev, a, blocks, stream);
    return f16 ? launch_expand_k<2, true>(dev, a, blocks, stream) : launch_expand_k<2, false>(dev, a, blocks, stream);
}
// The PFX engine's tuples: the stream is the haystack (a.ann = its 16-byte-aligned address, a.vlen / a.emit_from say which positions count), dev
// carries only V1 by byte (1024 bytes) and the 256 flag bytes behind it (v1_bytes = 1280, v2_bytes = 0)
hipError_t launch_emit3_expand_raw(const Gram2EmitDev &dev, const Expand3Args &a, bool f16, uint32_t blocks, hipStream_t stream) {
    if (f16) return a.has_len1 ? launch_expand_inst<1, true, true, true>(dev, a, blocks, stream) : launch_expand_inst<1, true, false, true>(dev, a, blocks, stream);
    return a.has_len1 ? launch_expand_inst<1, false, true, true>(dev, a, blocks, stream) : launch_expand_inst<1, false, false, true>(dev, a, blocks, stream);
}

}  // namespace daac
