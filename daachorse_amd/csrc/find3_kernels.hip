// find_iter without a state chain (gfx950): count (+ checksum) of FindIterator's match stream (reference src/bytewise/iter.rs:58-113) for
// the bytewise Standard dictionaries the tuple emitter serves, from the emitter's own front half.
//
// FindIterator restarted at r reports, at the first end e with a pattern of length <= e - r ending there, the LONGEST such pattern, and
// restarts at e.  With m(j) = the length of the shortest pattern ending at position j (none: infinity) and d(j) = the distance from j back
// to the last selected position (the restart point counts as one, just before it),
//
//     selected(j)  <=>  m(j) <= d(j),          length(j) = the longest pattern ending at j that is not longer than d(j)
//
// — a recurrence over POSITIONS whose dependencies point strictly backwards, not a walk over states.  DETECT + BIN of emit3_kernels.hip
// leave everything it needs: per position one byte `class | flags << 5` (which patterns of 1..3 bytes end here) and the deep matches
// (longer ones) as records binned by tile.  SELECT (this file) takes tiles of 2 048 positions, one wave each, a lane owning 32 consecutive
// positions as bit masks, and solves the recurrence by RELAXATION, bit-parallel:
//
//     S' = E1 | (E2 & ~S<<1) | (E3 & ~(S<<1 | S<<2)) | Sd            (Ek: the shortest pattern ending here has k bytes; shifts across lanes by DPP)
//
// iterated until no lane's word changes: position j is final once everything it looks back at is, so the iteration count is the longest
// run of positions that look back — a dozen on text, the tile's length on `aaaa...` against "aa" (the kernel gives such a tile up after
// 128 rounds and the chain walkers take the request).  Positions where ONLY deep patterns end (0.6 % of cfg3's text) need the exact distance:
// Sd = "no selection among the m - 1 positions before", re-evaluated when the short system has settled, until it stops changing.
// Tiles depend on each other through the last 32 bits of the tile before: pass A (find3_tail_kernel) solves only the LAST 128 positions of
// every tile, as if nothing had been selected in front of them, and leaves the tile's last word — right whenever the look-back chains are
// shorter than 96 positions; pass B enters every tile with A's word of the tile before, tallies, and flags a tile whose own last word
// comes out differently — then one more pass with B's words (passes are capped; the walkers stand behind).
// The tally is bit-parallel too: with A1 / A2 = "selected one / two positions before", a selected position takes its 3-byte pattern iff
// neither is set, else its 2-byte pattern iff A1 is clear, else its 1-byte pattern; the few selections a DEEP pattern fits into are settled
// one by one.  The short ones are compacted (one wave scan) into u16 entries {position | length << 11} over the spent length bits, and their
// h comes 64 matches at a time from the h tables in LDS (layout of the emitter's value tables).
//
// Roofline: one byte of the annotated stream per haystack byte (+ the records); integer / bit work only, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef __attribute__((address_space(3))) const uint32_t ldsq_cu32;
typedef __attribute__((address_space(3))) const uint16_t ldsq_cu16;
typedef __attribute__((address_space(3))) uint16_t ldsq_u16;
typedef __attribute__((address_space(3))) const uint8_t ldsq_cu8;

// lane i <- lane i - 1 of `v`; lane 0 gets `lane0`
__device__ __forceinline__ uint32_t wave_shr1_q(uint32_t v, uint32_t lane0) {
    uint32_t d = lane0;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ unsigned long long q_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t q_wave_max(uint32_t v) {   // (all lanes get it)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ void q_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
// h of the checksum definition (include/daachorse_amd.h): low32(mix64(value << 32 | length))
__device__ __forceinline__ uint32_t q_h32(uint32_t value, uint32_t length) {
    unsigned long long z = (static_cast<unsigned long long>(value) << 32) | length;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return static_cast<uint32_t>(z ^ (z >> 31));
}
// inclusive scan over the 64 lanes on the VALU (DPP row shifts + row broadcasts); all lanes must be active
__device__ __forceinline__ uint32_t q_wave_incl_scan(uint32_t x) {
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}
// bit 0 of each byte of x -> a nibble
__device__ __forceinline__ uint32_t q_nib(uint32_t x) { return ((x & 0x01010101u) * 0x01020408u) >> 24; }

}  // namespace

// One tuple of the list: daac_match16 {end, length, value} (one 16-byte store) or daac_match {start, end, value, pad}
__device__ __forceinline__ void q_put_tuple(void *out, bool f16, unsigned long long slot, unsigned long long end, uint32_t len, uint32_t value) {
    typedef uint32_t q_u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t q_u32x2 __attribute__((ext_vector_type(2)));
    if (f16) {
        *reinterpret_cast<q_u32x4 *>(static_cast<char *>(out) + slot * 16ull) = q_u32x4{static_cast<uint32_t>(end), static_cast<uint32_t>(end >> 32), len, value};
    } else {
        char *dst = static_cast<char *>(out) + slot * 24ull;
        const unsigned long long start = end - len;
        *reinterpret_cast<q_u32x4 *>(dst) = q_u32x4{static_cast<uint32_t>(start), static_cast<uint32_t>(start >> 32), static_cast<uint32_t>(end), static_cast<uint32_t>(end >> 32)};
        *reinterpret_cast<q_u32x2 *>(dst + 16) = q_u32x2{value, 0u};
    }
}

// HAS1: the dictionary has one-byte patterns; TALLY: count / checksum the selected matches (passes B ...), else leave the exit words only;
// EMIT: instead of the sums, the tuples themselves — in the iterator's order, tile t's at a.tile_off[t] — with g's tables holding the
// patterns' VALUES (the emitter's V1 / V2 / V3 rank structure) in place of their h
template <bool HAS1, bool TALLY, bool EMIT>
__global__ __launch_bounds__(1024) void find3_select_kernel(const Find3Dev g, const Find3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!find3_detect_usable(a)) return;
    if (TALLY) {
        q_copy(smem, g.h1, g.h1_bytes);
        q_copy(smem + g.h1_bytes, g.h2, g.h2_bytes);
        q_copy(smem + g.h1_bytes + g.h2_bytes, g.h3c, g.h3c_bytes);
        __syncthreads();
    }
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();   // (tables are read through absolute LDS addresses)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const uint32_t C = g.C, CC = g.C * g.C;
    const uint32_t h3_at = g.h1_bytes + g.h2_bytes;
    // per wave: [0, 16) the stream bytes before the tile | 2 048 stream bytes | 2 048 x u16 length bits of the deep matches | 64 x u32 deep selections
    char *wl = smem + a.off_wave + wave_in_wg * kFind3Wave;
    uint8_t *annb = reinterpret_cast<uint8_t *>(wl);
    uint32_t *dm32 = reinterpret_cast<uint32_t *>(wl + 2064);
    uint16_t *dm16 = reinterpret_cast<uint16_t *>(wl + 2064);
    uint16_t *stage = reinterpret_cast<uint16_t *>(wl + 2064);                 // (over the length bits, once they are spent)
    uint32_t *dlist = reinterpret_cast<uint32_t *>(wl + 2064 + 4096);           // [0]: how many; [1 ..]: deep selections {position | length << 11}
    uint32_t *dmask = reinterpret_cast<uint32_t *>(wl + 2064 + 4096 + 1024);    // per lane: its positions with deep matches
    const uint32_t ann_at = a.off_wave + wave_in_wg * kFind3Wave;   // LDS address of annb
    bool dirty = true;   // wave-uniform: dm holds bits of an earlier tile

    unsigned long long cnt = 0;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t t = wave_global; t < a.ntiles; t += nwaves) {
        const uint32_t v0 = t * kFind3Tile;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(a.ann + v0 + lane * 32u), q1 = *reinterpret_cast<const uint4 *>(a.ann + v0 + lane * 32u + 16u);
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const unsigned long long b0 = a.bin_off[2u * t], b1 = a.bin_off[(2u * t + 2u) < a.n1k ? 2u * t + 2u : a.n1k];
        const uint32_t n = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b1 - b0));
        if (TALLY) {   // (the stream bytes in LDS: the classes of a selected match's bytes)
            *reinterpret_cast<uint4 *>(annb + 16u + lane * 32u) = q0;
            *reinterpret_cast<uint4 *>(annb + 32u + lane * 32u) = q1;
            if (lane == 0) *reinterpret_cast<uint4 *>(annb) = t != 0 ? *reinterpret_cast<const uint4 *>(a.ann + v0 - 16u) : uint4{0u, 0u, 0u, 0u};
        }
        // ---- which patterns of 1 .. 3 bytes end at this lane's 32 positions ----
        uint32_t P1 = 0, P2 = 0, P3 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (HAS1) P1 |= q_nib(w[k] >> 5) << (4 * k);
            P2 |= q_nib(w[k] >> 6) << (4 * k);
            P3 |= q_nib(w[k] >> 7) << (4 * k);
        }
        // ---- the tile's deep matches: a length bit per position (lengths 4 .. 19) ----
        uint32_t D = 0;
        if (n != 0 || dirty) {
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<uint4 *>(dm32)[lane * 4 + q] = uint4{0u, 0u, 0u, 0u};
            dmask[lane] = 0u;
        }
        dirty = n != 0;
        if (n != 0) {
            for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                const uint32_t i = i0 + lane;
                if (i < n) {
                    const uint4 r = a.binned[b0 + i];
                    const uint32_t p = (r.x - v0) & (kFind3Tile - 1u), lb = (r.y & 0xffffffu) - 4u;
                    // (a further copy of a duplicate pattern is nothing to find_iter: it reports a state's first output)
                    if ((r.y >> 24) != 0u) {
                    } else if (lb < 16u) {
                        atomicOr(&dm32[p >> 1], 1u << (lb + 16u * (p & 1u)));
                        atomicOr(&dmask[p >> 5], 1u << (p & 31u));
                    } else atomicOr(a.flag, 2u);   // a pattern beyond 19 bytes: not this engine's
                }
            }
            D = dmask[lane];
        }
        const uint32_t any = P1 | P2 | P3;
        const uint32_t E1 = P1, E2 = P2 & ~P1, E3 = P3 & ~(P1 | P2), DO = D & ~any;   // DO: only deep patterns end here
        // the first four such positions of the lane with their shortest lengths {bit | length << 8} (a lane rarely has one)
        uint32_t pk[4] = {0x100u, 0x100u, 0x100u, 0x100u};   // (unused entries: length 1, an empty window)
        const uint32_t nd = __popc(DO);
        {
            uint32_t m = DO;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (m != 0) {
                    const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                    m &= m - 1u;
                    pk[k] = i | ((4u + static_cast<uint32_t>(__builtin_ctz(dm16[lane * 32u + i]))) << 8);
                }
            }
        }
        const bool wave_deep = __any(DO != 0);
        // no selection among the m - 1 positions before a deep-only position (X: this lane's word above the previous lane's)
        auto deep_eval = [&](uint32_t S, uint32_t Sp) -> uint32_t {
            const unsigned long long X = (static_cast<unsigned long long>(S) << 32) | Sp;
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = pk[k] & 31u, ml = pk[k] >> 8;
                const uint32_t win = static_cast<uint32_t>(X >> (33u + i - ml)) & ((1u << (ml - 1u)) - 1u);
                r |= (static_cast<uint32_t>(k) < nd && win == 0u) ? 1u << i : 0u;
            }
            if (nd > 4u) {   // (a lane with more than four: the rest straight from the length bits)
                uint32_t m = DO;
                for (int k = 0; k < 4; ++k) m &= m - 1u;
                while (m != 0) {
                    const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                    m &= m - 1u;
                    const uint32_t ml = 4u + static_cast<uint32_t>(__builtin_ctz(dm16[lane * 32u + i]));
                    const uint32_t win = static_cast<uint32_t>(X >> (33u + i - ml)) & ((1u << (ml - 1u)) - 1u);
                    r |= win == 0u ? 1u << i : 0u;
                }
            }
            return r;
        };
        // the restart point: a selection that is no match, just before the first position that counts
        uint32_t F = 0;
        if (a.force_pos != 0xffffffffu && (a.force_pos >> 5) == (v0 >> 5) + lane) F = 1u << (a.force_pos & 31u);
        const uint32_t entry = t == 0 ? (a.force_pos == 0xffffffffu ? 0x80000000u : 0u) : (a.entry_in ? a.entry_in[t - 1u] : 0u);
        // ---- relaxation ----
        uint32_t S = E1 | F, Sd = 0, Sp = 0;
        bool settled = false;   // wave-uniform
        for (uint32_t outer = 0; outer < 40u && !settled; ++outer) {
            bool inner_ok = false;
            for (uint32_t it = 0; it < 128u; ++it) {
                Sp = wave_shr1_q(S, entry);
                const uint32_t A1 = __builtin_amdgcn_alignbit(S, Sp, 31), A2 = __builtin_amdgcn_alignbit(S, Sp, 30);
                const uint32_t Sn = E1 | F | (E2 & ~A1) | (E3 & ~(A1 | A2)) | Sd;
                const bool moved = __any(Sn != S);
                S = Sn;
                if (!moved) { inner_ok = true; break; }
            }
            if (!inner_ok) break;
            if (!wave_deep) { settled = true; break; }
            Sp = wave_shr1_q(S, entry);
            const uint32_t Sd2 = deep_eval(S, Sp);
            if (!__any(Sd2 != Sd)) settled = true;
            Sd = Sd2;
        }
        if (!settled) {   // (text that keeps looking back: left to the chain walkers)
            if (lane == 0) atomicOr(a.flag, 4u);
            continue;
        }
        // ---- the tile's last word: what the next tile enters with ----
        {
            const uint32_t last = __builtin_amdgcn_readlane(S, 63);
            if (lane == 0) {
                if (a.entry_in && a.entry_in[t] != last) atomicOr(a.flag, 1u);
                a.exit_out[t] = last;
            }
        }
        if (!TALLY) continue;
        if (t + 2u >= a.ntiles) {   // where a window behind this one restarts
            const uint32_t hi = S != 0 ? v0 + lane * 32u + (31u - static_cast<uint32_t>(__builtin_clz(S))) + 1u : 0u;
            const uint32_t top = q_wave_max(hi);
            if (lane == 0 && top != 0) atomicMax(a.last_sel, top);
        }
        // ---- the selected matches ----
        Sp = wave_shr1_q(S, entry);
        const uint32_t A1 = __builtin_amdgcn_alignbit(S, Sp, 31), A2 = __builtin_amdgcn_alignbit(S, Sp, 30);
        // selections a deep pattern fits into (distance to the selection before >= its length): few — one by one
        uint32_t DS = 0;
        const bool any_deep_sel = __any((S & D & ~F) != 0);
        if (any_deep_sel) {
            if (lane == 0) dlist[0] = 0;
            const unsigned long long X = (static_cast<unsigned long long>(S) << 32) | Sp;
            uint32_t m = S & D & ~F;
            while (m != 0) {
                const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1u;
                const unsigned long long below = X & ((1ull << (32u + i)) - 1ull);
                const uint32_t d = below != 0 ? (32u + i) - (63u - static_cast<uint32_t>(__builtin_clzll(below))) : 64u;
                const uint32_t db = dm16[lane * 32u + i];
                const uint32_t cand = d >= 4u ? (d - 3u >= 16u ? db : db & ((1u << (d - 3u)) - 1u)) : 0u;
                if (cand != 0) {
                    DS |= 1u << i;
                    const uint32_t at = atomicAdd(&dlist[0], 1u) + 1u;
                    if (at < kFind3Deep) dlist[at] = (lane * 32u + i) | ((4u + 31u - static_cast<uint32_t>(__builtin_clz(cand))) << 11);
                }
            }
        }
        const uint32_t T = S & ~F & ~DS;                       // the short ones
        const uint32_t L3 = T & P3 & ~(A1 | A2), L2 = T & ~L3 & P2 & ~A1;
        if (!EMIT && a.tile_cnt) {   // how many tuples this tile will hold (the list's offsets are a scan over these)
            const unsigned long long tot = q_wave_sum(static_cast<unsigned long long>(__popc(T) + __popc(DS)));
            if (lane == 0) a.tile_cnt[t] = tot;
        }
        if (EMIT) {
            // ---- the tile's tuples in position order: entries {position | length << 11} (0: a deep match — its record's lane writes it) ----
            const uint32_t nsel = any_deep_sel ? __builtin_amdgcn_readfirstlane(dlist[0]) : 0u;
            if (nsel >= kFind3Deep) { if (lane == 0) atomicOr(a.flag, 4u); continue; }
            const uint32_t mine = __popc(T) + __popc(DS);
            const uint32_t incl = q_wave_incl_scan(mine);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            {
                uint32_t at = incl - mine, m = T | DS;
                while (m != 0) {
                    const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m)), bit = 1u << i;
                    m &= m - 1u;
                    const uint32_t pos = lane * 32u + i;
                    const uint32_t len = (DS & bit) ? 0u : (L3 & bit) ? 3u : (L2 & bit) ? 2u : 1u;
                    stage[at] = static_cast<uint16_t>(pos | (len << 11));
                    if (DS & bit)
                        for (uint32_t k = 1; k <= nsel; ++k)
                            if ((dlist[k] & 2047u) == pos) dlist[k] |= at << 16;
                    ++at;
                }
            }
            const unsigned long long tile_base = a.tile_off[t];
            const unsigned long long end0 = a.pos_base + v0;   // end of a match whose last byte is the tile's position 0
            auto cls_at = [&](uint32_t byte_addr) -> uint32_t { return *reinterpret_cast<ldsq_cu8 *>(static_cast<uintptr_t>(byte_addr)) & 31u; };
            for (uint32_t s0 = 0; s0 < total; s0 += 64u) {
                const bool ok = s0 + lane < total;
                const uint32_t ent = stage[ok ? s0 + lane : 0u];
                const uint32_t pos = ent & 2047u, len = ent >> 11;
                const uint32_t c0 = cls_at(ann_at + 16u + pos), c1 = cls_at(ann_at + 15u + pos), c2 = cls_at(ann_at + 14u + pos);
                const uint32_t i2 = __umul24(c1, C) + c0, i3 = __umul24(c2, CC) + i2;
                const uint32_t v1 = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(c0 * 4u));
                const uint32_t v2 = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(g.h1_bytes + i2 * 4u));
                const uint32_t wd = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(h3_at + (i3 >> 5) * 4u));
                const uint32_t dr = *reinterpret_cast<ldsq_cu16 *>(static_cast<uintptr_t>(h3_at + g.h3c_dir + (i3 >> 5) * 2u));
                const uint32_t v3 = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(h3_at + g.h3c_val + (dr + __popc(wd & ((1u << (i3 & 31u)) - 1u))) * 4u));
                const uint32_t val = len == 3u ? v3 : len == 2u ? v2 : v1;
                if (ok && len != 0u) q_put_tuple(a.out, a.f16 != 0, tile_base + s0 + lane, end0 + pos, len, val);
            }
            if (nsel != 0) {
                for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    uint4 r = uint4{0u, 0u, 0u, 0u};
                    if (i < n) r = a.binned[b0 + i];
                    const uint32_t key = ((r.x - v0) & (kFind3Tile - 1u)) | ((r.y & 0xffffffu) << 11);
                    uint32_t slot = 0xffffffffu;
                    for (uint32_t k = 1; k <= nsel; ++k) { const uint32_t d = dlist[k]; slot = (d & 0xffffu) == key ? d >> 16 : slot; }
                    if (i < n && (r.y >> 24) == 0u && (r.y & 0xffffffu) < 32u && slot != 0xffffffffu)
                        q_put_tuple(a.out, a.f16 != 0, tile_base + slot, a.pos_base + r.x, r.y & 0xffffffu, r.z);
                }
            }
            continue;
        }
        if (a.count_only) {
            cnt += __popc(T) + __popc(DS);
            continue;
        }
        // compaction, one list per length (a list's 64 matches then ask the same tables): 3-byte matches first, then 2-byte, then 1-byte
        const uint32_t L1 = T & ~L3 & ~L2;
        const uint32_t mine = __popc(T);
        const uint32_t c23 = __popc(L3) | (__popc(L2) << 16);
        const uint32_t incl23 = q_wave_incl_scan(c23);
        const uint32_t tot23 = __builtin_amdgcn_readlane(incl23, 63);
        const uint32_t tot3 = tot23 & 0xffffu, tot2 = tot23 >> 16;
        uint32_t tot1 = 0;
        {
            uint32_t at = (incl23 - c23) & 0xffffu, m = L3;
            while (m != 0) { stage[at++] = static_cast<uint16_t>(lane * 32u + static_cast<uint32_t>(__builtin_ctz(m))); m &= m - 1u; }
            at = tot3 + ((incl23 - c23) >> 16); m = L2;
            while (m != 0) { stage[at++] = static_cast<uint16_t>(lane * 32u + static_cast<uint32_t>(__builtin_ctz(m))); m &= m - 1u; }
            if (HAS1) {
                const uint32_t c1 = __popc(L1), incl1 = q_wave_incl_scan(c1);
                tot1 = __builtin_amdgcn_readlane(incl1, 63);
                at = tot3 + tot2 + (incl1 - c1); m = L1;
                while (m != 0) { stage[at++] = static_cast<uint16_t>(lane * 32u + static_cast<uint32_t>(__builtin_ctz(m))); m &= m - 1u; }
            }
        }
        const uint32_t end_tile = static_cast<uint32_t>(a.pos_base) + v0;   // low 32 bits of the end of a match on the tile's position 0
        auto cls_at = [&](uint32_t byte_addr) -> uint32_t { return *reinterpret_cast<ldsq_cu8 *>(static_cast<uintptr_t>(byte_addr)) & 31u; };
        for (uint32_t s0 = 0; s0 < tot3; s0 += 64u) {
            const bool ok = s0 + lane < tot3;
            const uint32_t pos = stage[ok ? s0 + lane : 0u];
            const uint32_t i3 = __umul24(cls_at(ann_at + 14u + pos), CC) + __umul24(cls_at(ann_at + 15u + pos), C) + cls_at(ann_at + 16u + pos);
            const uint32_t wd = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(h3_at + (i3 >> 5) * 4u));
            const uint32_t dr = *reinterpret_cast<ldsq_cu16 *>(static_cast<uintptr_t>(h3_at + g.h3c_dir + (i3 >> 5) * 2u));
            const uint32_t hv = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(h3_at + g.h3c_val + (dr + __popc(wd & ((1u << (i3 & 31u)) - 1u))) * 4u));
            const uint32_t h = ok ? hv : 0u;
            s1 += h;
            s2 += h * (end_tile + pos);
        }
        for (uint32_t s0 = 0; s0 < tot2; s0 += 64u) {
            const bool ok = s0 + lane < tot2;
            const uint32_t pos = stage[tot3 + (ok ? s0 + lane : 0u)];
            const uint32_t i2 = __umul24(cls_at(ann_at + 15u + pos), C) + cls_at(ann_at + 16u + pos);
            const uint32_t hv = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(g.h1_bytes + i2 * 4u));
            const uint32_t h = ok ? hv : 0u;
            s1 += h;
            s2 += h * (end_tile + pos);
        }
        if (HAS1) {
            for (uint32_t s0 = 0; s0 < tot1; s0 += 64u) {
                const bool ok = s0 + lane < tot1;
                const uint32_t pos = stage[tot3 + tot2 + (ok ? s0 + lane : 0u)];
                const uint32_t hv = *reinterpret_cast<ldsq_cu32 *>(static_cast<uintptr_t>(cls_at(ann_at + 16u + pos) * 4u));
                const uint32_t h = ok ? hv : 0u;
                s1 += h;
                s2 += h * (end_tile + pos);
            }
        }
        cnt += mine + __popc(DS);
        if (any_deep_sel) {   // their h needs the value: from the records (a record is selected iff the list has its {position | length})
            const uint32_t nsel = __builtin_amdgcn_readfirstlane(dlist[0]);
            if (nsel >= kFind3Deep) { if (lane == 0) atomicOr(a.flag, 4u); continue; }
            for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                const uint32_t i = i0 + lane;
                uint4 r = uint4{0u, 0u, 0u, 0u};
                if (i < n) r = a.binned[b0 + i];
                const uint32_t key = ((r.x - v0) & (kFind3Tile - 1u)) | ((r.y & 0xffffffu) << 11);
                bool sel = false;
                for (uint32_t k = 1; k <= nsel; ++k) sel = sel || dlist[k] == key;
                if (i < n && (r.y >> 24) == 0u && sel) {
                    const uint32_t h = q_h32(r.z, r.y & 0xffffffu);
                    s1 += h;
                    s2 += h * (static_cast<uint32_t>(a.pos_base) + r.x);
                }
            }
        }
    }
    if (TALLY && !EMIT) {
        const unsigned long long c = q_wave_sum(cnt), x1 = q_wave_sum(s1), x2 = q_wave_sum(s2);
        if (lane == 0 && c != 0) {
            atomicAdd(a.result, c);
            atomicAdd(a.result + 1, x1);
            atomicAdd(a.result + 2, x2);
        }
    }
}

// Pass A: the last 128 positions of every tile (four lanes a tile, sixteen tiles a wave), solved as if nothing had been selected in front
// of them; leaves each tile's last word.  Same masks, same relaxation as above, within groups of four lanes.
template <bool HAS1>
__global__ __launch_bounds__(256) void find3_tail_kernel(const Find3Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t dm_all[4][16 * 64];   // per wave, per group of four lanes: 128 x u16 length bits
    if (!find3_detect_usable(a)) return;
    const uint32_t lane = threadIdx.x & 63, li = lane & 3u, grp = lane >> 2;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    uint32_t *dm32 = &dm_all[wave_in_wg][grp * 64u];
    const uint16_t *dm16 = reinterpret_cast<const uint16_t *>(dm32);
    for (uint32_t t0 = wave_global * 16u; t0 < a.ntiles; t0 += nwaves * 16u) {
        const uint32_t t = t0 + grp;
        const bool live = t < a.ntiles;
        const uint32_t tt = live ? t : a.ntiles - 1u;
        const uint32_t tail0 = tt * kFind3Tile + (kFind3Tile - 128u);   // first position of the tile's tail
        const uint32_t p0 = tail0 + li * 32u;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(a.ann + p0), q1 = *reinterpret_cast<const uint4 *>(a.ann + p0 + 16u);
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t P1 = 0, P2 = 0, P3 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (HAS1) P1 |= q_nib(w[k] >> 5) << (4 * k);
            P2 |= q_nib(w[k] >> 6) << (4 * k);
            P3 |= q_nib(w[k] >> 7) << (4 * k);
        }
        // the deep matches that end in the tail: they sit in the bin of the tile's second half
        const uint32_t i1k = 2u * tt + 1u;
        const unsigned long long b0 = a.bin_off[i1k < a.n1k ? i1k : a.n1k], b1 = a.bin_off[i1k + 1u < a.n1k ? i1k + 1u : a.n1k];
        const uint32_t n = live ? static_cast<uint32_t>(b1 - b0) : 0u;
        *reinterpret_cast<uint4 *>(dm32 + li * 16u) = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4 *>(dm32 + li * 16u + 4u) = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4 *>(dm32 + li * 16u + 8u) = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4 *>(dm32 + li * 16u + 12u) = uint4{0u, 0u, 0u, 0u};
        for (uint32_t i = li; __any(i < n); i += 4u) {
            if (i < n) {
                const uint4 r = a.binned[b0 + i];
                const uint32_t p = r.x - tail0, lb = (r.y & 0xffffffu) - 4u;
                if (p < 128u && (r.y >> 24) == 0u) {
                    if (lb < 16u) atomicOr(&dm32[p >> 1], 1u << (lb + 16u * (p & 1u)));
                    else atomicOr(a.flag, 2u);
                }
            }
        }
        uint32_t D = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 d = *reinterpret_cast<const uint4 *>(dm32 + li * 16u + 4u * q);
            const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) D |= (((dw[k] & 0xffffu) != 0u ? 1u : 0u) | ((dw[k] >> 16) != 0u ? 2u : 0u)) << (8 * q + 2 * k);
        }
        const uint32_t any = P1 | P2 | P3;
        const uint32_t E1 = P1, E2 = P2 & ~P1, E3 = P3 & ~(P1 | P2), DO = D & ~any;
        uint32_t F = 0;
        if (a.force_pos != 0xffffffffu && (a.force_pos >> 5) == (p0 >> 5)) F = 1u << (a.force_pos & 31u);
        auto deep_eval = [&](uint32_t S, uint32_t Sp) -> uint32_t {
            const unsigned long long X = (static_cast<unsigned long long>(S) << 32) | Sp;
            uint32_t r = 0, m = DO;
            while (m != 0) {
                const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1u;
                const uint32_t ml = 4u + static_cast<uint32_t>(__builtin_ctz(dm16[li * 32u + i]));
                const uint32_t win = static_cast<uint32_t>(X >> (33u + i - ml)) & ((1u << (ml - 1u)) - 1u);
                r |= win == 0u ? 1u << i : 0u;
            }
            return r;
        };
        const bool wave_deep = __any(DO != 0);
        uint32_t S = E1 | F, Sd = 0;
        bool settled = false;
        for (uint32_t outer = 0; outer < 40u && !settled; ++outer) {
            bool inner_ok = false;
            for (uint32_t it = 0; it < 128u; ++it) {
                uint32_t Sp = wave_shr1_q(S, 0u);
                Sp = li == 0 ? 0u : Sp;
                const uint32_t A1 = __builtin_amdgcn_alignbit(S, Sp, 31), A2 = __builtin_amdgcn_alignbit(S, Sp, 30);
                const uint32_t Sn = E1 | F | (E2 & ~A1) | (E3 & ~(A1 | A2)) | Sd;
                const bool moved = __any(Sn != S);
                S = Sn;
                if (!moved) { inner_ok = true; break; }
            }
            if (!inner_ok) break;
            if (!wave_deep) { settled = true; break; }
            uint32_t Sp = wave_shr1_q(S, 0u);
            Sp = li == 0 ? 0u : Sp;
            const uint32_t Sd2 = deep_eval(S, Sp);
            if (!__any(Sd2 != Sd)) settled = true;
            Sd = Sd2;
        }
        if (!settled) {
            if (lane == 0) atomicOr(a.flag, 4u);
            continue;
        }
        if (live && li == 3u) a.exit_out[t] = S;
    }
}

uint32_t find3_lds_bytes(const Find3Dev &dev, bool tally) { return (tally ? dev.h1_bytes + dev.h2_bytes + dev.h3c_bytes : 0u) + 16u * kFind3Wave; }

hipError_t launch_find3_tail(const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream) {
    if (has_len1) hipLaunchKernelGGL(find3_tail_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(find3_tail_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <bool HAS1, bool TALLY, bool EMIT>
static hipError_t launch_find3_inst(const Find3Dev &dev, const Find3Args &a, uint32_t blocks, hipStream_t stream) {
    const uint32_t lds = find3_lds_bytes(dev, TALLY);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(find3_select_kernel<HAS1, TALLY, EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((find3_select_kernel<HAS1, TALLY, EMIT>), dim3(blocks), dim3(1024), lds, stream, dev, a);
    return hipGetLastError();
}
// (the kernel's TALLY = false form — exits only — was pass A before the tails had their own kernel; it is not instantiated any more)
hipError_t launch_find3_select(const Find3Dev &dev, const Find3Args &a, bool has_len1, bool tally, uint32_t blocks, hipStream_t stream) {
    if (!tally) return hipErrorInvalidValue;
    return has_len1 ? launch_find3_inst<true, true, false>(dev, a, blocks, stream) : launch_find3_inst<false, true, false>(dev, a, blocks, stream);
}
// the tuples of the selected matches (dev: the VALUE tables; a.tile_off / a.out / a.f16 set; a.entry_in = the verified exit words)
hipError_t launch_find3_emit(const Find3Dev &dev, const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream) {
    return has_len1 ? launch_find3_inst<true, true, true>(dev, a, blocks, stream) : launch_find3_inst<false, true, true>(dev, a, blocks, stream);
}

}  // namespace daac
