// leftmost_find_iter without a state chain (gfx950): count (+ checksum) of the leftmost iterators' match stream (reference
// src/bytewise/iter.rs:272-340, LeftmostLongest and LeftmostFirst) from the front half of the tuple emitter, run on a Standard automaton
// of the same patterns (api_upload.hip builds that shadow at upload from the handle's own trie).
//
// The leftmost iterator restarted at r reports the match with the smallest START >= r, the longest one among those that start there
// (LeftmostFirst: the earliest-registered one — the builder drops every pattern below an earlier-registered one, nfa_builder.rs:60-66, so
// among the nested survivors that start at one position the longer is the earlier: the same match), and restarts at its end.  With
// L(s) = the length of the longest pattern that starts at position s (none: 0),
//
//     selected(s)  <=>  L(s) > 0  and  no selected s' < s with s' + L(s') > s
//
// — again a recurrence over positions.  DETECT + BIN (emit3_kernels.hip) say which patterns END where: per position the flags of the
// patterns of 1 .. 3 bytes, the longer ones as records.  Seen from their starts: H_k(s) = P_k(s + k - 1) (mask shifts across lanes), the
// records scattered to a length bit per START.  One wave per tile of 2 048 starts, a lane owning 32 as bit masks, solved by relaxation:
//
//     S' = NZ & ~(C0 | (S & G2) << 1 | (S & G3) << 2 | Cd)      NZ: something starts here; Gk: something of k or more bytes does;
//                                                               C0: before the restart point / under the last match of the tile before;
//                                                               Cd: under a selected match of four or more bytes (set one by one: few)
//
// until no lane's word changes.  A tile hands the next one a single number: how many of its first positions lie under this tile's last
// match (0 .. 18).  Pass A runs every tile as if nothing reached into it, pass B enters with A's numbers, tallies, and flags a tile whose
// own number differs (then one more pass; passes are capped and the chain walkers stand behind).  Text that keeps the relaxation busy
// (`aaaa` against "aa") is given up the same way as in find3_kernels.hip.
// The tally: a selected start takes its longest pattern — deep if one starts there, else 3, 2, 1 bytes; h of the short ones from the h
// tables in LDS by the classes of the bytes FROM the start (the tile's stream bytes + 16 of the next tile), of the deep ones from their
// records.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef __attribute__((address_space(3))) const uint32_t ldsl_cu32;
typedef __attribute__((address_space(3))) const uint16_t ldsl_cu16;
typedef __attribute__((address_space(3))) const uint8_t ldsl_cu8;

// lane i <- lane i - 1 of `v`; lane 0 gets `lane0`
__device__ __forceinline__ uint32_t l_wave_shr1(uint32_t v, uint32_t lane0) {
    uint32_t d = lane0;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
// lane i <- lane i + 1 of `v`; lane 63 gets `lane63`
__device__ __forceinline__ uint32_t l_wave_shl1(uint32_t v, uint32_t lane63) {
    uint32_t d = lane63;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ unsigned long long l_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t l_wave_max(uint32_t v) {   // (all lanes get it)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ void l_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ uint32_t l_h32(uint32_t value, uint32_t length) {
    unsigned long long z = (static_cast<unsigned long long>(value) << 32) | length;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return static_cast<uint32_t>(z ^ (z >> 31));
}
__device__ __forceinline__ uint32_t l_wave_incl_scan(uint32_t x) {
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ uint32_t l_nib(uint32_t x) { return ((x & 0x01010101u) * 0x01020408u) >> 24; }

}  // namespace

// One tuple of the list: daac_match16 {end, length, value} (one 16-byte store) or daac_match {start, end, value, pad}
__device__ __forceinline__ void l_put_tuple(void *out, bool f16, unsigned long long slot, unsigned long long end, uint32_t len, uint32_t value) {
    typedef uint32_t l_u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t l_u32x2 __attribute__((ext_vector_type(2)));
    if (f16) {
        *reinterpret_cast<l_u32x4 *>(static_cast<char *>(out) + slot * 16ull) = l_u32x4{static_cast<uint32_t>(end), static_cast<uint32_t>(end >> 32), len, value};
    } else {
        char *dst = static_cast<char *>(out) + slot * 24ull;
        const unsigned long long start = end - len;
        *reinterpret_cast<l_u32x4 *>(dst) = l_u32x4{static_cast<uint32_t>(start), static_cast<uint32_t>(start >> 32), static_cast<uint32_t>(end), static_cast<uint32_t>(end >> 32)};
        *reinterpret_cast<l_u32x2 *>(dst + 16) = l_u32x2{value, 0u};
    }
}

// HAS1: the dictionary has one-byte patterns; TALLY: count / checksum the selected matches, else leave the tiles' numbers only;
// EMIT: instead of the sums, the tuples themselves in the iterator's order (tile t's at a.tile_off[t]), g's tables holding VALUES
template <bool HAS1, bool TALLY, bool EMIT>
__global__ __launch_bounds__(1024) void left3_select_kernel(const Find3Dev g, const Find3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!find3_detect_usable(a)) return;
    if (TALLY) {
        l_copy(smem, g.h1, g.h1_bytes);
        l_copy(smem + g.h1_bytes, g.h2, g.h2_bytes);
        l_copy(smem + g.h1_bytes + g.h2_bytes, g.h3c, g.h3c_bytes);
        __syncthreads();
    }
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();   // (tables are read through absolute LDS addresses)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const uint32_t C = g.C, CC = g.C * g.C;
    const uint32_t h3_at = g.h1_bytes + g.h2_bytes;
    // per wave: [0, 16) unused | 2 048 stream bytes | the 16 after them | 2 048 x u16 length bits by START (then the staged starts) | the
    // list of deep selections | per lane: its starts with deep matches | per lane (+ 1): positions under a selected deep match
    char *wl = smem + a.off_wave + wave_in_wg * kLeft3Wave;
    uint8_t *annb = reinterpret_cast<uint8_t *>(wl);
    uint32_t *dm32 = reinterpret_cast<uint32_t *>(wl + 2080);
    uint16_t *dm16 = reinterpret_cast<uint16_t *>(wl + 2080);
    uint16_t *stage = reinterpret_cast<uint16_t *>(wl + 2080);
    uint32_t *dlist = reinterpret_cast<uint32_t *>(wl + 2080 + 4096);
    uint32_t *dmask = reinterpret_cast<uint32_t *>(wl + 2080 + 4096 + 1024);
    uint32_t *covm = reinterpret_cast<uint32_t *>(wl + 2080 + 4096 + 1024 + 256);   // 65 words
    const uint32_t ann_at = a.off_wave + wave_in_wg * kLeft3Wave;   // LDS address of annb
    bool dirty = true;   // wave-uniform: dm holds bits of an earlier tile

    unsigned long long cnt = 0;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t t = wave_global; t < a.ntiles; t += nwaves) {
        const uint32_t v0 = t * kFind3Tile;
        const bool has_next = t + 1u < a.ntiles;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(a.ann + v0 + lane * 32u), q1 = *reinterpret_cast<const uint4 *>(a.ann + v0 + lane * 32u + 16u);
        const uint4 qn = has_next ? *reinterpret_cast<const uint4 *>(a.ann + v0 + kFind3Tile) : uint4{0u, 0u, 0u, 0u};   // the 16 stream bytes behind the tile
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        // deep matches that START in this tile end in it or in the first 18 positions behind it: the bins of both halves and the one after
        const unsigned long long b0 = a.bin_off[2u * t], b1 = a.bin_off[(2u * t + 3u) < a.n1k ? 2u * t + 3u : a.n1k];
        const uint32_t n = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b1 - b0));
        if (TALLY) {
            *reinterpret_cast<uint4 *>(annb + 16u + lane * 32u) = q0;
            *reinterpret_cast<uint4 *>(annb + 32u + lane * 32u) = q1;
            if (lane == 0) *reinterpret_cast<uint4 *>(annb + 16u + kFind3Tile) = qn;
        }
        // ---- which patterns of 1 .. 3 bytes END at this lane's 32 positions, then: which START there ----
        uint32_t P1 = 0, P2 = 0, P3 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (HAS1) P1 |= l_nib(w[k] >> 5) << (4 * k);
            P2 |= l_nib(w[k] >> 6) << (4 * k);
            P3 |= l_nib(w[k] >> 7) << (4 * k);
        }
        const uint32_t P2n = l_wave_shl1(P2, l_nib(qn.x >> 6)), P3n = l_wave_shl1(P3, l_nib(qn.x >> 7));
        const uint32_t H1 = P1, H2 = (P2 >> 1) | (P2n << 31), H3 = (P3 >> 2) | (P3n << 30);
        // ---- the deep matches by their START: a length bit per position (lengths 4 .. 19) ----
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
                    const uint32_t len = r.y & 0xffffffu, p = r.x + 1u - len - v0, lb = len - 4u;   // p: the start within the tile (beyond 2 047: another tile's)
                    if (p < kFind3Tile && (r.y >> 24) == 0u) {
                        if (lb < 16u) {
                            atomicOr(&dm32[p >> 1], 1u << (lb + 16u * (p & 1u)));
                            atomicOr(&dmask[p >> 5], 1u << (p & 31u));
                        } else atomicOr(a.flag, 2u);   // a pattern beyond 19 bytes: not this engine's
                    }
                }
            }
            D = dmask[lane];
        }
        const uint32_t NZ = H1 | H2 | H3 | D, G2 = H2 | H3 | D, G3 = H3 | D;
        // ---- what is covered from outside: starts before the restart point, and the first positions under the last match of the tile before ----
        uint32_t C0 = 0;
        {
            const uint32_t base = v0 + lane * 32u;
            if (a.first_start >= base + 32u) C0 = 0xffffffffu;
            else if (a.first_start > base) C0 = (1u << (a.first_start - base)) - 1u;
            if (a.last_start <= base) C0 = 0xffffffffu;                                          // starts from here on belong to the window behind this one
            else if (a.last_start < base + 32u) C0 |= ~((1u << (a.last_start - base)) - 1u);
            const uint32_t c_in = t == 0 ? 0u : (a.entry_in ? a.entry_in[t - 1u] : 0u);
            if (lane == 0) C0 |= (1u << (c_in < 31u ? c_in : 31u)) - 1u;
        }
        const bool wave_deep = __any(D != 0);
        // positions under the selected deep matches (beyond the two the shifts cover too): set one by one
        auto deep_cover = [&](uint32_t S) -> uint32_t {
            covm[lane + 1u] = 0u;
            if (lane == 0) covm[0] = 0u;
            uint32_t own = 0, m = S & D;
            while (m != 0) {
                const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1u;
                const uint32_t L = 4u + 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(dm16[lane * 32u + i])));
                const unsigned long long m64 = ((1ull << (L - 1u)) - 1ull) << (i + 1u);
                own |= static_cast<uint32_t>(m64);
                const uint32_t hi = static_cast<uint32_t>(m64 >> 32);
                if (hi != 0) atomicOr(&covm[lane + 1u], hi);
            }
            return own | covm[lane];
        };
        // ---- relaxation ----
        uint32_t S = NZ & ~C0, Cd = 0;
        bool settled = false;   // wave-uniform
        for (uint32_t outer = 0; outer < 40u && !settled; ++outer) {
            bool inner_ok = false;
            for (uint32_t it = 0; it < 128u; ++it) {
                const uint32_t SG2 = S & G2, SG3 = S & G3;
                const uint32_t p2 = l_wave_shr1(SG2, 0u), p3 = l_wave_shr1(SG3, 0u);
                const uint32_t C1 = __builtin_amdgcn_alignbit(SG2, p2, 31), C2 = __builtin_amdgcn_alignbit(SG3, p3, 30);
                const uint32_t Sn = NZ & ~(C0 | C1 | C2 | Cd);
                const bool moved = __any(Sn != S);
                S = Sn;
                if (!moved) { inner_ok = true; break; }
            }
            if (!inner_ok) break;
            if (!wave_deep) { settled = true; break; }
            const uint32_t Cd2 = deep_cover(S);
            if (!__any(Cd2 != Cd)) settled = true;
            Cd = Cd2;
        }
        if (!settled) {   // (text that keeps looking back: left to the chain walkers)
            if (lane == 0) atomicOr(a.flag, 4u);
            continue;
        }
        // ---- how far the tile's last match reaches into the next tile ----
        {
            uint32_t endrel = 0;
            if (S != 0) {
                const uint32_t i = 31u - static_cast<uint32_t>(__builtin_clz(S));
                const uint32_t bit = 1u << i;
                uint32_t L = (H3 & bit) ? 3u : (H2 & bit) ? 2u : 1u;
                if (D & bit) L = 4u + 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(dm16[lane * 32u + i])));
                endrel = lane * 32u + i + L;
            }
            const uint32_t top = l_wave_max(endrel);
            const uint32_t c_out = top > kFind3Tile ? top - kFind3Tile : 0u;
            // the window's last match ends here (virtual position behind its last byte): one of the last two tiles' business, or nobody's —
            // then the next window restarts at its own first start (one atomic per TILE on one address made this kernel twice as slow)
            if (TALLY && lane == 0 && top != 0 && t + 2u >= a.ntiles) atomicMax(a.last_sel, v0 + top);
            if (lane == 0) {
                if (a.entry_in && a.entry_in[t] != c_out) atomicOr(a.flag, 1u);
                a.exit_out[t] = c_out;
            }
        }
        if (!TALLY) continue;
        // ---- the selected matches ----
        const uint32_t DS = S & D;
        const bool any_deep_sel = __any(DS != 0);
        if (any_deep_sel) {
            if (lane == 0) dlist[0] = 0;
            uint32_t m = DS;
            while (m != 0) {
                const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1u;
                const uint32_t L = 4u + 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(dm16[lane * 32u + i])));
                const uint32_t at = atomicAdd(&dlist[0], 1u) + 1u;
                if (at < kFind3Deep) dlist[at] = (lane * 32u + i) | (L << 11);
            }
        }
        const uint32_t T = S & ~D;
        if (!EMIT && a.tile_cnt) {
            const unsigned long long tot = l_wave_sum(static_cast<unsigned long long>(__popc(S)));
            if (lane == 0) a.tile_cnt[t] = tot;
        }
        if (EMIT) {
            // ---- the tile's tuples in position order: entries {start | length << 11} (0: a deep match — its record's lane writes it) ----
            const uint32_t nsel = any_deep_sel ? __builtin_amdgcn_readfirstlane(dlist[0]) : 0u;
            if (nsel >= kFind3Deep) { if (lane == 0) atomicOr(a.flag, 4u); continue; }
            const uint32_t mine = __popc(S);
            const uint32_t incl = l_wave_incl_scan(mine);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            {
                uint32_t at = incl - mine, m = S;
                while (m != 0) {
                    const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m)), bit = 1u << i;
                    m &= m - 1u;
                    const uint32_t pos = lane * 32u + i;
                    const uint32_t len = (D & bit) ? 0u : (H3 & bit) ? 3u : (H2 & bit) ? 2u : 1u;
                    stage[at] = static_cast<uint16_t>(pos | (len << 11));
                    if (D & bit)
                        for (uint32_t k = 1; k <= nsel; ++k)
                            if ((dlist[k] & 2047u) == pos) dlist[k] |= at << 16;
                    ++at;
                }
            }
            const unsigned long long tile_base = a.tile_off[t];
            const unsigned long long end0 = a.pos_base + v0;   // end of a match whose LAST byte is the tile's position 0
            auto cls_at = [&](uint32_t byte_addr) -> uint32_t { return *reinterpret_cast<ldsl_cu8 *>(static_cast<uintptr_t>(byte_addr)) & 31u; };
            for (uint32_t s0 = 0; s0 < total; s0 += 64u) {
                const bool ok = s0 + lane < total;
                const uint32_t ent = stage[ok ? s0 + lane : 0u];
                const uint32_t pos = ent & 2047u, len = ent >> 11;
                const uint32_t c0 = cls_at(ann_at + 16u + pos), c1 = cls_at(ann_at + 17u + pos), c2 = cls_at(ann_at + 18u + pos);
                const uint32_t i2 = __umul24(c0, C) + c1, i3 = __umul24(c0, CC) + __umul24(c1, C) + c2;
                const uint32_t v1 = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(c0 * 4u));
                const uint32_t v2 = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(g.h1_bytes + i2 * 4u));
                const uint32_t wd = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(h3_at + (i3 >> 5) * 4u));
                const uint32_t dr = *reinterpret_cast<ldsl_cu16 *>(static_cast<uintptr_t>(h3_at + g.h3c_dir + (i3 >> 5) * 2u));
                const uint32_t v3 = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(h3_at + g.h3c_val + (dr + __popc(wd & ((1u << (i3 & 31u)) - 1u))) * 4u));
                const uint32_t val = len == 3u ? v3 : len == 2u ? v2 : v1;
                if (ok && len != 0u) l_put_tuple(a.out, a.f16 != 0, tile_base + s0 + lane, end0 + pos + len - 1u, len, val);
            }
            if (nsel != 0) {
                for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    uint4 r = uint4{0u, 0u, 0u, 0u};
                    if (i < n) r = a.binned[b0 + i];
                    const uint32_t len = r.y & 0xffffffu, p = r.x + 1u - len - v0;
                    const uint32_t key = p | (len << 11);
                    uint32_t slot = 0xffffffffu;
                    for (uint32_t k = 1; k <= nsel; ++k) { const uint32_t d = dlist[k]; slot = (d & 0xffffu) == key ? d >> 16 : slot; }
                    if (i < n && p < kFind3Tile && len < 32u && (r.y >> 24) == 0u && slot != 0xffffffffu)
                        l_put_tuple(a.out, a.f16 != 0, tile_base + slot, a.pos_base + r.x, len, r.z);
                }
            }
            continue;
        }
        if (a.count_only) {
            cnt += __popc(S);
            continue;
        }
        const uint32_t L3 = T & H3, L2 = T & ~H3 & H2, L1 = T & ~H3 & ~H2;
        // compaction, one list per length: 3-byte matches first, then 2-byte, then 1-byte
        const uint32_t c23 = __popc(L3) | (__popc(L2) << 16);
        const uint32_t incl23 = l_wave_incl_scan(c23);
        const uint32_t tot23 = __builtin_amdgcn_readlane(incl23, 63);
        const uint32_t tot3 = tot23 & 0xffffu, tot2 = tot23 >> 16;
        uint32_t tot1 = 0;
        {
            uint32_t at = (incl23 - c23) & 0xffffu, m = L3;
            while (m != 0) { stage[at++] = static_cast<uint16_t>(lane * 32u + static_cast<uint32_t>(__builtin_ctz(m))); m &= m - 1u; }
            at = tot3 + ((incl23 - c23) >> 16); m = L2;
            while (m != 0) { stage[at++] = static_cast<uint16_t>(lane * 32u + static_cast<uint32_t>(__builtin_ctz(m))); m &= m - 1u; }
            if (HAS1) {
                const uint32_t c1 = __popc(L1), incl1 = l_wave_incl_scan(c1);
                tot1 = __builtin_amdgcn_readlane(incl1, 63);
                at = tot3 + tot2 + (incl1 - c1); m = L1;
                while (m != 0) { stage[at++] = static_cast<uint16_t>(lane * 32u + static_cast<uint32_t>(__builtin_ctz(m))); m &= m - 1u; }
            }
        }
        const uint32_t end_tile = static_cast<uint32_t>(a.pos_base) + v0;   // low 32 bits of the end of a match whose LAST byte is the tile's position 0
        auto cls_at = [&](uint32_t byte_addr) -> uint32_t { return *reinterpret_cast<ldsl_cu8 *>(static_cast<uintptr_t>(byte_addr)) & 31u; };
        for (uint32_t s0 = 0; s0 < tot3; s0 += 64u) {
            const bool ok = s0 + lane < tot3;
            const uint32_t pos = stage[ok ? s0 + lane : 0u];
            const uint32_t i3 = __umul24(cls_at(ann_at + 16u + pos), CC) + __umul24(cls_at(ann_at + 17u + pos), C) + cls_at(ann_at + 18u + pos);
            const uint32_t wd = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(h3_at + (i3 >> 5) * 4u));
            const uint32_t dr = *reinterpret_cast<ldsl_cu16 *>(static_cast<uintptr_t>(h3_at + g.h3c_dir + (i3 >> 5) * 2u));
            const uint32_t hv = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(h3_at + g.h3c_val + (dr + __popc(wd & ((1u << (i3 & 31u)) - 1u))) * 4u));
            const uint32_t h = ok ? hv : 0u;
            s1 += h;
            s2 += h * (end_tile + pos + 2u);
        }
        for (uint32_t s0 = 0; s0 < tot2; s0 += 64u) {
            const bool ok = s0 + lane < tot2;
            const uint32_t pos = stage[tot3 + (ok ? s0 + lane : 0u)];
            const uint32_t i2 = __umul24(cls_at(ann_at + 16u + pos), C) + cls_at(ann_at + 17u + pos);
            const uint32_t hv = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(g.h1_bytes + i2 * 4u));
            const uint32_t h = ok ? hv : 0u;
            s1 += h;
            s2 += h * (end_tile + pos + 1u);
        }
        if (HAS1) {
            for (uint32_t s0 = 0; s0 < tot1; s0 += 64u) {
                const bool ok = s0 + lane < tot1;
                const uint32_t pos = stage[tot3 + tot2 + (ok ? s0 + lane : 0u)];
                const uint32_t hv = *reinterpret_cast<ldsl_cu32 *>(static_cast<uintptr_t>(cls_at(ann_at + 16u + pos) * 4u));
                const uint32_t h = ok ? hv : 0u;
                s1 += h;
                s2 += h * (end_tile + pos);
            }
        }
        cnt += __popc(S);
        if (any_deep_sel) {   // their h needs the value: from the records (a record is selected iff the list has its {start | length})
            const uint32_t nsel = __builtin_amdgcn_readfirstlane(dlist[0]);
            if (nsel >= kFind3Deep) { if (lane == 0) atomicOr(a.flag, 4u); continue; }
            for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                const uint32_t i = i0 + lane;
                uint4 r = uint4{0u, 0u, 0u, 0u};
                if (i < n) r = a.binned[b0 + i];
                const uint32_t len = r.y & 0xffffffu, p = r.x + 1u - len - v0;
                const uint32_t key = p | (len << 11);
                bool sel = false;
                for (uint32_t k = 1; k <= nsel; ++k) sel = sel || dlist[k] == key;
                if (i < n && p < kFind3Tile && (r.y >> 24) == 0u && sel) {
                    const uint32_t h = l_h32(r.z, len);
                    s1 += h;
                    s2 += h * (static_cast<uint32_t>(a.pos_base) + r.x);
                }
            }
        }
    }
    if (TALLY && !EMIT) {
        const unsigned long long c = l_wave_sum(cnt), x1 = l_wave_sum(s1), x2 = l_wave_sum(s2);
        if (lane == 0 && c != 0) {
            atomicAdd(a.result, c);
            atomicAdd(a.result + 1, x1);
            atomicAdd(a.result + 2, x2);
        }
    }
}

// Pass A: the last 128 starts of every tile (four lanes a tile, sixteen tiles a wave), solved as if nothing reached into them; leaves how far
// the tile's last match reaches into the next tile — right whenever the tile's chain of matches falls in step within its last 128 positions
// (pass B checks).  Same masks, same relaxation as above, within groups of four lanes.
template <bool HAS1>
__global__ __launch_bounds__(256) void left3_tail_kernel(const Find3Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t dm_all[4][16 * 64];   // per wave, per group of four lanes: 128 x u16 length bits by start
    __shared__ uint32_t cov_all[4][16 * 8];                                // per group: cover words of its four lanes (+ 1)
    if (!find3_detect_usable(a)) return;
    const uint32_t lane = threadIdx.x & 63, li = lane & 3u, grp = lane >> 2;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    uint32_t *dm32 = &dm_all[wave_in_wg][grp * 64u];
    const uint16_t *dm16 = reinterpret_cast<const uint16_t *>(dm32);
    uint32_t *covm = &cov_all[wave_in_wg][grp * 8u];
    for (uint32_t t0 = wave_global * 16u; t0 < a.ntiles; t0 += nwaves * 16u) {
        const uint32_t t = t0 + grp;
        const bool live = t < a.ntiles;
        const uint32_t tt = live ? t : a.ntiles - 1u;
        const uint32_t tail0 = tt * kFind3Tile + (kFind3Tile - 128u);   // first start of the tile's tail
        const uint32_t p0 = tail0 + li * 32u;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(a.ann + p0), q1 = *reinterpret_cast<const uint4 *>(a.ann + p0 + 16u);
        const uint32_t qn = tt + 1u < a.ntiles ? *reinterpret_cast<const uint32_t *>(a.ann + tail0 + 128u) : 0u;
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t P1 = 0, P2 = 0, P3 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (HAS1) P1 |= l_nib(w[k] >> 5) << (4 * k);
            P2 |= l_nib(w[k] >> 6) << (4 * k);
            P3 |= l_nib(w[k] >> 7) << (4 * k);
        }
        uint32_t P2n = l_wave_shl1(P2, 0u), P3n = l_wave_shl1(P3, 0u);
        if (li == 3u) { P2n = l_nib(qn >> 6); P3n = l_nib(qn >> 7); }
        const uint32_t H1 = P1, H2 = (P2 >> 1) | (P2n << 31), H3 = (P3 >> 2) | (P3n << 30);
        // the deep matches that start in the tail end in the tile's second half or just behind it
        const uint32_t i1k = 2u * tt + 1u;
        const unsigned long long b0 = a.bin_off[i1k < a.n1k ? i1k : a.n1k], b1 = a.bin_off[i1k + 2u < a.n1k ? i1k + 2u : a.n1k];
        const uint32_t n = live ? static_cast<uint32_t>(b1 - b0) : 0u;
        *reinterpret_cast<uint4 *>(dm32 + li * 16u) = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4 *>(dm32 + li * 16u + 4u) = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4 *>(dm32 + li * 16u + 8u) = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4 *>(dm32 + li * 16u + 12u) = uint4{0u, 0u, 0u, 0u};
        for (uint32_t i = li; __any(i < n); i += 4u) {
            if (i < n) {
                const uint4 r = a.binned[b0 + i];
                const uint32_t len = r.y & 0xffffffu, p = r.x + 1u - len - tail0, lb = len - 4u;
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
        const uint32_t NZ = H1 | H2 | H3 | D, G2 = H2 | H3 | D, G3 = H3 | D;
        uint32_t C0 = 0;
        if (a.first_start >= p0 + 32u) C0 = 0xffffffffu;
        else if (a.first_start > p0) C0 = (1u << (a.first_start - p0)) - 1u;
        if (a.last_start <= p0) C0 = 0xffffffffu;
        else if (a.last_start < p0 + 32u) C0 |= ~((1u << (a.last_start - p0)) - 1u);
        const bool wave_deep = __any(D != 0);
        auto deep_cover = [&](uint32_t S) -> uint32_t {
            covm[li + 1u] = 0u;
            if (li == 0) covm[0] = 0u;
            uint32_t own = 0, m = S & D;
            while (m != 0) {
                const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1u;
                const uint32_t L = 4u + 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(dm16[li * 32u + i])));
                const unsigned long long m64 = ((1ull << (L - 1u)) - 1ull) << (i + 1u);
                own |= static_cast<uint32_t>(m64);
                const uint32_t hi = static_cast<uint32_t>(m64 >> 32);
                if (hi != 0) atomicOr(&covm[li + 1u], hi);
            }
            return own | covm[li];
        };
        uint32_t S = NZ & ~C0, Cd = 0;
        bool settled = false;
        for (uint32_t outer = 0; outer < 40u && !settled; ++outer) {
            bool inner_ok = false;
            for (uint32_t it = 0; it < 128u; ++it) {
                const uint32_t SG2 = S & G2, SG3 = S & G3;
                uint32_t p2 = l_wave_shr1(SG2, 0u), p3 = l_wave_shr1(SG3, 0u);
                if (li == 0) { p2 = 0u; p3 = 0u; }
                const uint32_t C1 = __builtin_amdgcn_alignbit(SG2, p2, 31), C2 = __builtin_amdgcn_alignbit(SG3, p3, 30);
                const uint32_t Sn = NZ & ~(C0 | C1 | C2 | Cd);
                const bool moved = __any(Sn != S);
                S = Sn;
                if (!moved) { inner_ok = true; break; }
            }
            if (!inner_ok) break;
            if (!wave_deep) { settled = true; break; }
            const uint32_t Cd2 = deep_cover(S);
            if (!__any(Cd2 != Cd)) settled = true;
            Cd = Cd2;
        }
        if (!settled) {
            if (lane == 0) atomicOr(a.flag, 4u);
            continue;
        }
        uint32_t endrel = 0;
        if (S != 0) {
            const uint32_t i = 31u - static_cast<uint32_t>(__builtin_clz(S));
            const uint32_t bit = 1u << i;
            uint32_t L = (H3 & bit) ? 3u : (H2 & bit) ? 2u : 1u;
            if (D & bit) L = 4u + 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(dm16[li * 32u + i])));
            endrel = li * 32u + i + L;
        }
        { const uint32_t o = __shfl_xor(endrel, 1, 64); endrel = o > endrel ? o : endrel; }
        { const uint32_t o = __shfl_xor(endrel, 2, 64); endrel = o > endrel ? o : endrel; }
        if (live && li == 3u) a.exit_out[t] = endrel > 128u ? endrel - 128u : 0u;
    }
}

hipError_t launch_left3_tail(const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream) {
    if (has_len1) hipLaunchKernelGGL(left3_tail_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(left3_tail_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

uint32_t left3_lds_bytes(const Find3Dev &dev, bool tally) { return (tally ? dev.h1_bytes + dev.h2_bytes + dev.h3c_bytes : 0u) + 16u * kLeft3Wave; }

template <bool HAS1, bool TALLY, bool EMIT>
static hipError_t launch_left3_inst(const Find3Dev &dev, const Find3Args &a, uint32_t blocks, hipStream_t stream) {
    const uint32_t lds = left3_lds_bytes(dev, TALLY);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(left3_select_kernel<HAS1, TALLY, EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((left3_select_kernel<HAS1, TALLY, EMIT>), dim3(blocks), dim3(1024), lds, stream, dev, a);
    return hipGetLastError();
}
// (the kernel's TALLY = false form — exits only — was pass A before the tails had their own kernel; it is not instantiated any more)
hipError_t launch_left3_select(const Find3Dev &dev, const Find3Args &a, bool has_len1, bool tally, uint32_t blocks, hipStream_t stream) {
    if (!tally) return hipErrorInvalidValue;
    return has_len1 ? launch_left3_inst<true, true, false>(dev, a, blocks, stream) : launch_left3_inst<false, true, false>(dev, a, blocks, stream);
}
hipError_t launch_left3_emit(const Find3Dev &dev, const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream) {
    return has_len1 ? launch_left3_inst<true, true, true>(dev, a, blocks, stream) : launch_left3_inst<false, true, true>(dev, a, blocks, stream);
}

}  // namespace daac
