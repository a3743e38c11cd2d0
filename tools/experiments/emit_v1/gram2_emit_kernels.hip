// GRAM engine, tuple emission (gfx950): the find_overlapping match stream as (start, end, value) tuples in the reference's
// order (bytewise/iter.rs:133-176: by end, longest first), written to device memory without a state chain.
//
// Detection is that of gram2_kernels.hip (one M word per position, hits compacted into batches, walkers for the branches
// that go on).  What emission adds is ORDER.  The haystack is cut into tiles of 1024 end positions, one wave-step each:
//   * patterns of length <= K ending at a byte are three flag bits of its M word; their values are table lookups by the
//     1-/2-/3-gram (v1, v2 in LDS, v3 in L2), longest first — already in order within a position;
//   * longer ("deep") matches are found from their START by hits and walkers, so they arrive out of order.  Each sets bit
//     (length - K - 1) in a u16 per end position of the tile (LDS) and is logged as a record {byte, length, value};
//     at one position deep matches have distinct lengths (no duplicate patterns), so the number of set bits above
//     its own gives a record its rank, and they all come before the short ones;
//   * when the tile's walkers have run out, every lane knows the tuple count of each of its 16 positions; a wave scan gives
//     the offsets; the lanes write their short matches, and the records are replayed into their slots.
// Matches that start in one tile and end in the next are carried over in the wave's second record list; a wave begins a
// region with a "prologue" pass over the tile before it (detection only) to pick up those that reach into its first tile.
// Two launches: COUNT leaves the number of tuples per tile, an exclusive scan turns them into offsets, WRITE emits.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

namespace {

typedef uint32_t ge_u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kERing = 128;        // entries of a wave's hit stack
constexpr uint32_t kTile = 1024;        // end positions per wave-step
constexpr uint32_t kMaxExtras = 64;     // matches per tile the per-position length bits cannot carry (longer than K + 16 bytes, further copies of a duplicate pattern)
// per wave: dm[1024] u16, lanebase[64] u32, nsw[64] u32, extras[kMaxExtras] x {position, length | copy << 24}, counters
constexpr uint32_t kWaveLds = 2048 + 256 + 256 + kMaxExtras * 8 + 32;
typedef __attribute__((address_space(3))) const uint32_t ldse_cu32;
typedef __attribute__((address_space(3))) const uint8_t ldse_cu8;

__device__ __forceinline__ uint32_t epin(uint32_t x) {
    asm("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void ge_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}
// One tuple.  daac_match (24 bytes: start, end, value) takes a 16-byte and an 8-byte store; daac_match16 (the crate's own Match
// fields, src/lib.rs:287-291: end, length, value) ONE 16-byte store and a third less write traffic.  Plain stores on purpose: the
// lanes of one store instruction write to different 128-byte lines, and it is the L2 that puts the lines together before they go
// to HBM — the same stores marked non-temporal ran 4.4x slower (profiles/r02_emit_experiments.txt)
template <bool F16>
__device__ __forceinline__ void put_tuple(void *out, unsigned long long slot, unsigned long long start, unsigned long long end, uint32_t value) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (F16) {
        const u32x4 t = {static_cast<uint32_t>(end), static_cast<uint32_t>(end >> 32), static_cast<uint32_t>(end - start), value};
        *reinterpret_cast<u32x4 *>(static_cast<char *>(out) + slot * 16ull) = t;
    } else {
        char *dst = static_cast<char *>(out) + slot * 24ull;
        const u64x2 se = {start, end};
        const u32x2 vp = {value, 0u};
        *reinterpret_cast<u64x2 *>(dst) = se;
        *reinterpret_cast<u32x2 *>(dst + 16) = vp;
    }
}
// inclusive scan over the 64 lanes on the VALU (DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x) {
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t &total) {
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= static_cast<uint32_t>(off)) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

}  // namespace

// K = context length; EM = 0 count per tile, 1 emit; F16 = 16-byte tuples {end, length, value} instead of daac_match; S16 = rank
// directory entries are u16
template <int K, int EM, bool F16, bool S16, int TPB>
__global__ __launch_bounds__(TPB) void gram2_emit_kernel(const Gram2EmitDev g, const EmitArgs a) {
    constexpr int P = 16;
    constexpr bool WRITE = EM != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t offM = kGram2OffM, offS = g.off_s;
    ge_copy(smem, g.cls, 256);
    ge_copy(smem + offM, g.me, g.m_bytes);
    ge_copy(smem + offS, g.sdir, g.s_bytes);
    ge_copy(smem + g.off_v1, g.v1, g.v1_bytes);
    ge_copy(smem + g.off_v2, g.v2, g.v2_bytes);
    __syncthreads();
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<ldse_cu8 *>(static_cast<uintptr_t>(byte)); };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldse_cu32 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t C4 = g.C * 4u, CC4 = g.C * g.C * 4u;
    const uint32_t ub4 = g.unused_byte * 0x01010101u;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t wave_global = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    uint2 *__restrict__ slab = a.wq + wave_global * a.wq_slab;
    uint4 *__restrict__ rec_a = a.recs + wave_global * 2ull * a.rec_cap, *__restrict__ rec_b = rec_a + a.rec_cap;

    // per-wave LDS: deep-match lengths per end position of the tile, lane offsets, short counts, list counters
    char *wl = smem + g.off_wave + wave_in_wg * kWaveLds;
    uint32_t *dm32 = reinterpret_cast<uint32_t *>(wl);            // 512 dwords = 1024 x u16
    uint32_t *lb = reinterpret_cast<uint32_t *>(wl + 2048);       // 64
    uint32_t *nsw = reinterpret_cast<uint32_t *>(wl + 2048 + 256);
    uint2 *xs = reinterpret_cast<uint2 *>(wl + 2048 + 512);       // the tile's extras
    uint32_t *ctr = reinterpret_cast<uint32_t *>(wl + 2048 + 512 + kMaxExtras * 8);  // [0] records of this tile, [1] of the next, [2] / [3] extras among them, [4] scratch

    auto load_chunk = [&](uint32_t v) -> uint4 {
        if (v >= a.vlen) return uint4{ub4, ub4, ub4, ub4};
        const ge_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const ge_u32x4_t *>(hay + v));
        uint4 r{q.x, q.y, q.z, q.w};
        if (v < a.lead || v + 16 > a.vlen) {
            uint32_t w[4] = {r.x, r.y, r.z, r.w};
            for (int b = 0; b < 16; ++b) {
                const uint32_t p = v + b;
                if (p < a.lead || p >= a.vlen) w[b >> 2] = (w[b >> 2] & ~(0xffu << (8 * (b & 3)))) | (g.unused_byte << (8 * (b & 3)));
            }
            r = uint4{w[0], w[1], w[2], w[3]};
        }
        return r;
    };
    auto byte_at = [&](uint32_t p) -> uint32_t { return (p >= a.lead && p < a.vlen) ? hay[p] : g.unused_byte; };

    uint32_t wq_n = 0;                  // wave-uniform
    uint32_t sb = 0;                    // first byte of the current tile (wave-uniform)
    bool prologue = false;              // wave-uniform
    uint4 *cur_list = rec_a, *next_list = rec_b;

    // a deep match: `p` = its last byte, `len` its length, found while the wave works on tile [sb, sb + 1024); `copy` > 0: a further copy of
    // a pattern registered more than once (its value comes from the duplicate list).  Matches the per-position length bits cannot carry —
    // longer than K + 16 bytes, or copies — are EXTRAS: logged like the others, placed by the slower path at the end of the tile.
    auto log_deep = [&](uint32_t p, uint32_t len, uint32_t value, uint32_t copy) {
        if (p < a.emit_from) return;
        const bool here = p < sb + kTile;
        if (here && prologue) return;   // belongs to the tile before this wave's region
        const bool extra = len - (K + 1) >= 16u || copy != 0;
        const uint32_t slot = atomicAdd(&ctr[here ? 0 : 1], 1u);
        if (slot >= a.rec_cap) { atomicOr(a.fail, 1u); return; }   // (also in the COUNT pass: the caller falls back before it allocates)
        if (extra) atomicAdd(&ctr[here ? 2 : 3], 1u);
        if (WRITE) {
            (here ? cur_list : next_list)[slot] = uint4{p, len, value, extra ? (0x80000000u | copy) : 0u};
            if (here && !extra) atomicOr(&dm32[(p - sb) >> 1], 1u << ((len - (K + 1)) + 16u * ((p - sb) & 1u)));
        }
    };
    // a state that ends a pattern: its own match, then the further copies of a duplicate (erec.w / ehit4.w: count << 24)
    auto log_state = [&](uint32_t p, uint32_t len, uint32_t value, uint32_t ncopies, uint32_t state) {
        log_deep(p, len, value, 0u);
        if (ncopies != 0) {
            const uint32_t off = g.dupo[state];
            for (uint32_t k = 0; k < ncopies; ++k) log_deep(p, len, g.dupv[off + k], k + 1u);
        }
    };

    // walkers: {byte position p of the last byte of a (K+1)-gram, the depth-(K+2) state reached on the byte at p + 1 | class of
    // the byte at p + 2 << 27}
    // what one lane stored to the wave's slab / record lists is read back by another lane: the stores have to be out first
    auto mem_settle = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto drain = [&]() {
        if (wq_n != 0) mem_settle();
        for (uint32_t i = lane; i < wq_n; i += 64) {
            const uint2 e = slab[i];
            uint32_t vnext = e.x + 2;   // the state consumed the byte before vnext
            uint32_t state = e.y & 0x07ffffffu;
            uint4 r = g.erec[state];  // {cmap | own, first_child, own_value, depth | further copies << 24}
            uint32_t kn = e.y >> 27;
            uint32_t ahead = 0, n_ahead = 0;
            for (;;) {
                if (r.x & 1u) log_state(vnext - 1, r.w & 0xffffffu, r.z, r.w >> 24, state);
                if (((r.x >> kn) & 1u) == 0 || kn == 0) break;
                state = r.y + __popc(r.x & ((1u << kn) - 1u) & ~1u);
                r = g.erec[state];
                ++vnext;
                if (n_ahead == 0) {
                    ahead = 0;
                    for (int b = 3; b >= 0; --b) ahead = (ahead << 8) | byte_at(vnext + b);
                    n_ahead = 4;
                }
                kn = cls_of(ahead & 0xffu);
                ahead >>= 8;
                --n_ahead;
            }
        }
        wq_n = 0;
    };

    uint2 *ring = reinterpret_cast<uint2 *>(smem + g.off_ring) + wave_in_wg * kERing;
    uint32_t q_n = 0;
    uint4 pend = uint4{0u, 0u, 0u, 0u};
    uint32_t pend_item = 0, pend_pos = 0, pend_rank = 0;
    bool pend_valid = false;
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;  // {cmap | own, own_value, first_child, further copies << 24}; zero for idle lanes
        if (r.x & 1u) log_state(pend_pos, K + 1, r.y, r.w >> 24, g.level_start + pend_rank);
        const uint32_t k1 = (pend_item >> 22) & 31u;
        const bool go = k1 != 0 && ((r.x >> k1) & 1u);
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go)
                (slab + wq_n)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0))] =
                    uint2{pend_pos, (r.z + __popc(r.x & ((1u << k1) - 1u) & ~1u)) | ((pend_item >> 27) << 27)};
            wq_n += __popcll(m);
        }
    };
    auto deep_rank = [&](uint32_t am, uint32_t d) -> uint32_t {
        const uint32_t rel = am - offM;
        const uint4 q = *reinterpret_cast<const uint4 *>(smem + offM + (rel & ~15u));
        const uint32_t idx = (rel >> 2) & 3u;
        const uint32_t base = S16 ? *reinterpret_cast<const uint16_t *>(smem + offS + ((rel >> 4) << 1))
                                  : *reinterpret_cast<const uint32_t *>(smem + offS + ((rel >> 4) << 2));
        constexpr uint32_t kBits = 0x1ffffffeu;
        const uint32_t own = idx == 0 ? q.x : idx == 1 ? q.y : idx == 2 ? q.z : q.w;
        uint32_t below = __popc(own & kBits & ((1u << d) - 1u));
        below += idx > 0 ? __popc(q.x & kBits) : 0u;
        below += idx > 1 ? __popc(q.y & kBits) : 0u;
        below += idx > 2 ? __popc(q.z & kBits) : 0u;
        return base + below;
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
            pend_rank = deep_rank(it.x & 0x1ffffu, (it.x >> 17) & 31u);
            pend = g.ehit4[pend_rank];
        }
        pend_valid = true;
    };

    for (uint64_t region = wave_global; region < a.nregions; region += nwaves) {
        const uint32_t t_begin = static_cast<uint32_t>(region) * a.tiles_per_region;
        const uint32_t t_end = t_begin + a.tiles_per_region < a.ntiles ? t_begin + a.tiles_per_region : a.ntiles;
        uint32_t t = t_begin > 0 ? t_begin - 1 : t_begin;
        if (lane < 5) ctr[lane] = 0;
        cur_list = rec_a;
        next_list = rec_b;
        q_n = 0;
        wq_n = 0;
        pend_valid = false;
        // classes of the K bytes before the first tile, oldest in the low byte
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) carry |= (t * kTile >= static_cast<uint32_t>(K - i) ? cls_of(byte_at(t * kTile - (K - i))) : 0u) << (8 * i);
        uint4 pf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) pf[i] = load_chunk((t + i) * kTile + lane * P);

        for (; t < t_end; ++t) {
            sb = t * kTile;
            prologue = t < t_begin;
            const uint32_t v = sb + lane * P;
            const uint4 cur = pf[0];
            pf[0] = pf[1];
            pf[1] = load_chunk(v + 2 * kTile);
            unsigned long long tile_base = 0;
            if (WRITE && !prologue) tile_base = a.tile_cnt[t];  // exclusive offset of this tile's first tuple

            // the two bytes after the tile (lane 63 needs their classes)
            const uint32_t after2 = byte_at(sb + kTile) | (byte_at(sb + kTile + 1) << 8);

            uint32_t kx[K + P + 2];
            {
                const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    kx[K + b] = epin(cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu));
                    __builtin_assume(kx[K + b] < 32u);
                }
            }
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) pk |= kx[P + i] << (8 * i);
            uint32_t left = __shfl_up(pk, 1, 64);
            if (lane == 0) left = carry;
            carry = __builtin_amdgcn_readlane(pk, 63);
#pragma unroll
            for (int i = 0; i < K; ++i) { kx[i] = (left >> (8 * i)) & 0xffu; __builtin_assume(kx[i] < 32u); }
            const uint32_t right63 = cls_of(after2 & 0xffu) | (cls_of((after2 >> 8) & 0xffu) << 5);
            uint32_t right = __shfl_down(kx[K] | (kx[K + 1] << 5), 1, 64);
            right = lane == 63 ? right63 : right;
            kx[K + P] = right & 31u;
            kx[K + P + 1] = right >> 5;

            // LDS address of the M word of the K-gram ending at j, j = -1 .. P-1
            uint32_t am[P + 1];
#pragma unroll
            for (int j = -1; j < P; ++j) {
                uint32_t x = epin((kx[K + j] << 2) + offM);
                x = __umul24(kx[K + j - 1], C4) + x;
                if (K == 3) x = __umul24(kx[K + j - 2], CC4) + epin(x);
                am[j + 1] = x;
            }
            uint32_t tri[P];
            tri[P - 1] = (((kx[K + P + 1] << 5) | kx[K + P]) << 5) | kx[K + P - 1];
#pragma unroll
            for (int j = P - 2; j >= 0; --j) tri[j] = (tri[j + 1] << 5) | kx[K + j];

            // this tile's per-position deep lengths start empty; what the previous tile's walkers found for this one is replayed
            if (WRITE) {
#pragma unroll
                for (int q = 0; q < 2; ++q) reinterpret_cast<uint4 *>(dm32)[lane * 2 + q] = uint4{0u, 0u, 0u, 0u};
                const uint32_t ncur = min(*reinterpret_cast<volatile uint32_t *>(&ctr[0]), a.rec_cap);  // (a list that overflowed set a.fail: the scan is redone elsewhere)
                if (ncur != 0) mem_settle();  // only then: the counter drains this wave's stores too, the previous tile's tuples among them
                for (uint32_t i = lane; i < ncur; i += 64) {
                    const uint4 r = cur_list[i];
                    if (!(r.w >> 31)) atomicOr(&dm32[(r.x - sb) >> 1], 1u << ((r.y - (K + 1)) + 16u * ((r.x - sb) & 1u)));
                }
            }

            // ---- detection: M words, short flags, hits ----
            uint32_t flags = 0;       // 2 bits per position: number of short patterns ending there
            uint32_t fl3[2] = {0, 0};  // 3 flag bits per position (10 positions per dword would do; two dwords of 8 keep it simple)
            uint32_t mprev = lds_u32(am[0]);
            uint32_t am_prev = am[0];  // LDS address of the M word of the K-gram ending at j - 1
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const uint32_t am_here = am[j + 1];
                const uint32_t mw = lds_u32(am_here);
                uint32_t f = mw >> 29;
                if (prologue || v + j < a.emit_from) f = 0;  // (bytes at and beyond vlen are class 0: no flags there)
                fl3[j >> 3] |= f << (4 * (j & 7));
                flags |= static_cast<uint32_t>(__popc(f)) << (2 * j);
                // the value of a 3-byte pattern comes from L2: asked for now, used when the tile's tuples are written (by then
                // the hits and walkers of the tile have been through, and the answer is there); it takes the place of am[j + 1]
                if (WRITE && K == 3) am[j + 1] = (f & 4u) ? g.v3[(am_here - offM) >> 2] : 0u;
                const bool hit = __builtin_amdgcn_ubfe(mprev, kx[K + j], 1) != 0;
                const unsigned long long m = __ballot(hit);
                if (m != 0) {
                    const uint32_t q_s = q_n;
                    if (hit)
                        (ring + q_s)[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] =
                            uint2{(tri[j] << 17) | am_prev, v + j};
                    q_n = q_s + static_cast<uint32_t>(__popcll(m));
                    if (q_n >= 64u) process_batch();
                }
                mprev = mw;
                am_prev = am_here;
            }
            // everything this tile's hits lead to has to be known before its tuples can be placed
            while (q_n != 0) process_batch();
            consume_pending();
            drain();

            if (!prologue) {
                // ---- tuples per position: deep (bits of dm) + short (flags) ----
                uint32_t nshort = 0;
#pragma unroll
                for (int j = 0; j < P; ++j) nshort += (flags >> (2 * j)) & 3u;
                if (!WRITE) {
                    uint32_t total;
                    (void)wave_excl_scan(nshort, lane, total);
                    if (lane == 0) a.tile_cnt[t] = static_cast<unsigned long long>(total) + *reinterpret_cast<volatile uint32_t *>(&ctr[0]);
                } else {
                    uint32_t dmw[8];
                    {
                        const uint4 d0 = reinterpret_cast<const uint4 *>(dm32)[lane * 2], d1 = reinterpret_cast<const uint4 *>(dm32)[lane * 2 + 1];
                        dmw[0] = d0.x; dmw[1] = d0.y; dmw[2] = d0.z; dmw[3] = d0.w; dmw[4] = d1.x; dmw[5] = d1.y; dmw[6] = d1.z; dmw[7] = d1.w;
                    }
                    uint32_t ndeep = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) ndeep += __popc(dmw[w]);
                    mem_settle();
                    const uint32_t ncur = min(*reinterpret_cast<volatile uint32_t *>(&ctr[0]), a.rec_cap);  // (a list that overflowed set a.fail: the scan is redone elsewhere)
                    // ---- the tile's extras (rare): gathered into LDS, one per lane; every lane learns how many fall on each of its positions ----
                    uint32_t xn = *reinterpret_cast<volatile uint32_t *>(&ctr[2]);   // wave-uniform
                    uint2 ex = uint2{0xffffffffu, 0u};   // this lane's extra: {position, length | copy << 24}
                    unsigned long long xin = 0;          // extras on this lane's 16 positions, four bits each
                    uint32_t nex = 0;
                    if (xn != 0) {
                        if (xn > kMaxExtras) { if (lane == 0) atomicOr(a.fail, 2u); xn = kMaxExtras; }
                        if (lane == 0) ctr[4] = 0;
                        for (uint32_t i = lane; i < ncur; i += 64) {
                            const uint4 r = cur_list[i];
                            if (r.w >> 31) {
                                const uint32_t idx = atomicAdd(&ctr[4], 1u);
                                if (idx < kMaxExtras) xs[idx] = uint2{r.x, r.y | ((r.w & 0xffu) << 24)};
                            }
                        }
                        if (lane < xn) ex = xs[lane];
                        for (uint32_t i = 0; i < xn; ++i) {
                            const uint32_t ep = __builtin_amdgcn_readlane(ex.x, i);
                            if (ep - v < 16u) {
                                if (((xin >> (4u * (ep - v))) & 15u) == 15u) atomicOr(a.fail, 4u);
                                xin += 1ull << (4u * (ep - v));
                                ++nex;
                            }
                        }
                    }
                    uint32_t total;
                    const uint32_t lanebase = wave_excl_scan(nshort + ndeep + nex, lane, total);
                    lb[lane] = lanebase;
                    nsw[lane] = flags;
                    void *__restrict__ out = reinterpret_cast<char *>(a.out) + tile_base * (F16 ? 16ull : 24ull);
                    // tuples of lane L that lie before its position j, extras left aside: deep ones (length bits) and short ones (flags)
                    auto before_of = [&](uint32_t L, uint32_t j, uint32_t &here16) -> uint32_t {
                        const uint4 d0 = reinterpret_cast<const uint4 *>(dm32)[L * 2], d1 = reinterpret_cast<const uint4 *>(dm32)[L * 2 + 1];
                        const uint32_t dw[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                        uint32_t before = 0;
#pragma unroll
                        for (uint32_t w = 0; w < 8; ++w) {
                            const uint32_t keep = 2 * w + 1 < j ? 0xffffffffu : 2 * w < j ? 0xffffu : 0u;  // positions 2w, 2w+1 below j
                            before += __popc(dw[w] & keep);
                        }
                        const uint32_t x = nsw[L] & ((1u << (2 * j)) - 1u);
                        before += __popc(x & 0x55555555u) + 2u * __popc(x & 0xaaaaaaaau);
                        here16 = (dw[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
                        return before;
                    };
                    // extras of the same lane-group of 16 positions that come before the key (position, length desc, copy asc)
                    auto extras_before = [&](uint32_t p, uint32_t len, uint32_t copy) -> uint32_t {
                        uint32_t n = 0;
                        for (uint32_t i = 0; i < xn; ++i) {
                            const uint32_t ep = __builtin_amdgcn_readlane(ex.x, i), ey = __builtin_amdgcn_readlane(ex.y, i);
                            const uint32_t el = ey & 0xffffffu, ec = ey >> 24;
                            if (((ep - sb) >> 4) != ((p - sb) >> 4)) continue;
                            n += (ep < p || (ep == p && (el > len || (el == len && ec < copy)))) ? 1u : 0u;
                        }
                        return n;
                    };
                    // deep matches: replay the records of this tile into their slots
                    for (uint32_t i0 = 0; i0 < ncur; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        uint4 r = uint4{0u, 0u, 0u, 0x80000000u};
                        if (i < ncur) r = cur_list[i];
                        const bool normal = !(r.w >> 31);
                        uint32_t slot = 0;
                        if (normal) {
                            const uint32_t rel = r.x - sb;
                            uint32_t here16;
                            slot = lb[rel >> 4] + before_of(rel >> 4, rel & 15u, here16) + __popc(here16 >> (r.y - (K + 1) + 1));
                        }
                        if (xn != 0) slot += extras_before(normal ? r.x : sb, normal ? r.y : 0u, 0u);  // (all lanes walk the list together)
                        if (normal) {
                            const unsigned long long end = a.pos_base + r.x;
                            put_tuple<F16>(out, slot, end - r.y, end, r.z);
                        }
                    }
                    if (xn != 0) {  // the extras themselves, one per lane: after the deep matches of their position that are longer (and their own original)
                        const uint32_t ep = ex.x, el = ex.y & 0xffffffu, ec = ex.y >> 24;
                        const bool mine = lane < xn;
                        uint32_t slot = 0;
                        if (mine) {
                            const uint32_t rel = ep - sb;
                            uint32_t here16;
                            slot = lb[rel >> 4] + before_of(rel >> 4, rel & 15u, here16);
                            if (el - (K + 1) < 16u) slot += __popc(here16 >> (el - (K + 1) + 1)) + (ec != 0 ? 1u : 0u);
                        }
                        slot += extras_before(mine ? ep : sb, mine ? el : 0u, mine ? ec : 0u);
                        if (mine) {
                            // the value of an extra is in its record: find it again (the list is short and this is the slow path)
                            uint32_t val = 0;
                            for (uint32_t i = 0; i < ncur; ++i) {
                                const uint4 r = cur_list[i];
                                if ((r.w >> 31) && r.x == ep && r.y == el && (r.w & 0xffu) == ec) { val = r.z; break; }
                            }
                            const unsigned long long end = a.pos_base + ep;
                            put_tuple<F16>(out, slot, end - el, end, val);
                        }
                    }
                    // short matches: each lane walks its 16 positions
                    uint32_t running = lanebase;
                    const unsigned long long end0 = a.pos_base + v;
#pragma unroll
                    for (int j = 0; j < P; ++j) {
                        running += __popc((dmw[j >> 1] >> (16 * (j & 1))) & 0xffffu) + static_cast<uint32_t>((xin >> (4 * j)) & 15u);
                        const uint32_t f = (fl3[j >> 3] >> (4 * (j & 7))) & 7u;
                        if (__ballot(f != 0) == 0) continue;
                        const unsigned long long end = end0 + j;
                        if (K == 3 && (f & 4u)) {
                            put_tuple<F16>(out, running++, end - 3, end, am[j + 1]);  // (the value asked for during detection)
                        }
                        if (f & 2u) {
                            put_tuple<F16>(out, running++, end - 2, end, lds_u32(g.off_v2 + 4u * (kx[K + j - 1] * g.C + kx[K + j])));
                        }
                        if (f & 1u) {
                            put_tuple<F16>(out, running++, end - 1, end, lds_u32(g.off_v1 + 4u * kx[K + j]));
                        }
                    }
                }
            }
            // the next tile's list becomes the current one
            {
                uint4 *tmp = cur_list; cur_list = next_list; next_list = tmp;
                const uint32_t nn = *reinterpret_cast<volatile uint32_t *>(&ctr[1]), nx = *reinterpret_cast<volatile uint32_t *>(&ctr[3]);
                if (lane == 0) { ctr[0] = nn; ctr[1] = 0; ctr[2] = nx; ctr[3] = 0; }
            }
        }
    }
}

template <int K, int EM, bool F16>
static hipError_t launch_e(const Gram2EmitDev &dev, const EmitArgs &a, uint32_t blocks, hipStream_t stream) {
    hipError_t e;
    const uint32_t lds = dev.lds_bytes;
    if (dev.s16) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram2_emit_kernel<K, EM, F16, true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds))) != hipSuccess) return e;
        hipLaunchKernelGGL((gram2_emit_kernel<K, EM, F16, true, 1024>), dim3(blocks), dim3(1024), lds, stream, dev, a);
    } else {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(gram2_emit_kernel<K, EM, F16, false, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds))) != hipSuccess) return e;
        hipLaunchKernelGGL((gram2_emit_kernel<K, EM, F16, false, 1024>), dim3(blocks), dim3(1024), lds, stream, dev, a);
    }
    return hipGetLastError();
}

// em: 0 count, 1 write; f16: 16-byte tuples {end, length, value} instead of daac_match
hipError_t launch_gram2_emit(const Gram2EmitDev &dev, const EmitArgs &a, int em, bool f16, uint32_t blocks, hipStream_t stream) {
    if (dev.K == 3) return em == 0 ? launch_e<3, 0, false>(dev, a, blocks, stream) : f16 ? launch_e<3, 1, true>(dev, a, blocks, stream) : launch_e<3, 1, false>(dev, a, blocks, stream);
    return em == 0 ? launch_e<2, 0, false>(dev, a, blocks, stream) : f16 ? launch_e<2, 1, true>(dev, a, blocks, stream) : launch_e<2, 1, false>(dev, a, blocks, stream);
}

}  // namespace daac
