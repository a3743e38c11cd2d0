// PFX engine kernels (gfx950): `.count()` of the find_overlapping stream for bytewise automata over ANY byte alphabet.
// Tables and method: pfx.hpp.  Occurrences are found from their START (SURVEY §8a note C): every position hashes its next
// G bytes into a Bloom bitmap in LDS (one lookup per haystack byte, no byte classes); the survivors are queued, looked up in a
// perfect hash of the depth-G trie paths (one displacement from LDS, one 16-byte record from L2) and, where the trie goes on
// below depth G, walked goto-only over the double array (child = BASE ^ byte, CHECK == byte: reference src/bytewise.rs:1070-1077)
// adding the patterns that end in every state on the way.  One-byte patterns are a 256-entry table in LDS.
//
// The skeleton is gram3_kernels.hip's: 16 positions per lane and step, coalesced 16-byte non-temporal haystack loads two steps
// ahead, the step's kilobyte of text written to an LDS slot (plus the sixteen bytes behind it) so that the consumer — 64
// survivors wide — takes keys and follow-up text from LDS, lane-local hit masks queued one set bit per lane and turn, one batch
// of records in flight, walkers in per-wave slabs.
//
// Roofline: HBM bytes of haystack (1 B read per byte); integer work only, no MFMA.
// PFX_NO_* : decomposition builds only (tools/ab_libs_pfx.sh, profiles/r03_pfx_decomposition.txt) — they cut a stage out to price it, the counts they give are WRONG, and
// nothing in the shipped library defines them (_build.py passes no -D).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"
#include "pfx.hpp"

namespace daac {

namespace {

typedef uint32_t px_u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kRingP = 128;     // entries of a wave's survivor queue (FIFO; at most 63 left over + 64 new)
typedef __attribute__((address_space(3))) const uint32_t ldsp_cu32;
typedef __attribute__((address_space(3))) uint32_t ldsp_u32;
typedef __attribute__((address_space(3))) const uint16_t ldsp_cu16;
typedef __attribute__((address_space(3))) px_u32x4_t ldsp_u32x4;

// lane i <- lane i + 1 of `v`; lane 63 keeps `lane63`
__device__ __forceinline__ uint32_t wave_shl1_p(uint32_t v, uint32_t lane63) {
    uint32_t d = lane63;
    asm volatile("s_nop 1\nv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ unsigned long long px_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void px_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

}  // namespace

// G = key bytes (2 .. 6); LEN1 = the dictionary has one-byte patterns; EXACT = count + checksum: every match is met as its own state (no tail
// records, no path filter: slots_x / wrec_x) and adds h32 and h32 * end to the sums
template <int G, bool LEN1, bool EXACT, int TPB>
__global__ __launch_bounds__(TPB) void pfx_kernel(const PfxDev g, const GramArgs a) {
    constexpr int P = 16;
    constexpr uint32_t SB = 64u * P;          // bytes a wave takes per step
    constexpr uint32_t SLOT = SB + 32u;       // the step | the first 16 bytes of the next (read up to 20 bytes past a key's first byte)
    constexpr uint32_t KMASK0 = G >= 4 ? 0xffffffffu : ((1u << (8 * (G & 3))) - 1u);
    constexpr uint32_t KMASK1 = G <= 4 ? 0u : ((1u << (8 * ((G - 4) & 3))) - 1u);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    px_copy(smem, g.bloom, g.bloom_bytes);
    px_copy(smem + g.off_disp, g.disp, g.disp_bytes);
    px_copy(smem + g.off_cnt1, g.cnt1, 512);
    if (EXACT) px_copy(smem + g.off_cnt1 + 512, g.hs1, 1024);
    __syncthreads();
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();  // tables are read through absolute LDS addresses
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsp_cu32 *>(static_cast<uintptr_t>(addr)); };
    auto lds_u16 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsp_cu16 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    const uint32_t tb = g.off_wave + wave_in_wg * g.wave_stride;   // this wave's LDS: two text slots, then the survivor queue
    const uint32_t ringb = tb + 2u * SLOT;
    // this wave's slab of pending walkers: {low 32 bits of the virtual position of the key's first byte, BASE of the depth-G state,
    // the eight text bytes behind the key}
    uint4 *__restrict__ slab = reinterpret_cast<uint4 *>(a.wq) + (static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg) * a.wq_slab;
    uint32_t wq_n = 0, slab_hi = 0;  // wave-uniform

    unsigned long long tot_cnt = 0;
    uint32_t cnt32 = 0;
    uint32_t s1 = 0, s2 = 0;   // EXACT: sum of h32, sum of h32 * end (mod 2^32; ends count from the haystack's first byte)

    auto load_chunk = [&](uint64_t v) -> uint4 {
        if (v >= a.vlen) return uint4{0u, 0u, 0u, 0u};
        const px_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const px_u32x4_t *>(hay + v));
        return uint4{q.x, q.y, q.z, q.w};  // (bytes outside [lead, vlen) are whatever memory holds: starts there are masked out)
    };
    auto read_ahead = [&](uint64_t v) -> unsigned long long {
        unsigned long long x;
        if (v + 8 <= a.vlen) {
            __builtin_memcpy(&x, hay + v, 8);
        } else {
            x = 0;
            for (int b = 7; b >= 0; --b) x = (x << 8) | ((v + b < a.vlen) ? hay[v + b] : 0u);
        }
        return x;
    };

    // Finishes the queued branches: goto-only over the double array, every state met adds the patterns that end in it.  A step is one
    // dependent 8-byte gather from L2, so every lane walks W branches side by side (one at a time the drain was half of the kernel's
    // time on text that keeps the walkers busy: profiles/r03_pfx_decomposition.txt).
    auto drain = [&]() {
        constexpr int W = 4;
        for (uint32_t base_i = 0; base_i < wq_n; base_i += 64u * W) {
            uint64_t vn[W];
            uint32_t b[W], n_ahead[W];
            unsigned long long ah[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t i = base_i + 64u * w + lane;
                uint4 e = uint4{0u, 0u, 0u, 0u};
                if (i < wq_n) e = slab[i];
                vn[w] = ((static_cast<uint64_t>(slab_hi) << 32) | e.x) + G;  // the next byte to take
                b[w] = e.y;
                n_ahead[w] = 8;
                ah[w] = (static_cast<unsigned long long>(e.w) << 32) | e.z;
                if (vn[w] >= a.vlen) b[w] = 0;
            }
            for (;;) {
                uint4 r[W];
                bool any = false;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    r[w] = uint4{0u, 0u, 0u, 0u};
                    if (b[w] != 0) {
                        if (n_ahead[w] == 0) { ah[w] = read_ahead(vn[w]); n_ahead[w] = 8; }
                        const uint32_t slot = b[w] ^ (static_cast<uint32_t>(ah[w]) & 0xffu);
                        if (EXACT) r[w] = g.wrec_x[slot];
                        else { const uint2 q = g.wrec[slot]; r[w] = uint4{q.x, q.y, 0u, 0u}; }
                        any = true;
                    }
                }
                if (!__any(any)) break;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (b[w] == 0) continue;
                    const uint32_t c = static_cast<uint32_t>(ah[w]) & 0xffu;
                    if ((r[w].y & 0xffu) != c) { b[w] = 0; continue; }
                    cnt32 += r[w].y >> 8;
                    b[w] = r[w].x;
                    ++vn[w];
                    if (EXACT) { s1 += r[w].z; s2 += r[w].z * (static_cast<uint32_t>(vn[w]) - a.lead); }  // the match ends behind the byte just taken
                    ah[w] >>= 8;
                    --n_ahead[w];
                    if (vn[w] >= a.vlen) b[w] = 0;
                }
            }
        }
        wq_n = 0;
    };

    // ---- the survivor queue: entry = LDS address of the key's first byte in one of the wave's two text slots
    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running
    uint32_t posbias0 = 0, posbias1 = 0;  // per slot: (low 32 bits of the virtual position of a byte) - (its LDS address)
    uint4 pend = uint4{0u, kPfxEmpty, 0u, 0u};  // the record read for the previous batch (pfx.hpp: SLOTS)
    uint32_t pend_pos = 0, pend_k0 = 0, pend_k1 = 0, pend_t0 = 0, pend_t1 = 0;
    bool pend_valid = false;           // wave-uniform
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;
        // (an empty slot and an idle lane carry kPfxEmpty in r.y: 0x8000 in the upper half never equals a key)
        const bool match = r.x == pend_k0 && (r.y & 0x8000ffffu) == pend_k1;
        bool go = false;
        if (match) {
            if (!EXACT && (r.y & kPfxTail)) {  // one path below the key: compared with the text behind it, no walk
                const uint32_t edges = (r.y >> 16) & 15u;
                const unsigned long long path = (static_cast<unsigned long long>(r.w) << 32) | r.z;
                const unsigned long long text = (static_cast<unsigned long long>(pend_t1) << 32) | pend_t0;
                const unsigned long long diff = path ^ text;
                uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
                same = same < edges ? same : edges;
                // ... no further than the haystack goes
                const uint64_t after = ((static_cast<uint64_t>(slab_hi) << 32) | pend_pos) + G;
                const uint64_t left = a.vlen > after ? a.vlen - after : 0;
                same = left < same ? static_cast<uint32_t>(left) : same;
                cnt32 += __popc((r.y >> 20) & ((2u << same) - 1u) & 0x1ffu);
            } else {
                cnt32 += (r.y >> 16) & 0x3fffu;
                if (EXACT) {   // r.w: sum of h32 of the patterns that are the key
                    s1 += r.w;
                    s2 += r.w * (pend_pos - a.lead + G);
                    go = r.z != 0;
                } else {
                    // the two bytes behind the key against the record's filter of two-byte paths (with fewer than two bytes left: walk)
                    const uint64_t after = ((static_cast<uint64_t>(slab_hi) << 32) | pend_pos) + G;
                    go = r.z != 0 && (((r.w >> pfx_pair_bit(pend_t0)) & 1u) || after + 2 > a.vlen);
                }
            }
        }
#ifdef PFX_NO_WALKERS
        go = false;
#endif
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go) {
                const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                slab[at] = uint4{pend_pos, r.z, pend_t0, pend_t1};
            }
            wq_n += __popcll(m);
        }
    };
    auto process_batch = [&](uint32_t n) {  // n <= 64 entries from the head of the queue
#ifdef PFX_NO_CONSUMER
        q_head += n; return;
#endif
        __builtin_amdgcn_s_setprio(2);
        consume_pending();
        pend = uint4{0u, kPfxEmpty, 0u, 0u};
        if (lane < n) {
            const uint32_t e = lds_u32(ringb + (((q_head + lane) & (kRingP - 1u)) << 2));
            pend_pos = e + ((e - tb) >= SLOT ? posbias1 : posbias0);
            const uint32_t a0 = e & ~3u, sh = e & 3u;
            // (slots are self-contained: the key and the six bytes behind it never reach past slot + SLOT)
            const uint32_t d0 = lds_u32(a0), d1 = lds_u32(a0 + 4u), d2 = lds_u32(a0 + 8u), d3 = lds_u32(a0 + 12u);
            const uint32_t x0 = __builtin_amdgcn_alignbyte(d1, d0, sh), x1 = __builtin_amdgcn_alignbyte(d2, d1, sh), x2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
            const uint32_t key0 = x0 & KMASK0, key1 = x1 & KMASK1;
            pend_k0 = key0;
            pend_k1 = key1;
            // the eight bytes from s + G on (s + G + 7 <= s + 13: a fifth dword for the longer keys)
            if (G == 4) { pend_t0 = x1; pend_t1 = x2; }
            else if (G < 4) { pend_t0 = __builtin_amdgcn_alignbyte(x1, x0, G & 3); pend_t1 = __builtin_amdgcn_alignbyte(x2, x1, G & 3); }
            else {
                const uint32_t d4 = lds_u32(a0 + 16u);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
                pend_t0 = __builtin_amdgcn_alignbyte(x2, x1, G & 3);
                pend_t1 = __builtin_amdgcn_alignbyte(x3, x2, G & 3);
            }
            const uint32_t mb = key0 * kPfxMulBucket0 + (key1 ^ g.seed) * kPfxMulBucket1;
            const uint32_t ms = key0 * kPfxMulSlot0 + (key1 ^ g.seed) * kPfxMulSlot1;
            const uint32_t bucket = __umulhi(mb, g.buckets);
            const uint32_t d = lds_u16(g.off_disp + (bucket << 1));
            pend = (EXACT ? g.slots_x : g.slots)[pfx_slot(ms, d, g.n_slots)];
        }
        q_head += n;
        pend_valid = true;
    };

    uint32_t sl = 0;          // slot of the current step (wave-uniform)
    uint32_t carry_in = 0;    // queued entries that belong to the step before the current one
    uint64_t region = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    // starts are valid in [lead, vlen - G]
    const uint64_t start_end = a.vlen >= static_cast<uint64_t>(G) ? a.vlen - G + 1 : 0;
    while (region < a.nregions) {
      slab_hi = static_cast<uint32_t>((region * a.region_bytes) >> 32);
      for (; region < a.nregions && static_cast<uint32_t>((region * a.region_bytes) >> 32) == slab_hi; region += nwaves) {
        const uint64_t rbase = region * a.region_bytes;
        const uint64_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        // the chunk of the step at s0; past the region's end only lane 0's (it feeds the last step's trailer and lane 63's keys)
        auto fetch = [&](uint64_t s0) -> uint4 {
            if (s0 < rend) return load_chunk(s0 + lane * P);
            if (lane == 0 && s0 < rend + SB) return load_chunk(s0);
            return uint4{0u, 0u, 0u, 0u};
        };
        uint4 pf0 = fetch(rbase), pf1 = fetch(rbase + SB);

        for (uint64_t sb = rbase; sb < rend; sb += SB) {
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint64_t v = sb + lane * P;
            const uint4 cur = pf0;
            pf0 = pf1;
            consume_pending();  // before the next chunk is requested: loads retire in order
            __builtin_amdgcn_s_setprio(0);
            pf1 = fetch(sb + 2ull * SB);

            // ---- this step's text into its slot ----
            const uint32_t slot = tb + sl * SLOT;                 // wave-uniform
            const uint32_t my_text = slot + lane * P;             // LDS address of this lane's first byte
            {
                const uint32_t bias = static_cast<uint32_t>(sb) - slot;
                if (sl) posbias1 = bias; else posbias0 = bias;
            }
            *reinterpret_cast<ldsp_u32x4 *>(static_cast<uintptr_t>(my_text)) = px_u32x4_t{cur.x, cur.y, cur.z, cur.w};
            if (lane == 0) *reinterpret_cast<ldsp_u32x4 *>(static_cast<uintptr_t>(slot + SB)) = px_u32x4_t{pf0.x, pf0.y, pf0.z, pf0.w};

            // ---- keys: the G bytes from each of this lane's 16 positions on (the last ones reach into the next lane's chunk) ----
            uint32_t W[6] = {cur.x, cur.y, cur.z, cur.w, 0u, 0u};
            W[4] = wave_shl1_p(cur.x, __builtin_amdgcn_readfirstlane(pf0.x));
            W[5] = wave_shl1_p(cur.y, __builtin_amdgcn_readfirstlane(pf0.y));
            uint32_t H = 0, c1 = 0, h1 = 0, h2 = 0;   // EXACT, one-byte patterns: sum of h, sum of h * j
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int q = j >> 2, r = j & 3;
                uint32_t k0 = r == 0 ? W[q] : __builtin_amdgcn_alignbyte(W[q + 1], W[q], static_cast<uint32_t>(r));
                if (G < 4) k0 &= KMASK0;
                uint32_t m = k0 * kPfxMulBloom0;
                if (G > 4) {
                    uint32_t k1 = r == 0 ? W[q + 1] : __builtin_amdgcn_alignbyte(W[q + 2], W[q + 1], static_cast<uint32_t>(r));
                    k1 &= KMASK1;
                    m += k1 * kPfxMulBloom1;
                }
                const uint32_t word = lds_u32(__umulhi(m, g.bloom_words) << 2);   // BLOOM sits at LDS offset 0
                const uint32_t m2 = m * kPfxMulBits;
                H |= (__builtin_amdgcn_ubfe(word, m2 >> kPfxBit1, 1) & __builtin_amdgcn_ubfe(word, m2 >> kPfxBit2, 1)) << j;
                if (LEN1) {
                    const uint32_t byte = (W[q] >> (8 * r)) & 0xffu;
                    c1 += lds_u16(g.off_cnt1 + (byte << 1));
                    if (EXACT) { const uint32_t h = lds_u32(g.off_cnt1 + 512u + (byte << 2)); h1 += h; h2 += h * static_cast<uint32_t>(j); }
                }
            }
            // starts before the haystack's first byte or too close to its end do not count (first / last step only)
            if (v < a.lead || sb + SB + G > a.vlen + 1 || (LEN1 && sb + SB > a.vlen)) {
                const uint64_t lo = a.lead > v ? a.lead - v : 0, hi = start_end > v ? start_end - v : 0;
                uint32_t keep = hi >= 16 ? 0xffffu : ((1u << hi) - 1u);
                keep &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
                H &= keep;
                if (LEN1) {
                    c1 = 0; h1 = 0; h2 = 0;
                    for (int j = 0; j < P; ++j)
                        if (v + j >= a.lead && v + j < a.vlen) {
                            const uint32_t byte = (W[j >> 2] >> (8 * (j & 3))) & 0xffu;
                            c1 += lds_u16(g.off_cnt1 + (byte << 1));
                            if (EXACT) { const uint32_t h = lds_u32(g.off_cnt1 + 512u + (byte << 2)); h1 += h; h2 += h * static_cast<uint32_t>(j); }
                        }
                }
            }
            cnt32 += c1;
            if (EXACT && LEN1) { s1 += h1; s2 += h1 * (static_cast<uint32_t>(v) - a.lead + 1u) + h2; }  // a one-byte match at position v + j ends at v + j + 1

#ifdef PFX_NO_PRODUCER
            H = 0;
#endif
            // ---- queue the survivors, one per lane and turn ----
            bool did_batch = false;
            for (;;) {
                const bool has = H != 0;
                const unsigned long long m = __ballot(has);
                if (m == 0) break;
                if (has) {
                    const uint32_t b = static_cast<uint32_t>(__builtin_ctz(H));
                    H &= H - 1u;
                    const uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                    *reinterpret_cast<ldsp_u32 *>(static_cast<uintptr_t>(ringb + ((at & (kRingP - 1u)) << 2))) = my_text + b;
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
      drain();
      tot_cnt += cnt32;
      cnt32 = 0;
    }
    {
        const unsigned long long c = px_wave_sum(tot_cnt), x1 = px_wave_sum(s1), x2 = px_wave_sum(s2);
        __syncthreads();
        unsigned long long *scratch = reinterpret_cast<unsigned long long *>(smem);
        if (lane == 0) { scratch[wave_in_wg * 3] = c; scratch[wave_in_wg * 3 + 1] = x1; scratch[wave_in_wg * 3 + 2] = x2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long r0 = 0, r1 = 0, r2 = 0;
            for (int w = 0; w < static_cast<int>((blockDim.x + 63) >> 6); ++w) { r0 += scratch[w * 3]; r1 += scratch[w * 3 + 1]; r2 += scratch[w * 3 + 2]; }
            if (r0) atomicAdd(a.result, r0);
            if (EXACT && (r1 | r2)) { atomicAdd(a.result + 1, r1); atomicAdd(a.result + 2, r2); }
        }
    }
}

// ================================================================================================ tuple emission: DETECT
// The count + checksum kernel's search (every match met as its own state: slots_e / wrec_x) with every match LOGGED for the tuple emitter
// of emit3_kernels.hip instead of tallied: a 16-byte record {virtual position of the last byte, length, value, global tile} in the chunked
// list (a wave appends to its open chunk through an LDS cursor and takes a new chunk when the open one is half full), and one count per
// tile of 1024 positions the match ENDS in (atomic adds; the matches of a batch go through a four-entry cache in lanes 0 .. 3).  One-byte
// patterns are not logged: EXPAND meets them in the haystack itself (has1 / v1 by byte); here they are counted per tile — a wave-step is
// exactly one tile.  Needs a dictionary without duplicate patterns (PfxTables::emit_ok).  Windows of at most 1 GiB: 32-bit positions.
template <int G, bool LEN1>
__global__ __launch_bounds__(1024) void pfx_emit_kernel(const PfxDev g, const Emit3Args a) {
    constexpr int P = 16;
    constexpr uint32_t SB = 64u * P;          // bytes a wave takes per step = kEmit3Tile
    constexpr uint32_t SLOT = SB + 32u;       // the step | the first 16 bytes of the next | [SB + 16, SB + 20) of slot 0: the wave's record cursor
    constexpr uint32_t KMASK0 = G >= 4 ? 0xffffffffu : ((1u << (8 * (G & 3))) - 1u);
    constexpr uint32_t KMASK1 = G <= 4 ? 0u : ((1u << (8 * ((G - 4) & 3))) - 1u);
    static_assert(SB == kEmit3Tile, "a wave-step is one tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    px_copy(smem, g.bloom, g.bloom_bytes);
    px_copy(smem + g.off_disp, g.disp, g.disp_bytes);
    px_copy(smem + g.off_cnt1, g.cnt1, 512);
    __syncthreads();
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();  // tables are read through absolute LDS addresses
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsp_cu32 *>(static_cast<uintptr_t>(addr)); };
    auto lds_u16 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsp_cu16 *>(static_cast<uintptr_t>(addr)); };

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const uint32_t tb = g.off_wave + wave_in_wg * g.wave_stride;
    const uint32_t ringb = tb + 2u * SLOT;
    const uint32_t cur_at = tb + SB + 16u;    // LDS address of the record cursor
    uint32_t *cursor = reinterpret_cast<uint32_t *>(smem + cur_at);
    uint4 *__restrict__ slab = reinterpret_cast<uint4 *>(a.wq) + static_cast<uint64_t>(wave_global) * a.wq_slab;
    uint32_t wq_n = 0;  // wave-uniform

    // ---- the record list: this wave's open chunk (emit3_kernels.hip, DETECT) ----
    uint32_t chunk = 0;       // wave-uniform
    bool chunk_ok = false;    // wave-uniform: the chunk lies inside the list (else the records are only counted: the caller reruns with a longer list)
    auto take_chunk = [&]() {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(a.chunk_next, 1u);
        chunk = __builtin_amdgcn_readfirstlane(c);
        chunk_ok = chunk < a.chunk_cap;
    };
    auto rec_checkpoint = [&]() {   // (wave-uniform places only) the open chunk is closed once it is half full
        const uint32_t n = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile ldsp_u32 *>(static_cast<uintptr_t>(cur_at)));
        if (n > kEmit3Chunk / 2u) {
            if (lane == 0) {
                if (chunk_ok) a.chunk_fill[chunk] = n < kEmit3Chunk ? n : kEmit3Chunk;
                *reinterpret_cast<volatile ldsp_u32 *>(static_cast<uintptr_t>(cur_at)) = 0u;
            }
            take_chunk();
        }
    };
    if (lane == 0) *reinterpret_cast<volatile ldsp_u32 *>(static_cast<uintptr_t>(cur_at)) = 0u;
    take_chunk();
    // `p` = virtual position of the match's last byte; `counted`: the caller has added it to its tile's count already
    auto log_rec = [&](uint32_t p, uint32_t len, uint32_t value, bool counted) {
        if (p < a.emit_from) return;
        const uint32_t slot = atomicAdd(cursor, 1u);
        if (!counted) atomicAdd(&a.tile_deep[p >> 10], 1u);
        if (slot >= kEmit3Chunk) { atomicOr(a.fail, 2u); return; }
        if (chunk_ok) a.recs[static_cast<uint64_t>(chunk) * kEmit3Chunk + slot] = uint4{p, len, value, a.tile0 + (p >> 10)};
    };

    auto load_chunk = [&](uint32_t v) -> uint4 {
        if (v >= a.vlen) return uint4{0u, 0u, 0u, 0u};
        const px_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const px_u32x4_t *>(hay + v));
        return uint4{q.x, q.y, q.z, q.w};  // (bytes outside [lead, vlen) are whatever memory holds: starts there are masked out)
    };
    auto read_ahead = [&](uint32_t v) -> unsigned long long {
        unsigned long long x;
        if (v + 8 <= a.vlen) {
            __builtin_memcpy(&x, hay + v, 8);
        } else {
            x = 0;
            for (int b = 7; b >= 0; --b) x = (x << 8) | ((v + b < a.vlen) ? hay[v + b] : 0u);
        }
        return x;
    };

    // the queued branches: goto-only over the double array, W side by side per lane; every state that ends a pattern logs it
    auto drain = [&]() {
        constexpr int W = 4;
        if (wq_n != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (what one lane stored to the slab another lane reads back)
        for (uint32_t base_i = 0; base_i < wq_n; base_i += 64u * W) {
            uint32_t vn[W], st[W], b[W], n_ahead[W];
            unsigned long long ah[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t i = base_i + 64u * w + lane;
                uint4 e = uint4{0u, 0u, 0u, 0u};
                if (i < wq_n) e = slab[i];
                st[w] = e.x;
                vn[w] = e.x + G;  // the next byte to take
                b[w] = e.y;
                n_ahead[w] = 8;
                ah[w] = (static_cast<unsigned long long>(e.w) << 32) | e.z;
                if (vn[w] >= a.vlen) b[w] = 0;
            }
            for (;;) {
                rec_checkpoint();   // (at most 64 W records per turn: the open chunk has room for 512)
                uint4 r[W];
                bool any = false;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    r[w] = uint4{0u, 0u, 0u, 0u};
                    if (b[w] != 0) {
                        if (n_ahead[w] == 0) { ah[w] = read_ahead(vn[w]); n_ahead[w] = 8; }
                        r[w] = g.wrec_x[b[w] ^ (static_cast<uint32_t>(ah[w]) & 0xffu)];
                        any = true;
                    }
                }
                if (!__any(any)) break;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (b[w] == 0) continue;
                    const uint32_t c = static_cast<uint32_t>(ah[w]) & 0xffu;
                    if ((r[w].y & 0xffu) != c) { b[w] = 0; continue; }
                    b[w] = r[w].x;
                    ++vn[w];
                    if (r[w].y >> 8) log_rec(vn[w] - 1u, vn[w] - st[w], r[w].w, false);   // the match ends with the byte just taken
                    ah[w] >>= 8;
                    --n_ahead[w];
                    if (vn[w] >= a.vlen) b[w] = 0;
                }
            }
        }
        wq_n = 0;
    };

    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running
    uint32_t posbias0 = 0, posbias1 = 0;  // per slot: (virtual position of a byte) - (its LDS address)
    uint4 pend = uint4{0u, kPfxEmpty, 0u, 0u};
    uint32_t pend_pos = 0, pend_k0 = 0, pend_k1 = 0, pend_t0 = 0, pend_t1 = 0;
    bool pend_valid = false;           // wave-uniform
    uint32_t acc_tile = 0xffffffffu, acc_cnt = 0;   // lanes 0 .. 3: a direct-mapped cache of tile counts
    auto consume_pending = [&]() {
        if (!pend_valid) return;
        pend_valid = false;
        const uint4 r = pend;
        const bool match = r.x == pend_k0 && (r.y & 0x8000ffffu) == pend_k1;
        const uint32_t p = pend_pos + (G - 1u);
        const bool own = match && ((r.y >> 16) & 0x3fffu) != 0 && p >= a.emit_from;
        {
            const uint32_t my_tile = p >> 10;
            unsigned long long om = __ballot(own);
            while (om != 0) {
                const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(om));
                const uint32_t t0 = __builtin_amdgcn_readlane(my_tile, leader);
                const unsigned long long same = __ballot(own && my_tile == t0);
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
        if (own) log_rec(p, G, r.w, true);
        const bool go = match && r.z != 0;
        const unsigned long long m = __ballot(go);
        if (m != 0) {
            if (go) {
                const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                slab[at] = uint4{pend_pos, r.z, pend_t0, pend_t1};
            }
            wq_n += static_cast<uint32_t>(__popcll(m));
        }
    };
    auto process_batch = [&](uint32_t n) {  // n <= 64 entries from the head of the queue
        __builtin_amdgcn_s_setprio(2);
        rec_checkpoint();
        consume_pending();
        pend = uint4{0u, kPfxEmpty, 0u, 0u};
        if (lane < n) {
            const uint32_t e = lds_u32(ringb + (((q_head + lane) & (kRingP - 1u)) << 2));
            pend_pos = e + ((e - tb) >= SLOT ? posbias1 : posbias0);
            const uint32_t a0 = e & ~3u, sh = e & 3u;
            const uint32_t d0 = lds_u32(a0), d1 = lds_u32(a0 + 4u), d2 = lds_u32(a0 + 8u), d3 = lds_u32(a0 + 12u);
            const uint32_t x0 = __builtin_amdgcn_alignbyte(d1, d0, sh), x1 = __builtin_amdgcn_alignbyte(d2, d1, sh), x2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
            const uint32_t key0 = x0 & KMASK0, key1 = x1 & KMASK1;
            pend_k0 = key0;
            pend_k1 = key1;
            if (G == 4) { pend_t0 = x1; pend_t1 = x2; }
            else if (G < 4) { pend_t0 = __builtin_amdgcn_alignbyte(x1, x0, G & 3); pend_t1 = __builtin_amdgcn_alignbyte(x2, x1, G & 3); }
            else {
                const uint32_t d4 = lds_u32(a0 + 16u);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
                pend_t0 = __builtin_amdgcn_alignbyte(x2, x1, G & 3);
                pend_t1 = __builtin_amdgcn_alignbyte(x3, x2, G & 3);
            }
            const uint32_t mb = key0 * kPfxMulBucket0 + (key1 ^ g.seed) * kPfxMulBucket1;
            const uint32_t ms = key0 * kPfxMulSlot0 + (key1 ^ g.seed) * kPfxMulSlot1;
            const uint32_t bucket = __umulhi(mb, g.buckets);
            const uint32_t d = lds_u16(g.off_disp + (bucket << 1));
            pend = g.slots_e[pfx_slot(ms, d, g.n_slots)];
        }
        q_head += n;
        pend_valid = true;
    };

    uint32_t sl = 0;          // slot of the current step (wave-uniform)
    uint32_t carry_in = 0;    // queued entries that belong to the step before the current one
    const uint32_t start_end = a.vlen >= static_cast<uint32_t>(G) ? a.vlen - G + 1 : 0;   // starts are valid in [lead, vlen - G]
    for (uint32_t region = wave_global; region < a.nregions; region += nwaves) {
        const uint32_t rbase = region * a.region_bytes;
        const uint32_t rend = rbase + a.region_bytes < a.vlen ? rbase + a.region_bytes : a.vlen;
        auto fetch = [&](uint32_t s0) -> uint4 {
            if (s0 < rend) return load_chunk(s0 + lane * P);
            if (lane == 0 && s0 < rend + SB) return load_chunk(s0);
            return uint4{0u, 0u, 0u, 0u};
        };
        uint4 pf0 = fetch(rbase), pf1 = fetch(rbase + SB);
        for (uint32_t sb = rbase; sb < rend; sb += SB) {
            if (wq_n + 64u * P + 128u > a.wq_slab) drain();
            const uint32_t v = sb + lane * P;
            const uint4 cur = pf0;
            pf0 = pf1;
            rec_checkpoint();
            consume_pending();  // before the next chunk is requested: loads retire in order
            __builtin_amdgcn_s_setprio(0);
            pf1 = fetch(sb + 2u * SB);

            const uint32_t slot = tb + sl * SLOT;                 // wave-uniform
            const uint32_t my_text = slot + lane * P;
            {
                const uint32_t bias = sb - slot;
                if (sl) posbias1 = bias; else posbias0 = bias;
            }
            *reinterpret_cast<ldsp_u32x4 *>(static_cast<uintptr_t>(my_text)) = px_u32x4_t{cur.x, cur.y, cur.z, cur.w};
            if (lane == 0) *reinterpret_cast<ldsp_u32x4 *>(static_cast<uintptr_t>(slot + SB)) = px_u32x4_t{pf0.x, pf0.y, pf0.z, pf0.w};

            uint32_t W[6] = {cur.x, cur.y, cur.z, cur.w, 0u, 0u};
            W[4] = wave_shl1_p(cur.x, __builtin_amdgcn_readfirstlane(pf0.x));
            W[5] = wave_shl1_p(cur.y, __builtin_amdgcn_readfirstlane(pf0.y));
            uint32_t H = 0, H1 = 0;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int q = j >> 2, r = j & 3;
                uint32_t k0 = r == 0 ? W[q] : __builtin_amdgcn_alignbyte(W[q + 1], W[q], static_cast<uint32_t>(r));
                if (G < 4) k0 &= KMASK0;
                uint32_t m = k0 * kPfxMulBloom0;
                if (G > 4) {
                    uint32_t k1 = r == 0 ? W[q + 1] : __builtin_amdgcn_alignbyte(W[q + 2], W[q + 1], static_cast<uint32_t>(r));
                    k1 &= KMASK1;
                    m += k1 * kPfxMulBloom1;
                }
                const uint32_t word = lds_u32(__umulhi(m, g.bloom_words) << 2);   // BLOOM sits at LDS offset 0
                const uint32_t m2 = m * kPfxMulBits;
                H |= (__builtin_amdgcn_ubfe(word, m2 >> kPfxBit1, 1) & __builtin_amdgcn_ubfe(word, m2 >> kPfxBit2, 1)) << j;
                if (LEN1) {
                    const uint32_t byte = (W[q] >> (8 * r)) & 0xffu;
                    H1 |= (lds_u16(g.off_cnt1 + (byte << 1)) != 0 ? 1u : 0u) << j;
                }
            }
            // starts before the haystack's first byte or too close to its end do not count (first / last step only)
            if (v < a.lead || sb + SB + G > a.vlen + 1) {
                const uint32_t lo = a.lead > v ? a.lead - v : 0, hi = start_end > v ? start_end - v : 0;
                uint32_t keep = hi >= 16 ? 0xffffu : ((1u << hi) - 1u);
                keep &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
                H &= keep;
            }
            if (LEN1) {   // one-byte matches of this tile: positions in [emit_from, vlen)
                if (sb < a.emit_from || sb + SB > a.vlen) {
                    const uint32_t lo = a.emit_from > v ? a.emit_from - v : 0, hi = a.vlen > v ? a.vlen - v : 0;
                    uint32_t keep = hi >= 16 ? 0xffffu : ((1u << hi) - 1u);
                    keep &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
                    H1 &= keep;
                }
                uint32_t c1 = __popc(H1);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) c1 += __shfl_down(c1, off, 64);
                if (lane == 0) a.tile_short[sb >> 10] = c1;
            }

            bool did_batch = false;
            for (;;) {
                const bool has = H != 0;
                const unsigned long long m = __ballot(has);
                if (m == 0) break;
                if (has) {
                    const uint32_t b = static_cast<uint32_t>(__builtin_ctz(H));
                    H &= H - 1u;
                    const uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                    *reinterpret_cast<ldsp_u32 *>(static_cast<uintptr_t>(ringb + ((at & (kRingP - 1u)) << 2))) = my_text + b;
                }
                q_tail += static_cast<uint32_t>(__popcll(m));
                if (q_tail - q_head >= 64u) { process_batch(64u); did_batch = true; }
            }
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
        const uint32_t n = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile ldsp_u32 *>(static_cast<uintptr_t>(cur_at)));
        if (lane == 0 && chunk_ok) a.chunk_fill[chunk] = n < kEmit3Chunk ? n : kEmit3Chunk;
    }
}

template <int G, bool LEN1>
static hipError_t launch_pfx_emit_inst(const PfxDev &dev, const Emit3Args &a, uint32_t blocks, hipStream_t stream) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pfx_emit_kernel<G, LEN1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(dev.lds_bytes));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pfx_emit_kernel<G, LEN1>), dim3(blocks), dim3(1024), dev.lds_bytes, stream, dev, a);
    return hipGetLastError();
}
template <int G>
static hipError_t launch_pfx_emit_g(const PfxDev &dev, const Emit3Args &a, uint32_t blocks, hipStream_t stream) {
    return dev.has_len1 ? launch_pfx_emit_inst<G, true>(dev, a, blocks, stream) : launch_pfx_emit_inst<G, false>(dev, a, blocks, stream);
}
hipError_t launch_pfx_emit_detect(const PfxDev &dev, const Emit3Args &a, uint32_t blocks, hipStream_t stream) {
    switch (dev.G) {
        case 2: return launch_pfx_emit_g<2>(dev, a, blocks, stream);
        case 3: return launch_pfx_emit_g<3>(dev, a, blocks, stream);
        case 4: return launch_pfx_emit_g<4>(dev, a, blocks, stream);
        case 5: return launch_pfx_emit_g<5>(dev, a, blocks, stream);
        default: return launch_pfx_emit_g<6>(dev, a, blocks, stream);
    }
}

template <int G, bool LEN1, bool EXACT>
static hipError_t launch_pfx_inst(const PfxDev &dev, const GramArgs &a, uint32_t blocks, hipStream_t stream) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pfx_kernel<G, LEN1, EXACT, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(dev.lds_bytes));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pfx_kernel<G, LEN1, EXACT, 1024>), dim3(blocks), dim3(1024), dev.lds_bytes, stream, dev, a);
    return hipGetLastError();
}
template <int G>
static hipError_t launch_pfx_g(const PfxDev &dev, const GramArgs &a, bool exact, uint32_t blocks, hipStream_t stream) {
    if (exact) return dev.has_len1 ? launch_pfx_inst<G, true, true>(dev, a, blocks, stream) : launch_pfx_inst<G, false, true>(dev, a, blocks, stream);
    return dev.has_len1 ? launch_pfx_inst<G, true, false>(dev, a, blocks, stream) : launch_pfx_inst<G, false, false>(dev, a, blocks, stream);
}

// How many positions of a text survive the filter: 64 samples of 1 KiB, evenly spaced, one workgroup.  The engine is a FILTER: on text
// whose G-grams are mostly trie prefixes (a tokenizer vocabulary over its own language, a dictionary with every character as a pattern)
// every position is a survivor with a gather and a walk of its own, and the chain walker over the double array — one transition per
// byte whatever the text — is faster (tools/ab_wide.py: o200k-like 46 against 129 GB/s).  scan_count_impl asks before it chooses.
__global__ __launch_bounds__(1024) void pfx_probe_kernel(const PfxDev g, const uint8_t *__restrict__ hay, uint64_t len, unsigned int *__restrict__ out) {
    __shared__ unsigned int tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    const uint64_t stride = len / 64u;
    const uint64_t base = static_cast<uint64_t>(threadIdx.x >> 4) * stride + (threadIdx.x & 15u) * 64u;
    unsigned int c = 0;
    for (uint32_t j = 0; j < 64u; ++j) {
        const uint64_t v = base + j;
        if (v + g.G > len) break;
        uint32_t k0 = 0, k1 = 0;
        for (uint32_t b = 0; b < g.G; ++b) {
            const uint32_t x = hay[v + b];
            if (b < 4) k0 |= x << (8u * b); else k1 |= x << (8u * (b - 4u));
        }
        const uint32_t m = k0 * kPfxMulBloom0 + k1 * kPfxMulBloom1;
        const uint32_t word = g.bloom[__umulhi(m, g.bloom_words)];
        const uint32_t m2 = m * kPfxMulBits;
        c += (word >> (m2 >> kPfxBit1)) & (word >> ((m2 >> kPfxBit2) & 31u)) & 1u;
    }
    atomicAdd(&tot, c);
    __syncthreads();
    if (threadIdx.x == 0) *out = tot;
}
// survivors among 65 536 sampled positions of hay[0, len) (len >= 128 KiB), left in *out (device or page-locked memory)
hipError_t launch_pfx_probe(const PfxDev &dev, const uint8_t *hay, uint64_t len, unsigned int *out, hipStream_t stream) {
    hipLaunchKernelGGL(pfx_probe_kernel, dim3(1), dim3(1024), 0, stream, dev, hay, len, out);
    return hipGetLastError();
}

// LDS plan: BLOOM at 0, DISP, CNT1, then 16 waves x (two text slots + the survivor queue)
bool pfx_plan(PfxDev &d, uint32_t lds_limit) {
    d.off_disp = d.bloom_bytes;
    d.off_cnt1 = d.off_disp + d.disp_bytes;
    d.off_wave = d.off_cnt1 + 512u + 1024u;   // CNT1 (u16 x 256), then the h32 sums of the one-byte patterns (u32 x 256, count + checksum only)
    d.wave_stride = 2u * (1024u + 32u) + kRingP * 4u;
    d.lds_bytes = d.off_wave + 16u * d.wave_stride;
    d.threads = 1024;
    return d.lds_bytes <= lds_limit;
}

hipError_t launch_pfx_scan(const PfxDev &dev, const GramArgs &a, bool exact, uint32_t blocks, hipStream_t stream) {
    switch (dev.G) {
        case 2: return launch_pfx_g<2>(dev, a, exact, blocks, stream);
        case 3: return launch_pfx_g<3>(dev, a, exact, blocks, stream);
        case 4: return launch_pfx_g<4>(dev, a, exact, blocks, stream);
        case 5: return launch_pfx_g<5>(dev, a, exact, blocks, stream);
        default: return launch_pfx_g<6>(dev, a, exact, blocks, stream);
    }
}

}  // namespace daac
