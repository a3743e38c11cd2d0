// Restart scanners of the charwise automaton for gfx950: FindIterator (reference
// src/charwise/iter.rs:101-157) and LeftmostFindIterator (iter.rs:306-400) on the device.
//
// Same decomposition as restart_kernels.hip: the haystack is cut at *sync points* — character
// boundaries where the classic Aho-Corasick state of the text so far is ROOT — and the text
// between two sync points is scanned by one lane exactly as the reference scans a haystack of its
// own.  What differs from the bytewise file: symbols are UTF-8 scalars mapped to dense codes
// (charwise/mapper.rs), CHECK names the parent slot (charwise.rs:1022-1050), and positions move by
// whole characters while staying byte offsets.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "chain_scan.hpp"
#include "device_tables.hpp"

namespace daac {

__device__ __forceinline__ uint64_t cw_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

struct CwState { uint32_t idx, base, fail, opos, filt; };  // filt (the child filter): only kept by the micro-step walker

// MAPLDS: ASCII and the populated stretch [map_lo, table_len) of the code mapper are staged in LDS as u16 (0xffff =
// unmapped): one L2 round trip less per character; the code points in between (rare in CJK text) go to the L2 copy
template <int LVL>  // what is staged in LDS: 0 nothing, 1 the code mapper, 2 the mapper and ROOT's row of children
struct CwTablesT {
    static constexpr bool MAPLDS = LVL >= 1, ROWLDS = LVL >= 2;
    using State = CwState;
    using Stream = HayStream;
    static constexpr bool kMicro = DAAC_CW_MICRO != 0;  // chain_scan.hpp: the walker takes the transition one memory round trip at a time
    const CharDev &d;
    uint4 root_rec;
    const uint8_t *__restrict__ hay;
    uint64_t len;  // real end of the haystack: nothing at or beyond it is read
    const uint16_t *l_map = nullptr;
    const uint2 *l_row = nullptr;  // CharDev::root_row in LDS

    // (word 3 of a walkers' record: output_pos | child filter << obits; the plain records have no filter bits)
    __device__ __forceinline__ uint32_t opos_of(uint32_t w) const { return d.fbits ? (w & ((1u << d.obits) - 1u)) : w; }
    __device__ __forceinline__ uint32_t filt_of(uint32_t w) const { return d.fbits ? (w >> d.obits) : 0xffffffffu; }
    __device__ __forceinline__ CwState root() const { return CwState{0, root_rec.x, root_rec.z, opos_of(root_rec.w), filt_of(root_rec.w)}; }
    // the automaton as chain_scan.hpp wants it
    // one scalar through the lane's haystack window (same decoding as scalar_at below)
    __device__ __forceinline__ uint32_t symbol_at(HayWindow &win, uint64_t pos, uint32_t &clen) const {
        const uint32_t b0 = win.byte_at(hay + pos);
        if (b0 < 0x80u) { clen = 1; return b0; }
        const uint32_t n = b0 < 0xe0u ? 2u : b0 < 0xf0u ? 3u : 4u;
        uint32_t cp = b0 < 0xe0u ? (b0 & 0x1fu) : b0 < 0xf0u ? (b0 & 0x0fu) : (b0 & 0x07u);
        for (uint32_t k = 1; k < n; ++k) cp = (cp << 6) | (pos + k < len ? (win.byte_at(hay + pos + k) & 0x3fu) : 0u);
        clen = n;
        return cp;
    }
    __device__ __forceinline__ uint32_t opos(const CwState &st) const { return st.opos; }
    __device__ __forceinline__ bool is_root(const CwState &st) const { return st.idx == 0; }
    __device__ __forceinline__ uint64_t boundary_at_or_after(uint64_t x) const {
        while (x < len && (hay[x] & 0xc0u) == 0x80u) ++x;
        return x;
    }

    // One scalar at byte `pos` (a character boundary of well-formed UTF-8; charwise/iter.rs:64-98).
    // A sequence cut by the end of the haystack is completed with zero payload bits, never read past.
    __device__ __forceinline__ uint32_t scalar_at(uint64_t pos, uint32_t &clen) const {
        const uint32_t b0 = hay[pos];
        if (b0 < 0x80u) { clen = 1; return b0; }
        const uint32_t n = b0 < 0xe0u ? 2u : b0 < 0xf0u ? 3u : 4u;
        uint32_t cp = b0 < 0xe0u ? (b0 & 0x1fu) : b0 < 0xf0u ? (b0 & 0x0fu) : (b0 & 0x07u);
        for (uint32_t k = 1; k < n; ++k) cp = (cp << 6) | (pos + k < len ? (hay[pos + k] & 0x3fu) : 0u);
        clen = n;
        return cp;
    }
    __device__ __forceinline__ uint32_t code_of(uint32_t cp) const {
        if (MAPLDS) {  // l_map: 128 entries for ASCII, then the stretch [map_lo, table_len)
            const uint32_t rel = cp - d.map_lo;
            const bool low = cp < 128u, high = rel < d.table_len - d.map_lo;
            if (low || high) {
                const uint32_t c = l_map[low ? cp : rel + 128u];
                return c == 0xffffu ? 0xffffffffu : c;
            }
            // between ASCII and the stretch: rare, from L2 — asked and waited for in one piece, so that the optimiser does
            // not, on the common path, wait for a load it would otherwise believe might be outstanding
            uint32_t c = 0xffffffffu;
            if (cp < d.table_len) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(d.table + cp) : "memory");
            return c;
        }
        return cp < d.table_len ? d.table[cp] : 0xffffffffu;
    }
    // The code of the scalar that begins at p (a character boundary of well-formed UTF-8; charwise/iter.rs:64-98) and its
    // length in bytes; `avail` bytes are left in the haystack: a sequence cut by its end is completed with zero payload bits.
    __device__ __forceinline__ uint32_t symbol_code(HayStream &win, uint32_t pos, uint32_t avail, uint32_t &clen) const {
        uint32_t x = win.word_at(pos);
        if (__builtin_amdgcn_ballot_w64(avail < 4u) != 0) {  // (only at the very end of the haystack: skipped by the whole wave otherwise)
            if (avail < 4u) x &= (1u << (8u * avail)) - 1u;
        }
        const uint32_t b0 = x & 0xffu;
        // one formula for the four lengths: the lead byte's payload over three 6-bit groups, shifted down by the groups not there
        const uint32_t m2 = b0 >= 0x80u, m3 = b0 >= 0xe0u, m4 = b0 >= 0xf0u;
        const uint32_t n = 1u + m2 + m3 + m4;
        const uint32_t lead = b0 & (0xffu >> (n + m2));  // 0x7f, 0x1f, 0x0f, 0x07
        const uint32_t tail = (((x >> 8) & 0x3fu) << 12) | (((x >> 16) & 0x3fu) << 6) | ((x >> 24) & 0x3fu);
        clen = n;
        return code_of(((lead << 18) | tail) >> (24u - 6u * n));
    }
    // One memory round trip of the transition on `code` (charwise.rs:1022-1050 / 1056-1092 taken apart): a probe of the child
    // slot, or — after a failed probe (phase 1), or at once when the state's child filter rules the child out — the record the
    // failure link leads to; a link to ROOT needs no memory (ROOT's row, or its record, is at hand), a DEAD link ends the walk.
    // True once the transition is complete.  Every lane loads, every turn (an idle lane asks for slot 0), and the outcome is a
    // handful of selects.
    template <bool LM>
    __device__ __forceinline__ bool micro(CwState &st, uint32_t code, uint32_t &phase, bool act) const {
        const bool known = act && code != 0xffffffffu;  // charwise.rs:1031-1035
        const bool at_root = st.idx == 0;
        const bool possible = st.base != 0 && ((st.filt >> (code & (d.fbits ? d.fbits - 1u : 0u))) & 1u) != 0;
        // with ROOT's row at hand a lane standing at ROOT asks memory nothing
        const bool probe = known && phase == 0 && possible && !(ROWLDS && at_root);
        const bool no_child = known && !probe;
        const bool stop = LM && st.fail == 1u;
        const bool follow = no_child && !at_root && !stop && st.fail != 0;
        const uint32_t slot = probe ? (st.base ^ code) : follow ? st.fail : 0u;
        uint2 e = uint2{2u << 30, 0u};
        if (ROWLDS) e = l_row[known ? code : 0u];
        // the turn's one memory round trip: all four words in one request, and the turn's one full wait with it
        typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
        U32x4 rv;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(rv) : "v"(d.wstates + slot) : "memory");
        const uint4 r = uint4{rv.x, rv.y, rv.z, rv.w};
        uint32_t f = r.z;
        if (!LM && d.fail_plain) f = d.fail_plain[slot];
        const bool hit = probe && r.y == st.idx;
        const bool fell = no_child || (probe && !hit);             // no child on this symbol
        const bool take = hit || follow;                           // the record read becomes the state
        const bool dead = fell && !at_root && stop;
        const bool rootward = fell && !dead && (at_root || st.fail == 0);  // the symbol is ROOT's to take
        const bool by_row = ROWLDS && rootward;
        const bool child = by_row && (e.x >> 30) != 2u;
        const bool to_root = (act && !known) || dead || (rootward && !child);
        const bool done = (act && !known) || hit || dead || by_row || (fell && at_root);
        phase = (probe && !hit && !dead && !rootward) ? 1u : 0u;   // a failed probe whose link leads on: that record next turn
        const CwState rt = root();
        st.idx = take ? slot : child ? (rt.base ^ code) : to_root ? rt.idx : st.idx;
        st.base = take ? r.x : child ? (e.x & 0x3fffffffu) : to_root ? rt.base : st.base;
        st.fail = take ? f : child ? (e.x >> 30) : to_root ? rt.fail : st.fail;
        const uint32_t w = take ? r.w : e.y;
        st.opos = (take || child) ? opos_of(w) : to_root ? rt.opos : st.opos;
        st.filt = (take || child) ? filt_of(w) : to_root ? rt.filt : st.filt;
        return done;
    }
    __device__ __forceinline__ void load(CwState &st, uint32_t slot, bool plain) const {
        const uint4 r = d.states[slot];
        st = CwState{slot, r.x, (plain && d.fail_plain) ? d.fail_plain[slot] : r.z, r.w};
    }

    // classic delta: next_state_id_unchecked (charwise.rs:1022-1050) over links that never stop at DEAD
    __device__ __forceinline__ void step_plain(CwState &st, uint32_t cp) const {
        const uint32_t code = code_of(cp);
        if (code == 0xffffffffu) { st = root(); return; }
        for (;;) {
            if (st.base != 0) {
                const uint32_t child = st.base ^ code;
                const uint4 r = d.states[child];
                if (r.y == st.idx) { st = CwState{child, r.x, d.fail_plain ? d.fail_plain[child] : r.z, r.w}; return; }
            }
            if (st.idx == 0) return;
            load(st, st.fail, true);
        }
    }

    // next_state_id_leftmost_unchecked (charwise.rs:1056-1092): DEAD links end the walk at ROOT
    __device__ __forceinline__ void step_leftmost(CwState &st, uint32_t cp) const {
        const uint32_t code = code_of(cp);
        if (code == 0xffffffffu) { st = root(); return; }
        for (;;) {
            if (st.base != 0) {
                const uint32_t child = st.base ^ code;
                const uint4 r = d.states[child];
                if (r.y == st.idx) { st = CwState{child, r.x, r.z, r.w}; return; }
            }
            if (st.idx == 0) return;
            if (st.fail == 1u) { st = root(); return; }
            load(st, st.fail, false);
        }
    }

    // first sync point >= x
    __device__ __forceinline__ uint64_t sync_from(uint64_t x, uint32_t halo, uint64_t floor) const {
        if (x <= floor) return floor;  // the window start is a sync point by contract
        if (x >= len) return len;
        uint64_t pos = x > halo ? x - halo : 0;
        if (pos <= floor) pos = floor;
        else while (pos < len && (hay[pos] & 0xc0u) == 0x80u) ++pos;  // up to the next character boundary
        CwState st = root();
        uint32_t clen;
        while (pos < x) { const uint32_t cp = scalar_at(pos, clen); pos += clen; step_plain(st, cp); }
        while (st.idx != 0 && pos < len) { const uint32_t cp = scalar_at(pos, clen); pos += clen; step_plain(st, cp); }
        if (pos > len) pos = len;
        return st.idx == 0 ? pos : len;
    }
};
using CwTables = CwTablesT<0>;

// KMODE 0: totals {count, S1, S2}; 1: per-segment counts; 2: write matches at out + seg_counts[seg]
template <bool LEFTMOST, int KMODE>
__global__ __launch_bounds__(256) void char_restart_kernel(const CharDev dev, const ScanArgs a, unsigned long long *next_begin) {
    __shared__ unsigned long long scratch[3 * 4];
    const CwTables T{dev, dev.states[0], a.hay, a.total_len};
    const uint8_t *__restrict__ hay = a.hay;
    const uint64_t len = a.total_len;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;

    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        const bool positional = !LEFTMOST && dev.root_flag;
        uint64_t p = 0, q = 0;
        const bool final_seg = hi == a.len;
        if (!positional) {
            p = T.sync_from(lo, a.halo, a.begin);
            if (p < hi || final_seg) q = T.sync_from(hi, a.halo, a.begin);
        }
        if (final_seg && next_begin) *next_begin = positional ? hi : q;

        unsigned long long cnt = 0;
        uint32_t s1 = 0, s2 = 0;
        daac_match *o = nullptr;
        if (KMODE == 2) o = a.out + a.seg_counts[seg];
        auto emit = [&](uint32_t opos, uint64_t end) {
            const uint32_t *r = dev.outputs + 3u * (opos - 1u);
            const uint32_t value = r[0], length = r[1];
            if (KMODE == 2) {
                daac_match m;
                m.start = end - length; m.end = end; m.value = value; m._pad = 0;
                *o++ = m;
            } else {
                const uint32_t h = static_cast<uint32_t>(cw_mix64((static_cast<uint64_t>(value) << 32) | length));
                cnt += 1; s1 += h; s2 += h * static_cast<uint32_t>(end);
            }
        };

        if (positional) {
            // "" is a pattern: FindIterator reports (e, e, first "" value) at 0 and after every character
            // (iter.rs:115-131); ends in (lo, hi] belong to this segment
            const uint32_t op = T.root_rec.w;
            if (lo == 0) emit(op, 0);
            for (uint64_t e = lo + 1; e <= hi; ++e)
                if (e >= len || (hay[e] & 0xc0u) != 0x80u) emit(op, e);
        } else if (p < hi || (LEFTMOST && final_seg && p >= len && p == a.begin)) {  // (an empty range still ends the haystack)
            uint32_t clen;
            if (!LEFTMOST) {
                // FindIterator::next (iter.rs:133-156): back to ROOT after every match, report the list head
                CwState st = T.root();
                for (uint64_t pos = p; pos < q;) {
                    const uint32_t cp = T.scalar_at(pos, clen);
                    pos += clen;
                    T.step_plain(st, cp);
                    if (st.opos != 0) {
                        emit(st.opos, pos);
                        st = T.root();
                    }
                }
            } else {
                // LeftmostFindIterator::next (iter.rs:325-399), call by call, over the characters of [p, q).
                // `init` is the ROOT state's output_pos: non-zero when "" is a pattern, which then matches at
                // every position no longer match starts from (iter.rs:311-318); under that setting every walk
                // is anchored at `pos` (all failure links are DEAD) and has died by the sync point q.
                uint64_t pos = p;
                uint32_t init = T.root_rec.w;
                bool skip_empty = false;
                const bool real_end = q >= len;
                for (;;) {                       // one pass = one call of next()
                    if (!real_end && pos >= q) break;  // the owner of the next region continues from q
                    CwState st = T.root();
                    uint32_t best = init;        // last_output_pos
                    const uint32_t init_at_entry = init;
                    // One emit site per path: hipcc 7.2 -O3 loses the advance of the output cursor when emit() is
                    // inlined behind the nested break/continue of the literal transcription (seen in the ISA).
                    uint32_t ret_op = 0;         // what this call returns, if the walk dies on a character
                    uint64_t ret_end = 0;
                    bool again;
                    do {                         // the reference's loop 'a
                        again = false;
                        uint64_t i = pos, skips = 0;
                        while (i < q) {
                            const uint32_t cp = T.scalar_at(i, clen);
                            i += clen;
                            skips += clen;
                            T.step_leftmost(st, cp);
                            if (st.idx == 0) {
                                if (best != 0) {
                                    ret_end = pos;
                                    if (best != init) {
                                        skip_empty = true;
                                        ret_op = best;
                                    } else {
                                        pos += clen;
                                        if (skip_empty) { skip_empty = false; again = true; }
                                        else ret_op = best;
                                    }
                                    break;
                                }
                            } else if (st.opos != 0) {
                                best = st.opos;
                                pos += skips;
                                skips = 0;
                            }
                        }
                    } while (again);
                    if (ret_op != 0) { emit(ret_op, ret_end); continue; }
                    // the characters ran out (:385-398)
                    if (!real_end) {             // at a sync point only a match already seen can be pending
                        if (best != 0 && best != init_at_entry) { emit(best, pos); continue; }
                        break;
                    }
                    if (pos >= len) init = 0;
                    if (best == 0) break;        // None
                    if (best == init_at_entry && pos < len) {
                        // "" is a pattern and the haystack ends inside a longer one: the reference yields the
                        // same empty match forever from here (SURVEY 8a note D).  Reported, not imitated.
                        if (a.flags) atomicOr(a.flags, 1ull);
                        break;
                    }
                    emit(best, pos);
                }
            }
        }

        if (KMODE == 0) { tot_cnt += cnt; tot_s1 += s1; tot_s2 += s2; }
        else if (KMODE == 1) a.seg_counts[seg] = cnt;
    }

    if (KMODE == 0) {
        unsigned long long c = tot_cnt, x1 = tot_s1, x2 = tot_s2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); x1 += __shfl_down(x1, off, 64); x2 += __shfl_down(x2, off, 64); }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { scratch[wave * 3] = c; scratch[wave * 3 + 1] = x1; scratch[wave * 3 + 2] = x2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long r0 = 0, r1 = 0, r2 = 0;
            for (int w = 0; w < static_cast<int>((blockDim.x + 63) >> 6); ++w) { r0 += scratch[w * 3]; r1 += scratch[w * 3 + 1]; r2 += scratch[w * 3 + 2]; }
            if (r0 | r1 | r2) { atomicAdd(a.result, r0); atomicAdd(a.result + 1, r1); atomicAdd(a.result + 2, r2); }
        }
    }
}

// ---- speculate / reconcile / emit (chain_scan.hpp) over the charwise double array -------------------------
template <bool LEFTMOST, int PASS, int KMODE, int LVL>
__global__ __launch_bounds__(LVL == 2 ? 1024 : LVL == 1 ? 512 : 256) void char_chain_kernel(const CharDev dev, const ScanArgs a, const ChainArgs c,
                                                                                           unsigned long long *next_begin) {
    extern __shared__ __attribute__((aligned(16))) uint16_t l_map[];
    __shared__ unsigned long long scratch[3 * 16];
    const uint32_t n_map = 128u + (dev.table_len - dev.map_lo);
    uint2 *l_row = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(l_map) + ((n_map * 2u + 15u) & ~15u));
    if (LVL >= 1 && PASS != 3) {
        for (uint32_t i = threadIdx.x; i < n_map; i += blockDim.x) {
            const uint32_t cp = i < 128u ? i : i - 128u + dev.map_lo;
            const uint32_t code = cp < dev.table_len ? dev.table[cp] : 0xffffffffu;
            l_map[i] = code == 0xffffffffu ? 0xffffu : static_cast<uint16_t>(code);
        }
        if (LVL >= 2)
            for (uint32_t i = threadIdx.x; i < dev.alphabet; i += blockDim.x) l_row[i] = dev.root_row[i];
        __syncthreads();
    }
    const CwTablesT<LVL> T{dev, dev.wstates[0], a.hay, a.total_len, l_map, l_row};
    if (PASS == 0) chain_spec_body<CwTablesT<LVL>, LEFTMOST>(T, a, c, dev.ohash);
    else if (PASS == 1) chain_fix_body<CwTablesT<LVL>, LEFTMOST>(T, a, c, dev.ohash);
    else if (PASS == 3) chain_sum_body<KMODE>(a, c, next_begin, scratch);
    else chain_emit_body<CwTablesT<LVL>, LEFTMOST, KMODE>(T, a, c, dev.outputs, next_begin, scratch);
}

template <int LVL>
static hipError_t launch_char_chain_ml(const CharDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, bool leftmost,
                                       unsigned long long *next_begin, uint32_t blocks, hipStream_t stream) {
    // `blocks` counts 256-lane workgroups (8 per CU).  With the mapper in LDS (<= 32 KB) four 512-lane workgroups share a
    // CU, with ROOT's row as well (<= 80 KB together) two 1024-lane ones: the same 2048 lanes per CU every time.
    constexpr uint32_t per = LVL == 2 ? 4u : LVL == 1 ? 2u : 1u;
    const dim3 g((blocks + per - 1u) / per), b(256u * per);
    const uint32_t map_bytes = ((128u + dev.table_len - dev.map_lo) * 2u + 15u) & ~15u;
    const uint32_t lds = pass == 3 ? 0u : LVL == 2 ? map_bytes + dev.alphabet * 8u : LVL == 1 ? map_bytes : 0u;
    if (LVL == 2 && lds > 48u * 1024u) {
        hipError_t e;
#define DAAC_ATTR(L, P, M)                                                                                                      \
    do {                                                                                                                        \
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(char_chain_kernel<L, P, M, LVL>),                           \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))) != hipSuccess) return e; \
    } while (0)
        if (pass == 0) { if (leftmost) DAAC_ATTR(true, 0, 0); else DAAC_ATTR(false, 0, 0); }
        else if (pass == 1) { if (leftmost) DAAC_ATTR(true, 1, 0); else DAAC_ATTR(false, 1, 0); }
        else if (pass == 2 && leftmost) { if (kmode == 0) DAAC_ATTR(true, 2, 0); else if (kmode == 1) DAAC_ATTR(true, 2, 1); else DAAC_ATTR(true, 2, 2); }
        else if (pass == 2) { if (kmode == 0) DAAC_ATTR(false, 2, 0); else if (kmode == 1) DAAC_ATTR(false, 2, 1); else DAAC_ATTR(false, 2, 2); }
#undef DAAC_ATTR
    }
#define DAAC_CC(L, P, M) hipLaunchKernelGGL((char_chain_kernel<L, P, M, LVL>), g, b, lds, stream, dev, a, c, next_begin)
    if (pass == 0) { if (leftmost) DAAC_CC(true, 0, 0); else DAAC_CC(false, 0, 0); }
    else if (pass == 1) { if (leftmost) DAAC_CC(true, 1, 0); else DAAC_CC(false, 1, 0); }
    else if (pass == 3) { if (kmode == 0) DAAC_CC(false, 3, 0); else DAAC_CC(false, 3, 1); }
    else if (leftmost) { if (kmode == 0) DAAC_CC(true, 2, 0); else if (kmode == 1) DAAC_CC(true, 2, 1); else DAAC_CC(true, 2, 2); }
    else { if (kmode == 0) DAAC_CC(false, 2, 0); else if (kmode == 1) DAAC_CC(false, 2, 1); else DAAC_CC(false, 2, 2); }
#undef DAAC_CC
    return hipGetLastError();
}

// find_overlapping_iter().count() (+ checksum) of a charwise automaton with the micro-step walker (overlap_count_body)
template <int LVL, bool HEADS>
__global__ __launch_bounds__(LVL == 2 ? 1024 : LVL == 1 ? 512 : 256) void char_overlap_count_kernel(const CharDev dev, const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t l_map[];
    __shared__ unsigned long long scratch[3 * 16];
    const uint32_t n_map = 128u + (dev.table_len - dev.map_lo);
    uint2 *l_row = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(l_map) + ((n_map * 2u + 15u) & ~15u));
    if (LVL >= 1) {
        for (uint32_t i = threadIdx.x; i < n_map; i += blockDim.x) {
            const uint32_t cp = i < 128u ? i : i - 128u + dev.map_lo;
            const uint32_t code = cp < dev.table_len ? dev.table[cp] : 0xffffffffu;
            l_map[i] = code == 0xffffffffu ? 0xffffu : static_cast<uint16_t>(code);
        }
        if (LVL >= 2)
            for (uint32_t i = threadIdx.x; i < dev.alphabet; i += blockDim.x) l_row[i] = dev.root_row[i];
        __syncthreads();
    }
    const CwTablesT<LVL> T{dev, dev.wstates[0], a.hay, a.total_len, l_map, l_row};
    overlap_count_body<CwTablesT<LVL>, HEADS>(T, a, dev.osum, dev.ohash, scratch);
}

template <int LVL, bool HEADS>
static hipError_t launch_char_overlap_ml(const CharDev &dev, const ScanArgs &a, uint32_t blocks, hipStream_t stream) {
    constexpr uint32_t per = LVL == 2 ? 4u : LVL == 1 ? 2u : 1u;
    const dim3 g((blocks + per - 1u) / per), b(256u * per);
    const uint32_t map_bytes = ((128u + dev.table_len - dev.map_lo) * 2u + 15u) & ~15u;
    const uint32_t lds = LVL == 2 ? map_bytes + dev.alphabet * 8u : LVL == 1 ? map_bytes : 0u;
    if (lds > 48u * 1024u) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(char_overlap_count_kernel<LVL, HEADS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((char_overlap_count_kernel<LVL, HEADS>), g, b, lds, stream, dev, a);
    return hipGetLastError();
}

// `blocks` counts 256-lane workgroups
hipError_t launch_char_overlap_count(const CharDev &dev, const ScanArgs &a, bool heads, uint32_t blocks, hipStream_t stream) {
    const int lvl = dev.map_in_lds != 0 ? (dev.row_in_lds != 0 ? 2 : 1) : 0;
    if (heads) return lvl == 2 ? launch_char_overlap_ml<2, true>(dev, a, blocks, stream) : lvl == 1 ? launch_char_overlap_ml<1, true>(dev, a, blocks, stream)
                                                                                                      : launch_char_overlap_ml<0, true>(dev, a, blocks, stream);
    return lvl == 2 ? launch_char_overlap_ml<2, false>(dev, a, blocks, stream) : lvl == 1 ? launch_char_overlap_ml<1, false>(dev, a, blocks, stream)
                                                                                           : launch_char_overlap_ml<0, false>(dev, a, blocks, stream);
}

hipError_t launch_char_chain(const CharDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, bool leftmost,
                             unsigned long long *next_begin, uint32_t blocks, hipStream_t stream) {
    if (dev.map_in_lds != 0 && dev.row_in_lds != 0) return launch_char_chain_ml<2>(dev, a, c, pass, kmode, leftmost, next_begin, blocks, stream);
    if (dev.map_in_lds != 0) return launch_char_chain_ml<1>(dev, a, c, pass, kmode, leftmost, next_begin, blocks, stream);
    return launch_char_chain_ml<0>(dev, a, c, pass, kmode, leftmost, next_begin, blocks, stream);
}

hipError_t launch_char_restart_scan(const CharDev &dev, const ScanArgs &a, int kmode, bool leftmost, unsigned long long *next_begin,
                                    uint32_t blocks, uint32_t threads, hipStream_t stream) {
    const dim3 g(blocks), b(threads > 256 ? 256 : threads);
#define DAAC_CW(L, M) hipLaunchKernelGGL((char_restart_kernel<L, M>), g, b, 0, stream, dev, a, next_begin)
    if (leftmost) { if (kmode == 0) DAAC_CW(true, 0); else if (kmode == 1) DAAC_CW(true, 1); else DAAC_CW(true, 2); }
    else { if (kmode == 0) DAAC_CW(false, 0); else if (kmode == 1) DAAC_CW(false, 1); else DAAC_CW(false, 2); }
#undef DAAC_CW
    return hipGetLastError();
}

}  // namespace daac
