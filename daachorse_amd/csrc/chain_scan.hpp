// Speculate / reconcile / emit: the parallel form of the restart iterators (FindIterator and
// LeftmostFindIterator of the reference, bytewise and charwise) for gfx950.
//
// Both iterators are a CHAIN: from a position r at ROOT they scan to the next match (s, e), report
// it and continue from e.  What they report after r depends on nothing but r, so the haystack can
// be cut wherever the chain passes — but where it passes is only known by following it.  Instead:
//
//   1. speculate   every lane follows the chain of its segment [lo, hi) as if it started fresh at
//                  lo, and records where that chain leaves the segment (its first position >= hi);
//   2. reconcile   the chain really enters segment k where segment k-1 was left.  A lane follows the
//                  true entry and its speculative chain in lockstep; as soon as both stand on the
//                  same position they are one chain from there on, so the speculative exit is the
//                  true exit.  Different starting points fall in step after a few matches, so one
//                  or two rounds settle every segment (a round re-reads only a few links per lane);
//   3. emit        with the true entries known the segments are independent.  Counts need no third pass:
//                  the speculative pass tallies what it reports and reconciliation records what the true
//                  chain reports differently before the two meet; materialising runs count per segment ->
//                  exclusive scan -> one write pass, as for the overlapping scan.
//
// A link also ends, without a match, once it is past its segment and the automaton is back at ROOT
// with nothing pending: a fresh start there is indistinguishable from going on (this keeps links
// short in text with few matches).  A link that neither matches nor returns to ROOT for `cap` bytes
// past its segment raises the overflow flag, and the driver falls back to the sync-point scanners
// (restart_kernels.hip), as it does for automata that contain the empty pattern.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

// settings of the round-2 chain-walker experiments (profiles/r02_bench_restart_scanners.txt), frozen at the measured best
#define DAAC_WIN_AHEAD 1
#define DAAC_TALLY_DEFER 1
#define DAAC_CW_MICRO 1

namespace daac {

// A lane's window on the haystack: the aligned 16-byte granule around the last byte it asked for.  A chain
// moves through the text byte by byte (and steps back a little after a leftmost match), so 15 of 16 requests
// are served from registers instead of the memory pipeline, where 64 lanes asking for 64 different lines cost
// 64 cycles each time.  A granule never crosses a page, so reading all of it is safe at either end of a buffer.
struct HayWindow {
    typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
    uintptr_t base = ~static_cast<uintptr_t>(0);
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    __device__ __forceinline__ uint32_t byte_at(const uint8_t *p) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(p), b = addr & ~static_cast<uintptr_t>(15);
        if (b != base) {
            const U32x4 v = *reinterpret_cast<const U32x4 __attribute__((address_space(1))) *>(b);
            w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
            base = b;
        }
        const uint32_t k = static_cast<uint32_t>(addr) & 15u;
        uint32_t a0 = w0, a1 = w1, a2 = w2, a3 = w3;  // (as values, not as four adjacent fields the optimiser may index in memory)
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        const uint32_t word = (k & 8u) ? ((k & 4u) ? a3 : a2) : ((k & 4u) ? a1 : a0);
        return (word >> ((k & 3u) * 8u)) & 0xffu;
    }
};

// The window of the walkers that take one memory round trip per turn (run_micro): 32 resident bytes [base, base + 32) and
// the 16 after them in flight.  word_at() slides the window when the four bytes asked for reach beyond it: the granule in
// flight was asked for a slide (16 bytes of text, several turns) ago and every turn ends with a full wait for the turn's
// own transition, so it has arrived — the optimiser's bookkeeping of outstanding loads sees that too and puts no wait of
// its own behind the new request.  A step back of up to 12 bytes after a leftmost match stays inside the window.
struct HayStream {
    typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
    const uint8_t *org = nullptr;  // positions are 32-bit offsets from here
    uintptr_t limit = 0;           // granules beginning at or beyond this address are never read
    uint32_t boff = 0x80000000u;   // offset of the window's first byte (far from any position: the first request is cold)
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, n0 = 0, n1 = 0, n2 = 0, n3 = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0;

    __device__ __forceinline__ uintptr_t addr_of(uint32_t off) const {
        return reinterpret_cast<uintptr_t>(org) + static_cast<uintptr_t>(static_cast<intptr_t>(static_cast<int32_t>(off)));
    }
    // a new run: positions count from `first`; nothing at or beyond `end` is read
    template <class T>
    __device__ __forceinline__ void open(const T &, const uint8_t *first, const uint8_t *end) {
        org = first;
        limit = reinterpret_cast<uintptr_t>(end);
        boff = 0x80000000u;
        cpos = 0x80000000u;
    }
    uint32_t cword = 0, cpos = 0x80000000u;  // byte_at: the last word fetched and where it begins
    // one byte; three of four requests are answered by the word the previous one fetched
    __device__ __forceinline__ uint32_t byte_at(uint32_t pos) {
        uint32_t dlt = pos - cpos;
        if (dlt > 3u) { cword = word_at(pos); cpos = pos; dlt = 0; }
        return (cword >> (8u * dlt)) & 0xffu;
    }
    // the four bytes from position `pos` on, first byte lowest; bytes at or beyond `limit`'s granule read as zero
    __device__ __forceinline__ uint32_t word_at(uint32_t pos) {
        uint32_t off = pos - boff;
        if (off > 28u) {
            if (off < 44u) {  // the usual step: slide by one granule
                w0 = n0; w1 = n1; w2 = n2; w3 = n3;
                n0 = f0; n1 = f1; n2 = f2; n3 = f3;
                boff += 16;
            } else {  // a segment's first request, or a step back beyond the window: two granules fetched and waited for
                const uintptr_t addr = addr_of(pos), b = addr & ~static_cast<uintptr_t>(15);
                const bool two = b + 16 < limit;
                const uintptr_t b2 = two ? b + 16 : b;
                U32x4 v0, v1;
                asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v0), "=&v"(v1) : "v"(b), "v"(b2) : "memory");
                w0 = v0.x; w1 = v0.y; w2 = v0.z; w3 = v0.w;
                n0 = two ? v1.x : 0u; n1 = two ? v1.y : 0u; n2 = two ? v1.z : 0u; n3 = two ? v1.w : 0u;
                boff = pos - (static_cast<uint32_t>(addr) & 15u);
            }
            const uintptr_t g = addr_of(boff + 32u);
            U32x4 v = U32x4{0u, 0u, 0u, 0u};
            if (g < limit) v = *reinterpret_cast<const U32x4 __attribute__((address_space(1))) *>(g);
            f0 = v.x; f1 = v.y; f2 = v.z; f3 = v.w;
            off = pos - boff;
        }
        // dwords q and q + 1 of the eight resident ones, q = off / 4 (off <= 28: the last word is only ever the low one)
        const bool up = (off & 16u) != 0;
        const uint32_t b0 = up ? n0 : w0, b1 = up ? n1 : w1, b2 = up ? n2 : w2, b3 = up ? n3 : w3, b4 = up ? n3 : n0;
        const bool mid = (off & 8u) != 0;
        const uint32_t c0 = mid ? b2 : b0, c1 = mid ? b3 : b1, c2 = mid ? b4 : b2;
        const bool odd = (off & 4u) != 0;
        return __builtin_amdgcn_alignbyte(odd ? c2 : c1, odd ? c1 : c0, off & 3u);
    }
};

// T supplies the automaton: State, root(), symbol_at(window, pos, clen), step_plain / step_leftmost, opos(), is_root().
template <class T, bool LEFTMOST>
struct ChainWalker {
    const T &t;
    uint64_t len;
    uint64_t cap;
    bool overflow = false;
    HayWindow win;
    typename T::Stream str;  // run_micro's window

    // One link of the chain from position r (reference bytewise/iter.rs:87-112 / 272-340, charwise/iter.rs:
    // 133-156 / 325-399, one call of next()); returns the next chain position, > r.
    template <class Emit>
    __device__ __forceinline__ uint64_t link(uint64_t r, uint64_t hi, Emit &&emit) {
        typename T::State st = t.root();
        uint64_t pos = r;
        uint32_t clen;
        if (!LEFTMOST) {
            for (;;) {
                if (pos >= len) return len;
                const uint32_t sym = t.symbol_at(win, pos, clen);
                pos += clen;
                t.step_plain(st, sym);
                if (t.opos(st) != 0) { emit(t.opos(st), pos); return pos; }
                if (pos >= hi) {
                    if (t.is_root(st)) return pos;
                    if (pos - hi > cap) { overflow = true; return pos; }
                }
            }
        } else {
            uint32_t best = 0;       // last_output_pos
            uint64_t best_end = r;   // self.pos
            for (;;) {
                if (pos >= len) {
                    if (best != 0) { emit(best, best_end); return best_end; }
                    return len;
                }
                const uint32_t sym = t.symbol_at(win, pos, clen);
                t.step_leftmost(st, sym);
                if (t.is_root(st)) {
                    if (best != 0) { emit(best, best_end); return best_end; }
                    pos += clen;
                    if (pos >= hi) return pos;
                } else {
                    pos += clen;
                    if (t.opos(st) != 0) { best = t.opos(st); best_end = pos; }
                    else if (best == 0 && pos >= hi && pos - hi > cap) { overflow = true; return pos; }
                }
            }
        }
    }

    // The chain from `entry` until it leaves [.., hi): `r = entry; while (r < hi) r = link(r)`, written as ONE loop that
    // takes one symbol per turn.  As nested loops (a loop over links around the loop over a link's symbols) the 64 lanes of a
    // wave re-converge at the end of every link: a lane whose link ended after three symbols idles until the longest link of
    // the wave is through, and with a match every few bytes most of the wave idles most of the time (the scan then runs at
    // a quarter of the rate at which the memory system answers).  Here a lane that reports a match goes straight on to the
    // first symbol of its next link in the same turn of the loop.
    template <class Emit>
    __device__ __forceinline__ uint64_t run_micro(uint64_t entry, uint64_t hi, Emit &&emit);

    template <class Emit>
    __device__ __forceinline__ uint64_t run(uint64_t entry, uint64_t hi, Emit &&emit) {
        if (entry >= hi || overflow) return entry;
        if constexpr (T::kMicro) {
            if (hi - entry >= 0x40000000ull) { overflow = true; return entry; }  // (segments are KBs; the driver falls back)
            return run_micro(entry, hi, emit);
        }
        typename T::State st = t.root();
        uint64_t pos = entry;
        uint32_t clen;
        if (!LEFTMOST) {
            for (;;) {
                if (pos >= len) return len;
                const uint32_t sym = t.symbol_at(win, pos, clen);
                pos += clen;
                t.step_plain(st, sym);
                const uint32_t op = t.opos(st);
                if (op != 0) {
                    emit(op, pos);
                    if (pos >= hi) return pos;
                    st = t.root();  // the next link
                } else if (pos >= hi) {
                    if (t.is_root(st)) return pos;
                    if (pos - hi > cap) { overflow = true; return pos; }
                }
            }
        } else {
            uint32_t best = 0;          // last_output_pos
            uint64_t best_end = entry;  // self.pos
            for (;;) {
                bool report = false;
                if (pos >= len) {
                    if (best == 0) return len;
                    report = true;
                } else {
                    const uint32_t sym = t.symbol_at(win, pos, clen);
                    t.step_leftmost(st, sym);
                    if (t.is_root(st)) {
                        if (best != 0) {
                            report = true;
                        } else {
                            pos += clen;
                            if (pos >= hi) return pos;
                        }
                    } else {
                        pos += clen;
                        const uint32_t op = t.opos(st);
                        if (op != 0) { best = op; best_end = pos; }
                        else if (best == 0 && pos >= hi && pos - hi > cap) { overflow = true; return pos; }
                    }
                }
                if (report) {  // the link ends with its longest / first match; the next one starts where that match ended
                    emit(best, best_end);
                    if (best_end >= hi) return best_end;
                    st = t.root();
                    pos = best_end;
                    best = 0;
                }
            }
        }
    }
};

// The same chain with the TRANSITION taken apart as well: a turn of the loop is exactly one memory round trip of the
// automaton (T::micro: the probe of a child slot, or the record a failure link leads to), for every lane.  With whole
// transitions per turn a wave waits for its slowest lane's failure walk — on a CJK dictionary 1.6 round trips per symbol
// on average but 4.5 for the slowest of 64 lanes — while here a lane that is through takes its next symbol at once.
// The turn is written as straight-line selects (the lanes of a wave are in every phase at once: nested branches would be
// executed all the same, plus the bookkeeping of their masks); positions are 32-bit offsets from `entry`.
template <class T, bool LEFTMOST>
template <class Emit>
__device__ __forceinline__ uint64_t ChainWalker<T, LEFTMOST>::run_micro(uint64_t entry, uint64_t hi, Emit &&emit) {
    const uint64_t room = len - entry;
    const uint32_t end32 = room > 0xffffff00ull ? 0xffffff00u : static_cast<uint32_t>(room);  // the haystack's end
    const uint32_t hi32 = static_cast<uint32_t>(hi - entry);
    const uint32_t cap32 = cap > 0x3fffffffull ? 0x3fffffffu : static_cast<uint32_t>(cap);
    str.open(t, t.hay + entry, t.hay + len);
    typename T::State st = t.root();
    uint32_t pos = 0, clen = 0, code = 0, phase = 0;
    bool pending = false;   // a symbol has been read and its transition is under way
    uint32_t best = 0;      // leftmost: last_output_pos
    uint32_t best_end = 0;  // leftmost: self.pos
    uint32_t ret = 0;
    for (;;) {
        bool fin = false, report = false;
        if (!pending) {
            if (pos >= end32) {  // out of text
                if (LEFTMOST && best != 0) report = true;
                else { ret = end32; fin = true; }
            } else {
                code = t.symbol_code(str, pos, end32 - pos, clen);
                phase = 0;
                pending = true;
            }
        }
        const bool done = t.template micro<LEFTMOST>(st, code, phase, pending);  // (nothing happens for a lane that is not pending)
        pending = pending && !done;
        const bool at_root = t.is_root(st);
        const uint32_t op = t.opos(st);
        if (!LEFTMOST) {
            const uint32_t npos = pos + clen;
            pos = done ? npos : pos;
            const bool match = done && op != 0;
            if (match) emit(op, entry + pos);
            const bool out = done && pos >= hi32;
            const bool lost = out && !match && !at_root && pos - hi32 > cap32;
            if (lost) overflow = true;
            if (out && (match || at_root || lost)) { ret = pos; fin = true; }
            if (match) st = t.root();  // the next link
        } else {
            const bool ends = done && at_root && best != 0;  // the walk died with a match in hand: the symbol is not taken
            report = report || ends;
            const bool adv = done && !ends;
            pos = adv ? pos + clen : pos;
            const bool better = adv && !at_root && op != 0;
            best_end = better ? pos : best_end;
            best = better ? op : best;
            const bool out = adv && pos >= hi32;
            const bool lost = out && !at_root && op == 0 && best == 0 && pos - hi32 > cap32;
            if (lost) overflow = true;
            if (out && (at_root || lost)) { ret = pos; fin = true; }
            if (report) {  // the link ends with its longest / first match; the next one starts where that match ended
                emit(best, entry + best_end);
                if (best_end >= hi32) { ret = best_end; fin = true; }
                st = t.root();
                pos = best_end;
                best = 0;
            }
        }
        if (fin) break;
    }
    return ret == end32 ? len : entry + ret;
}

// find_overlapping_iter().count() (+ checksum) by segment with the same one-round-trip-per-turn walker: a lane enters its
// segment `halo` bytes early at ROOT (classic Aho-Corasick transitions, nothing reported before lo) and tallies the output
// list of every state it passes with an end in (lo, hi] — the list's {count, sum of h32} come precomputed (osum), asked for
// when the state is entered and folded in at the next report.  T as for ChainWalker (T::micro<false> = the classic delta).
// HEADS: FindOverlappingNoSuffixIterator — only the head of a state's list (`ohash`: h of the record alone).
template <class T, bool HEADS>
__device__ __forceinline__ void overlap_count_body(const T &t, const ScanArgs &a, const uint2 *__restrict__ osum, const uint32_t *__restrict__ ohash,
                                                   unsigned long long *scratch) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;
    typename T::Stream str;
    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        const uint64_t p0 = t.boundary_at_or_after(lo > a.halo ? lo - a.halo : 0);
        if (p0 >= hi) continue;
        const uint32_t warm = lo > p0 ? static_cast<uint32_t>(lo - p0) : 0u;  // ends <= warm belong to the segment before
        const uint32_t end32 = static_cast<uint32_t>(hi - p0);
        const uint64_t room = a.total_len - p0;
        const uint32_t text32 = room > 0xffffff00ull ? 0xffffff00u : static_cast<uint32_t>(room);
        str.open(t, t.hay + p0, t.hay + a.total_len);
        typename T::State st = t.root();
        uint32_t pos = 0, clen = 0, code = 0, phase = 0;
        bool pending = false;
        unsigned long long cnt = 0;
        uint32_t s1 = 0, s2 = 0;
        uint2 q_wait = uint2{0u, 0u};
        uint32_t end_wait = 0;
        for (;;) {
            if (!pending && pos < end32) {
                code = t.symbol_code(str, pos, text32 - pos, clen);
                phase = 0;
                pending = true;
            }
            const bool done = t.template micro<false>(st, code, phase, pending);
            pending = pending && !done;
            pos = done ? pos + clen : pos;
            const uint32_t op = t.opos(st);
            if (done && op != 0 && pos > warm && pos <= end32) {  // (a character cut by hi ends in the next segment)
                cnt += q_wait.x; s1 += q_wait.y; s2 += q_wait.y * end_wait;
                if (HEADS) q_wait = uint2{1u, ohash[op - 1u]};
                else q_wait = osum[op - 1u];
                end_wait = static_cast<uint32_t>(p0) + pos;
            }
            if (!pending && pos >= end32) break;
        }
        cnt += q_wait.x; s1 += q_wait.y; s2 += q_wait.y * end_wait;
        tot_cnt += cnt; tot_s1 += s1; tot_s2 += s2;
    }
    unsigned long long cc = tot_cnt, x1 = tot_s1, x2 = tot_s2;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { cc += __shfl_down(cc, off, 64); x1 += __shfl_down(x1, off, 64); x2 += __shfl_down(x2, off, 64); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // (`scratch` may lie over tables the walkers of other waves were still using)
    if (lane == 0) { scratch[wave * 3] = cc; scratch[wave * 3 + 1] = x1; scratch[wave * 3 + 2] = x2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r0 = 0, r1 = 0, r2 = 0;
        for (int v = 0; v < static_cast<int>((blockDim.x + 63) >> 6); ++v) { r0 += scratch[v * 3]; r1 += scratch[v * 3 + 1]; r2 += scratch[v * 3 + 2]; }
        if (r0 | r1 | r2) { atomicAdd(a.result, r0); atomicAdd(a.result + 1, r1); atomicAdd(a.result + 2, r2); }
    }
}

struct ChainNoEmit {
    __device__ __forceinline__ void operator()(uint32_t, uint64_t) const {}
};

// {count, S1, S2} of the matches a chain reports (the checksum of include/daachorse_amd.h, daac_scan_count).  h of a match
// is a function of its output record alone and comes precomputed (`ohash`, one u32 per record); the value asked for at one
// report is folded in at the next one (or by packed()), so that the walker never waits for it: the answer arrives behind
// the transition's own memory round trip.
struct ChainTally {
    const uint32_t *ohash;
    unsigned long long cnt = 0;
    uint32_t s1 = 0, s2 = 0;
    uint32_t h_wait = 0, end_wait = 0;
    __device__ __forceinline__ void operator()(uint32_t opos, uint64_t end) {
#if DAAC_TALLY_DEFER
        s1 += h_wait; s2 += h_wait * end_wait;
        h_wait = ohash[opos - 1u];
        end_wait = static_cast<uint32_t>(end);
#else
        const uint32_t h = ohash[opos - 1u];
        s1 += h; s2 += h * static_cast<uint32_t>(end);
#endif
        cnt += 1;
    }
    __device__ __forceinline__ uint4 packed() const {
        return uint4{static_cast<uint32_t>(cnt), static_cast<uint32_t>(cnt >> 32), s1 + h_wait, s2 + h_wait * end_wait};
    }
};
__device__ __forceinline__ uint4 tally_sub(const uint4 &a, const uint4 &b) {  // a - b, field by field
    const unsigned long long ca = (static_cast<unsigned long long>(a.y) << 32) | a.x, cb = (static_cast<unsigned long long>(b.y) << 32) | b.x;
    const unsigned long long c = ca - cb;
    return uint4{static_cast<uint32_t>(c), static_cast<uint32_t>(c >> 32), a.z - b.z, a.w - b.w};
}

// pass 1: exits of the speculative chains
template <class T, bool LEFTMOST>
__device__ __forceinline__ void chain_spec_body(const T &t, const ScanArgs &a, const ChainArgs &c, const uint32_t *ohash) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    ChainWalker<T, LEFTMOST> w{t, a.total_len, c.cap};
    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        ChainTally tally{ohash};
        c.x_out[seg] = w.run(seg == 0 ? lo : t.boundary_at_or_after(lo), hi, tally);
        c.tally_spec[seg] = tally.packed();
    }
    if (w.overflow) c.flags[1] = 1u;
}

// pass 2 (repeated until nothing changes): exits given the previous round's exits as entries
template <class T, bool LEFTMOST>
__device__ __forceinline__ void chain_fix_body(const T &t, const ScanArgs &a, const ChainArgs &c, const uint32_t *ohash) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    ChainWalker<T, LEFTMOST> w{t, a.total_len, c.cap};
    bool changed = false;
    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        const uint64_t spec_exit = c.x_spec[seg];
        uint64_t exit = spec_exit;
        uint4 delta{0u, 0u, 0u, 0u};        // true chain minus speculative chain, in reported matches
        if (seg != 0) {
            const uint64_t entry = c.x_prev[seg - 1];
            uint64_t s = t.boundary_at_or_after(lo);
            if (entry >= hi) {
                exit = entry;               // the chain jumps over this segment: nothing of it is reported
                delta = tally_sub(delta, c.tally_spec[seg]);
            } else if (entry != s) {
                uint64_t x = entry;         // the true chain; s follows the speculative one
                ChainTally of_true{ohash}, of_spec{ohash};
                bool merged = false;
                while (x < hi && !w.overflow) {
                    while (s < x && s < hi && !w.overflow) s = w.link(s, hi, of_spec);
                    if (s == x) { merged = true; break; }
                    x = w.link(x, hi, of_true);
                }
                if (merged) {
                    delta = tally_sub(of_true.packed(), of_spec.packed());  // they differ only before they met
                } else {
                    exit = x;
                    delta = tally_sub(of_true.packed(), c.tally_spec[seg]);
                }
            }
        }
        if (exit != c.x_prev[seg]) changed = true;
        c.x_out[seg] = exit;
        c.tally_delta[seg] = delta;
    }
    if (changed) c.flags[0] = 1u;
    if (w.overflow) c.flags[1] = 1u;
}

// the tallies of the speculative pass, corrected by the last reconciliation round.  KMODE 0: totals, 1: per-segment counts
template <int KMODE>
__device__ __forceinline__ void chain_sum_body(const ScanArgs &a, const ChainArgs &c, unsigned long long *next_begin, unsigned long long *scratch) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;
    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint4 p = c.tally_spec[seg], d = c.tally_delta[seg];
        const unsigned long long cnt = ((static_cast<unsigned long long>(p.y) << 32) | p.x) + ((static_cast<unsigned long long>(d.y) << 32) | d.x);
        if (KMODE == 1) a.seg_counts[seg] = cnt;
        else { tot_cnt += cnt; tot_s1 += p.z + d.z; tot_s2 += p.w + d.w; }
        if (next_begin && seg + 1 == a.nseg) *next_begin = c.x_prev[seg] > a.len ? c.x_prev[seg] : a.len;  // where a following window starts
    }
    if (KMODE == 0) {
        unsigned long long cc = tot_cnt, x1 = tot_s1, x2 = tot_s2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { cc += __shfl_down(cc, off, 64); x1 += __shfl_down(x1, off, 64); x2 += __shfl_down(x2, off, 64); }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();  // (`scratch` may lie over tables the walkers of other waves were still using)
        if (lane == 0) { scratch[wave * 3] = cc; scratch[wave * 3 + 1] = x1; scratch[wave * 3 + 2] = x2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long r0 = 0, r1 = 0, r2 = 0;
            for (int v = 0; v < static_cast<int>((blockDim.x + 63) >> 6); ++v) { r0 += scratch[v * 3]; r1 += scratch[v * 3 + 1]; r2 += scratch[v * 3 + 2]; }
            if (r0 | r1 | r2) { atomicAdd(a.result, r0); atomicAdd(a.result + 1, r1); atomicAdd(a.result + 2, r2); }
        }
    }
}

// pass 3: the segments with their true entries.  KMODE 0: totals, 1: per-segment counts, 2: write
template <class T, bool LEFTMOST, int KMODE>
__device__ __forceinline__ void chain_emit_body(const T &t, const ScanArgs &a, const ChainArgs &c, const uint32_t *outputs,
                                                unsigned long long *next_begin, unsigned long long *scratch) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    ChainWalker<T, LEFTMOST> w{t, a.total_len, c.cap};
    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;
    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        const uint64_t entry = seg == 0 ? lo : c.x_prev[seg - 1];
        unsigned long long cnt = 0;
        uint32_t s1 = 0, s2 = 0;
        daac_match *o = nullptr;
        if (KMODE == 2) o = a.out + a.seg_counts[seg];
        auto emit = [&](uint32_t opos, uint64_t end) {
            const uint32_t *r = outputs + 3u * (opos - 1u);
            const uint32_t value = r[0], length = r[1];
            if (KMODE == 2) {
                daac_match m;
                m.start = end - length; m.end = end; m.value = value; m._pad = 0;
                *o++ = m;
            } else {
                uint64_t z = (static_cast<uint64_t>(value) << 32) | length;  // h of the checksum definition
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
                const uint32_t h = static_cast<uint32_t>(z ^ (z >> 31));
                cnt += 1; s1 += h; s2 += h * static_cast<uint32_t>(end);
            }
        };
        const uint64_t exit = w.run(entry, hi, emit);
        if (hi == a.len && next_begin) *next_begin = exit > a.len ? exit : a.len;  // where a following window starts
        if (KMODE == 0) { tot_cnt += cnt; tot_s1 += s1; tot_s2 += s2; }
        else if (KMODE == 1) a.seg_counts[seg] = cnt;
    }
    if (w.overflow) c.flags[1] = 1u;
    if (KMODE == 0) {
        unsigned long long cc = tot_cnt, x1 = tot_s1, x2 = tot_s2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { cc += __shfl_down(cc, off, 64); x1 += __shfl_down(x1, off, 64); x2 += __shfl_down(x2, off, 64); }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();  // (`scratch` may lie over tables the walkers of other waves were still using)
        if (lane == 0) { scratch[wave * 3] = cc; scratch[wave * 3 + 1] = x1; scratch[wave * 3 + 2] = x2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long r0 = 0, r1 = 0, r2 = 0;
            for (int v = 0; v < static_cast<int>((blockDim.x + 63) >> 6); ++v) { r0 += scratch[v * 3]; r1 += scratch[v * 3 + 1]; r2 += scratch[v * 3 + 2]; }
            if (r0 | r1 | r2) { atomicAdd(a.result, r0); atomicAdd(a.result + 1, r1); atomicAdd(a.result + 2, r2); }
        }
    }
}

}  // namespace daac
