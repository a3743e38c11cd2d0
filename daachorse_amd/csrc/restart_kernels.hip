// Restart scanners for gfx950: FindIterator (reference src/bytewise/iter.rs:58-113) and
// LeftmostFindIterator (iter.rs:272-340) on the device.
//
// Both iterators restart from ROOT after every match, so the scan is a chain through the match
// positions and cannot be cut at arbitrary offsets.  It CAN be cut at a *sync point*: a position p
// where the classic Aho-Corasick state of the text before p is ROOT, i.e. no pattern occurrence
// starts before p and ends after it.  Whatever the iterator did before p, at p it is at ROOT with
// nothing pending, and nothing it reports later starts before p — so the text from one sync point
// to the next can be scanned as if it were a haystack of its own.
//
// One lane per segment [lo, hi):
//   p = first sync point >= lo   (classic automaton warmed up over Lmax bytes — enough to land on the state a
//                                 lane that followed the text from further back is in, whatever its depth — then
//                                 advanced until it sits at ROOT); the lane owns nothing if p >= hi
//   q = first sync point >= hi   (same procedure) — the next owner starts exactly there
//   then the reference's loop, literally, over [p, q), positions offset by p.
// Regions [p, q) partition the haystack in order, so matches are placed by the same
// count -> exclusive scan -> write passes as the overlapping scan.  Text without sync points
// degrades gracefully: fewer, longer regions (in the limit one lane scans everything; still exact).
//
// Tables: the reference's double array, hot {base, opos_ch} / cold fail (DArrayDev); for leftmost
// automata the classic failure links are recomputed on the host (repack.cpp) next to the
// automaton's own (DEAD-terminated) ones.
#include <hip/hip_runtime.h>

#include <cstdint>

#define DAAC_RS_MICRO 1
#include "chain_scan.hpp"
#include "device_tables.hpp"

namespace daac {

__device__ __forceinline__ uint64_t rs_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

struct RsState { uint32_t idx, base, opos_ch, fail, fmap; };  // fail, fmap (the child filter): only kept by the micro-step walker (chain_scan.hpp, run_micro)

struct RestartTables {
    using State = RsState;
    static constexpr bool kMicro = DAAC_RS_MICRO != 0;  // chain_scan.hpp: the walker takes the transition one memory round trip at a time
    const DArrayDev &d;
    const uint4 *l_root;  // in LDS: 256 x {child, child.base, child.opos_ch, child.fail} (restart_scan_kernel), or DArrayDev::root_chain (the walkers)
    const uint8_t *__restrict__ hay = nullptr;
    using Stream = HayStream;

    // the automaton as chain_scan.hpp wants it
    __device__ __forceinline__ RsState root() const { return RsState{0, 0, 0}; }
    __device__ __forceinline__ uint32_t symbol_at(HayWindow &win, uint64_t pos, uint32_t &clen) const { clen = 1; return win.byte_at(hay + pos); }
    __device__ __forceinline__ uint32_t opos(const RsState &st) const { return st.opos_ch >> 8; }
    __device__ __forceinline__ bool is_root(const RsState &st) const { return st.idx == 0; }
    __device__ __forceinline__ uint64_t boundary_at_or_after(uint64_t x) const { return x; }

    // ---- the micro-step walker's view (chain_scan.hpp, run_micro) ----
    __device__ __forceinline__ uint32_t symbol_code(Stream &win, uint32_t pos, uint32_t, uint32_t &clen) const {
        clen = 1;
        return win.byte_at(pos);
    }
    // One memory round trip of the transition on byte c (bytewise.rs:1063-1088 / 1094-1128 taken apart) over the 16-byte
    // records {base, opos_ch, fail, child filter}: phase 0 probes the child slot, phase 1 fetches the record a failure link
    // leads to.  ROOT's row is in LDS: a lane at ROOT, or one whose failed probe leaves it with a link to ROOT, is through
    // without asking memory — and so is a probe the state's child filter rules out (no child on any byte with these low five
    // bits): a failed probe lands on an arbitrary element of the array, i.e. on a cache line nobody else wants, and on text
    // that leaves the dictionary's words after three or four bytes those were most of the walkers' misses.  Every lane loads,
    // every turn (an idle lane asks for slot 0); the outcome is a handful of selects.
    template <bool LM>
    __device__ __forceinline__ bool micro(RsState &st, uint32_t c, uint32_t &phase, bool act) const {
        const bool at_root = st.idx == 0;
        const bool child_possible = st.base != 0 && ((st.fmap >> (c & 31u)) & 1u) != 0;
        // what this turn asks memory: the child slot (a probe), or — after a failed probe (phase 1), or at once when the filter
        // rules the child out — the record the failure link leads to, unless that link ends the walk (DEAD) or leads to ROOT
        const bool probe = act && phase == 0 && !at_root && child_possible;
        const bool no_child = act && !at_root && !probe;
        const bool stop = LM && st.fail == 1u;          // the link is DEAD: the walk ends
        const bool follow = no_child && !stop && st.fail != 0;
        const uint32_t slot = probe ? (st.base ^ c) : follow ? st.fail : 0u;
        const uint4 rr = l_root[c];
        typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
        U32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(d.rec + slot) : "memory");
        const bool hit = probe && (r.y & 0xffu) == c;
        const bool miss = probe && !hit;
        const bool dead = (no_child || miss) && stop;
        const bool rootward = act && !dead && (at_root || ((no_child || miss) && st.fail == 0));
        const bool take = hit || follow;
        phase = (miss && !dead && !rootward) ? 1u : 0u;   // a failed probe whose link leads on: that record next turn
        st.idx = take ? slot : rootward ? rr.x : dead ? 0u : st.idx;
        st.base = take ? r.x : rootward ? rr.y : dead ? 0u : st.base;
        st.opos_ch = take ? r.y : rootward ? rr.z : dead ? 0u : st.opos_ch;
        st.fail = take ? r.z : rootward ? (rr.z & 0xffu) : dead ? 0u : st.fail;
        st.fmap = take ? r.w : rootward ? rr.w : dead ? 0u : st.fmap;
        return hit || dead || rootward;
    }

    // classic delta (failure links never stop): reference src/bytewise.rs:1063-1088 over fail_plain
    __device__ __forceinline__ void step_plain(RsState &st, uint32_t c) const {
        for (;;) {
            if (st.idx == 0) {
                const uint4 r = l_root[c];
                st = RsState{r.x, r.y, r.z};
                return;
            }
            if (st.base != 0) {
                const uint32_t child = st.base ^ c;
                const uint2 h = d.hot[child];
                if ((h.y & 0xffu) == c) { st = RsState{child, h.x, h.y}; return; }
            }
            const uint32_t f = d.fail_plain[st.idx];
            if (f == 0) { st.idx = 0; continue; }
            const uint2 h = d.hot[f];
            st = RsState{f, h.x, h.y};
        }
    }

    // next_state_id_leftmost_unchecked, reference src/bytewise.rs:1094-1128 (returns ROOT on DEAD)
    __device__ __forceinline__ void step_leftmost(RsState &st, uint32_t c) const {
        for (;;) {
            if (st.idx == 0) {  // at ROOT: the child if there is one, else stay (":1113-1116")
                const uint4 r = l_root[c];
                st = RsState{r.x, r.y, r.z};
                return;
            }
            if (st.base != 0) {
                const uint32_t child = st.base ^ c;
                const uint2 h = d.hot[child];
                if ((h.y & 0xffu) == c) { st = RsState{child, h.x, h.y}; return; }
            }
            const uint32_t f = d.fail[st.idx];
            if (f <= 1u) {  // DEAD (1): stop; ROOT (0): retry from the root row
                if (f == 1u) { st = RsState{0, 0, 0}; return; }
                st.idx = 0;
                continue;
            }
            const uint2 h = d.hot[f];
            st = RsState{f, h.x, h.y};
        }
    }

    // first sync point >= x: warm the classic automaton up over the halo, then run it to ROOT
    __device__ __forceinline__ uint64_t sync_from(const uint8_t *hay, uint64_t x, uint32_t halo, uint64_t floor, uint64_t len) const {
        if (x <= floor) return floor;  // the window start is a sync point by contract
        if (x >= len) return len;
        uint64_t pos = x > halo ? x - halo : 0;
        if (pos < floor) pos = floor;
        RsState st{0, 0, 0};
        while (pos < x) step_plain(st, hay[pos++]);
        while (st.idx != 0 && pos < len) step_plain(st, hay[pos++]);
        return st.idx == 0 ? pos : len;
    }
};

// KMODE 0: totals {count, S1, S2}; 1: per-segment counts; 2: write matches at out + seg_counts[seg]
template <bool LEFTMOST, int KMODE>
__global__ __launch_bounds__(256) void restart_scan_kernel(const DArrayDev dev, const ScanArgs a, unsigned long long *next_begin) {
    __shared__ uint4 l_root[256];
    __shared__ unsigned long long scratch[3 * 4];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) l_root[i] = dev.root[i];
    __syncthreads();
    const RestartTables T{dev, l_root, a.hay};
    const uint8_t *__restrict__ hay = a.hay;
    const uint64_t len = a.total_len;  // real end of the haystack; a.len is the nominal end of this window
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;

    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        const uint64_t p = T.sync_from(hay, lo, a.halo, a.begin, len);
        const bool last = hi == a.len;
        uint64_t q = 0;
        if (p < hi || last) q = T.sync_from(hay, hi, a.halo, a.begin, len);
        if (last && next_begin) *next_begin = q;  // where the next window (if any) has to start

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
                const uint32_t h = static_cast<uint32_t>(rs_mix64((static_cast<uint64_t>(value) << 32) | length));
                cnt += 1; s1 += h; s2 += h * static_cast<uint32_t>(end);
            }
        };

        if (!LEFTMOST && dev.root_flag) {
            // "" is a pattern: FindIterator degenerates to (p, p, first "" value) for every p
            // (iter.rs:60-85), whatever the text; positions (lo, hi] belong to this segment, 0 to the first
            const uint32_t op = dev.hot[0].y >> 8;
            if (lo == 0) emit(op, 0);
            for (uint64_t e = lo + 1; e <= hi; ++e) emit(op, e);
        } else if (p < hi || (LEFTMOST && last && p >= len && p == a.begin)) {  // (an empty range still ends the haystack)
            if (!LEFTMOST) {
                // FindIterator::next (iter.rs:87-112): restart at ROOT after every match, report the list head
                RsState st{0, 0, 0};
                for (uint64_t pos = p; pos < q; ++pos) {
                    T.step_plain(st, hay[pos]);
                    if ((st.opos_ch >> 8) != 0) {
                        emit(st.opos_ch >> 8, pos + 1);
                        st = RsState{0, 0, 0};
                    }
                }
            } else {
                // LeftmostFindIterator::next (iter.rs:272-340), call by call, over the bytes of [p, q).
                // `init` is ROOT's output_pos: non-zero when "" is a pattern, which then matches wherever no
                // longer match starts (iter.rs:254-261); every walk is then anchored at `pos` (all failure links
                // are DEAD) and has died by the sync point q.
                uint64_t pos = p;
                uint32_t init = dev.hot[0].y >> 8;
                bool skip_empty = false;
                const bool real_end = q >= len;
                for (;;) {                           // one pass = one call of next()
                    if (!real_end && pos >= q) break;  // the owner of the next region continues from q
                    RsState st{0, 0, 0};
                    uint32_t best = init;            // last_output_pos
                    const uint32_t init_at_entry = init;
                    // One emit site per path: hipcc 7.2 -O3 loses the advance of the output cursor when emit()
                    // is inlined behind the nested break/continue of the literal transcription.
                    uint32_t ret_op = 0;             // what this call returns, if the walk dies on a byte
                    uint64_t ret_end = 0;
                    bool again;
                    do {                             // the reference's loop 'a
                        again = false;
                        for (uint64_t i = pos; i < q; ++i) {
                            T.step_leftmost(st, hay[i]);
                            if (st.idx == 0) {
                                if (best != 0) {
                                    ret_end = pos;
                                    if (best != init) {
                                        skip_empty = true;
                                        ret_op = best;
                                    } else {
                                        pos += 1;
                                        if (skip_empty) { skip_empty = false; again = true; }
                                        else ret_op = best;
                                    }
                                    break;
                                }
                            } else if ((st.opos_ch >> 8) != 0) {
                                best = st.opos_ch >> 8;
                                pos = i + 1;
                            }
                        }
                    } while (again);
                    if (ret_op != 0) { emit(ret_op, ret_end); continue; }
                    // the bytes ran out (iter.rs:320-339)
                    if (!real_end) {                 // at a sync point only a match already seen can be pending
                        if (best != 0 && best != init_at_entry) { emit(best, pos); continue; }
                        break;
                    }
                    if (pos >= len) init = 0;
                    if (best == 0) break;            // None
                    if (best == init_at_entry && pos < len) {
                        // "" is a pattern and the haystack ends inside a longer one: the reference yields the same
                        // empty match forever from here (SURVEY 8a note D).  Reported, not imitated.
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

// ---- speculate / reconcile / emit (chain_scan.hpp) over the bytewise double array -------------------------
template <bool LEFTMOST, int PASS, int KMODE>
__global__ __launch_bounds__(256) void chain_kernel(const DArrayDev dev, const ScanArgs a, const ChainArgs c, unsigned long long *next_begin) {
    __shared__ uint4 l_root[256];
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(l_root);  // (used behind a barrier, when the walkers are through)
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) l_root[i] = dev.root_chain[i];
    __syncthreads();
    const RestartTables T{dev, l_root, a.hay};
    if (PASS == 0) chain_spec_body<RestartTables, LEFTMOST>(T, a, c, dev.ohash);
    else if (PASS == 1) chain_fix_body<RestartTables, LEFTMOST>(T, a, c, dev.ohash);
    else if (PASS == 3) chain_sum_body<KMODE>(a, c, next_begin, scratch);
    else chain_emit_body<RestartTables, LEFTMOST, KMODE>(T, a, c, dev.outputs, next_begin, scratch);
}

// find_overlapping_iter().count() (+ checksum) over the double array with the micro-step walker (chain_scan.hpp,
// overlap_count_body): the engine for automata the GRAM tables do not fit (more than 62 byte classes)
template <bool HEADS>
__global__ __launch_bounds__(256) void overlap_count_kernel(const DArrayDev dev, const ScanArgs a) {
    __shared__ uint4 l_root[256];
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(l_root);  // (used behind a barrier, when the walkers are through)
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) l_root[i] = dev.root_chain[i];
    __syncthreads();
    const RestartTables T{dev, l_root, a.hay};
    overlap_count_body<RestartTables, HEADS>(T, a, dev.osum, dev.ohash, scratch);
}

hipError_t launch_overlap_count(const DArrayDev &dev, const ScanArgs &a, bool heads, uint32_t blocks, hipStream_t stream) {
    if (heads) hipLaunchKernelGGL(overlap_count_kernel<true>, dim3(blocks), dim3(256), 0, stream, dev, a);
    else hipLaunchKernelGGL(overlap_count_kernel<false>, dim3(blocks), dim3(256), 0, stream, dev, a);
    return hipGetLastError();
}

hipError_t launch_chain(const DArrayDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, bool leftmost,
                        unsigned long long *next_begin, uint32_t blocks, hipStream_t stream) {
    const dim3 g(blocks), b(256);
#define DAAC_CH(L, P, M) hipLaunchKernelGGL((chain_kernel<L, P, M>), g, b, 0, stream, dev, a, c, next_begin)
    if (pass == 0) { if (leftmost) DAAC_CH(true, 0, 0); else DAAC_CH(false, 0, 0); }
    else if (pass == 1) { if (leftmost) DAAC_CH(true, 1, 0); else DAAC_CH(false, 1, 0); }
    else if (pass == 3) { if (kmode == 0) DAAC_CH(false, 3, 0); else DAAC_CH(false, 3, 1); }
    else if (leftmost) { if (kmode == 0) DAAC_CH(true, 2, 0); else if (kmode == 1) DAAC_CH(true, 2, 1); else DAAC_CH(true, 2, 2); }
    else { if (kmode == 0) DAAC_CH(false, 2, 0); else if (kmode == 1) DAAC_CH(false, 2, 1); else DAAC_CH(false, 2, 2); }
#undef DAAC_CH
    return hipGetLastError();
}

hipError_t launch_restart_scan(const DArrayDev &dev, const ScanArgs &a, int kmode, bool leftmost, unsigned long long *next_begin,
                               uint32_t blocks, uint32_t threads, hipStream_t stream) {
    const dim3 g(blocks), b(threads > 256 ? 256 : threads);
#define DAAC_RS(L, M) hipLaunchKernelGGL((restart_scan_kernel<L, M>), g, b, 0, stream, dev, a, next_begin)
    if (leftmost) { if (kmode == 0) DAAC_RS(true, 0); else if (kmode == 1) DAAC_RS(true, 1); else DAAC_RS(true, 2); }
    else { if (kmode == 0) DAAC_RS(false, 0); else if (kmode == 1) DAAC_RS(false, 1); else DAAC_RS(false, 2); }
#undef DAAC_RS
    return hipGetLastError();
}

}  // namespace daac
