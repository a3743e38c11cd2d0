// JUMP engine (gfx950): find_iter of a Standard bytewise automaton as three position-parallel passes — see jump.hpp for the
// reduction (reference src/bytewise/iter.rs:58-113: restart at ROOT after every match, report the head of the first output list).
//
//   jump_len_kernel    L(s) for every start s: one K-gram lookup in LDS per position (MS word: the shortest pattern of up to K
//                      bytes that is a prefix of the K-gram, else the continuation bits of the (K+1)-grams); the few starts
//                      that have no short pattern but a trie path below them are compacted by wave ballot into an LDS queue
//                      and walked 64 at a time, in lockstep, down to the first state that ends a pattern.  Output: LSH[i] =
//                      L(i - 2) (u8, indexed by the position of the K-gram's LAST byte) and, for starts settled by a walk,
//                      HDEEP[i] = h32 of that pattern.
//   jump_nd_kernel     N(e) = min over s >= e of (s - e + L(s)), saturated at 255, and D(e) = the smallest s attaining it, minus e:
//                      a suffix-minimum, right to left: 32 positions per lane, the lanes of a wave joined by a scan of
//                      (value, offset) pairs, 256 positions of look-ahead per wave.  Output ND[i] = N | D << 8 (same shift).
//   jump_chain_kernel  the chain e -> e + N(e) through the matches, by segment: speculate / reconcile / sum exactly as
//                      chain_scan.hpp does it for the automaton walkers, but a link is ONE 2-byte load.  The match at a link is
//                      {start = e + D, end = e + N}; its h32 comes from H1 / H2 / H3 by the classes of its bytes (or HDEEP), two
//                      loads that are asked for at one link and the next and folded in one link later: the chain never waits
//                      for them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "chain_scan.hpp"
#include "device_tables.hpp"

namespace daac {

namespace {

typedef uint32_t j_u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const uint32_t ldsj_cu32;
typedef __attribute__((address_space(3))) uint32_t ldsj_u32;
typedef __attribute__((address_space(3))) const uint8_t ldsj_cu8;
constexpr uint32_t kJRing = 128;       // entries of a wave's queue of deep starts
constexpr uint32_t kJOffMS = 256;      // LDS offset of MS (= kJumpOffMS of jump.hpp)
constexpr uint32_t kNdOut = 1792;      // positions a wave of jump_nd_kernel settles (of the 2048 it reads: 256 are look-ahead)

__device__ __forceinline__ void j_copy(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

}  // namespace

// ---- pass A: L(s) ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void jump_len_kernel(const JumpDev g, const JumpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    j_copy(smem, g.cls, 256);
    j_copy(smem + kJOffMS, g.ms, g.ms_bytes);
    j_copy(smem + kJOffMS + g.ms_bytes, g.sdir, g.sdir_bytes);
    __syncthreads();
    auto cls_of = [&](uint32_t byte) -> uint32_t { return *reinterpret_cast<ldsj_cu8 *>(static_cast<uintptr_t>(byte)); };
    auto lds_u32 = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<ldsj_cu32 *>(static_cast<uintptr_t>(addr)); };
    const uint32_t lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const uint32_t offS = kJOffMS + g.ms_bytes;
    const uint32_t ringb = offS + g.sdir_bytes + wave_in_wg * kJRing * 4u;
    const uint32_t C = g.C;
    const uint8_t *__restrict__ hay = a.hay_al;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave_in_wg;
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    const uint32_t ub = g.unused_byte;

    auto byte_at = [&](uint64_t v) -> uint32_t { return (v >= a.lead && v < a.vlen) ? hay[v] : ub; };
    // eight text bytes from v on (unused bytes outside the haystack)
    auto read8 = [&](uint64_t v) -> unsigned long long {
        unsigned long long x;
        if (v >= a.lead && v + 8 <= a.vlen) {
            __builtin_memcpy(&x, hay + v, 8);
        } else {
            x = 0;
            for (int b = 7; b >= 0; --b) x = (x << 8) | byte_at(v + b);
        }
        return x;
    };

    uint32_t q_head = 0, q_tail = 0;  // wave-uniform, free running
    // Walks up to 64 queued starts, in lockstep, down to the first state that ends a pattern.  An entry names the position i of
    // the K-gram's last byte: (turn of this wave) << 10 | offset in the turn's KiB.
    auto process_batch = [&](uint32_t n) {
        bool live = lane < n;
        uint64_t i = 0;
        uint4 rec = uint4{0u, 0u, 0u, 0u};
        unsigned long long text = 0;
        uint32_t have = 0;
        uint64_t pos = 0;
        if (live) {
            const uint32_t e = lds_u32(ringb + (((q_head + lane) & (kJRing - 1u)) << 2));
            i = ((static_cast<uint64_t>(e >> 10) * nwaves + wave) << 10) + (e & 1023u);
            text = read8(i - 2);  // (i >= 2: position i - 2 is a start inside the haystack, or the entry would not exist)
            const uint32_t c0 = cls_of(static_cast<uint32_t>(text) & 0xffu), c1 = cls_of(static_cast<uint32_t>(text >> 8) & 0xffu),
                           c2 = cls_of(static_cast<uint32_t>(text >> 16) & 0xffu), d = cls_of(static_cast<uint32_t>(text >> 24) & 0xffu);
            const uint32_t idx = (c0 * C + c1) * C + c2;
            const uint32_t am = kJOffMS + (idx << 2), grp = kJOffMS + ((idx & ~3u) << 2);
            const uint32_t own = lds_u32(am), qx = lds_u32(grp), qy = lds_u32(grp + 4u), qz = lds_u32(grp + 8u);
            const uint32_t sub = idx & 3u;
            uint32_t rank = lds_u32(offS + ((idx >> 2) << 2)) + __popc(own & ((1u << d) - 2u));
            rank += sub > 0 ? __popc(qx & 0x3ffffffeu) : 0u;
            rank += sub > 1 ? __popc(qy & 0x3ffffffeu) : 0u;
            rank += sub > 2 ? __popc(qz & 0x3ffffffeu) : 0u;
            rec = g.jhit[rank];
            text >>= 32;   // bytes i + 2 .. i + 5
            have = 4;
            pos = i + 2;
        }
        for (;;) {
            if (live) {
                if (rec.x & 1u) {  // this state ends a pattern: the shortest one that starts at i - 2
                    a.lsh[i] = static_cast<uint8_t>(rec.w);
                    a.hdeep[i] = rec.z;
                    live = false;
                } else {
                    if (have == 0) { text = read8(pos); have = 8; }
                    const uint32_t k = cls_of(static_cast<uint32_t>(text) & 0xffu);
                    if (pos >= a.vlen || !((rec.x >> k) & 1u) || k == 0) {
                        live = false;  // the path ends: no pattern starts at i - 2
                    } else {
                        rec = g.jrec[rec.y + __popc(rec.x & ((1u << k) - 2u))];
                        text >>= 8;
                        --have;
                        ++pos;
                    }
                }
            }
            if (__ballot(live) == 0) break;
        }
        q_head += n;
    };

    uint32_t turn = 0;
    for (uint64_t step = wave; step < a.nsteps; step += nwaves, ++turn) {
        const uint64_t v = (step << 10) + lane * 16u;
        // the lane's 16 bytes, the two before them and the one behind
        uint32_t w[4] = {ub * 0x01010101u, ub * 0x01010101u, ub * 0x01010101u, ub * 0x01010101u};
        if (v < a.vlen) {
            const j_u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const j_u32x4_t *>(hay + v));
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            if (v < a.lead || v + 16 > a.vlen) {
                for (int b = 0; b < 16; ++b) {
                    const uint64_t p = v + b;
                    if (p < a.lead || p >= a.vlen) w[b >> 2] = (w[b >> 2] & ~(0xffu << (8 * (b & 3)))) | (ub << (8 * (b & 3)));
                }
            }
        }
        uint32_t kx[19];
        kx[0] = cls_of(v >= 2 ? byte_at(v - 2) : ub);
        kx[1] = cls_of(v >= 1 ? byte_at(v - 1) : ub);
#pragma unroll
        for (int b = 0; b < 16; ++b) kx[2 + b] = cls_of((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
        kx[18] = cls_of(byte_at(v + 16));
        uint32_t H = 0;
        uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t idx = (kx[j] * C + kx[j + 1]) * C + kx[j + 2];
            const uint32_t m = lds_u32(kJOffMS + (idx << 2));
            out[j >> 2] |= (m >> 30) << (8 * (j & 3));
            H |= ((m >> kx[j + 3]) & 1u) << j;   // (continuation bits are only set where no short pattern starts; bit 0 never)
        }
        // (positions outside the haystack read as unused bytes: L = 0 there by construction)
        *reinterpret_cast<j_u32x4_t *>(a.lsh + v) = j_u32x4_t{out[0], out[1], out[2], out[3]};
        // ---- queue the deep starts, one per lane and turn of the loop ----
        for (;;) {
            const bool has = H != 0;
            const unsigned long long mm = __ballot(has);
            if (mm == 0) break;
            if (has) {
                const uint32_t b = static_cast<uint32_t>(__builtin_ctz(H));
                H &= H - 1u;
                const uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mm), 0u));
                *reinterpret_cast<ldsj_u32 *>(static_cast<uintptr_t>(ringb + ((at & (kJRing - 1u)) << 2))) = (turn << 10) | (lane * 16u + b);
            }
            q_tail += static_cast<uint32_t>(__popcll(mm));
            if (q_tail - q_head >= 64u) process_batch(64u);
        }
    }
    if (q_tail != q_head) process_batch(q_tail - q_head);
}

// ---- pass B: the suffix minimum ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jump_nd_kernel(const JumpArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    for (uint64_t chunk = wave; chunk < a.nd_chunks; chunk += nwaves) {
        const uint64_t base = chunk * kNdOut + lane * 32u;
        const j_u32x4_t q0 = *reinterpret_cast<const j_u32x4_t *>(a.lsh + base), q1 = *reinterpret_cast<const j_u32x4_t *>(a.lsh + base + 16);
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        // the lane's first position with nothing known to its right: (A, DA)
        uint32_t n = 255, d = 0;
#pragma unroll
        for (int q = 31; q >= 0; --q) {
            const uint32_t len = (w[q >> 2] >> (8 * (q & 3))) & 0xffu;
            const uint32_t other = n < 255u ? n + 1u : 255u;
            const bool mine = len != 0 && len <= other;
            d = mine ? 0u : d + 1u;
            n = mine ? len : other;
        }
        // what stands at the first position of the lane to the right: the nearest of the next eight lanes' values
        uint32_t X = 255, DX = 0;
#pragma unroll
        for (int j = 8; j >= 1; --j) {
            const uint32_t an = __shfl_down(n, j, 64), ad = __shfl_down(d, j, 64);
            const uint32_t cand = (lane + j < 64u && an < 255u) ? an + 32u * (j - 1) : 255u;
            if (cand < 255u && cand <= X) { X = cand; DX = ad + 32u * (j - 1); }
        }
        // second pass with the right-hand value in place
        n = X; d = DX;
        uint32_t o[16];
#pragma unroll
        for (int q = 31; q >= 0; --q) {
            const uint32_t len = (w[q >> 2] >> (8 * (q & 3))) & 0xffu;
            const uint32_t other = n < 255u ? n + 1u : 255u;
            const bool mine = len != 0 && len <= other;
            d = mine ? 0u : d + 1u;
            n = mine ? len : other;
            const uint32_t nd = n | ((d & 0xffu) << 8);
            if (q & 1) o[q >> 1] = nd << 16; else o[q >> 1] |= nd;
        }
        if (lane < kNdOut / 32u) {
            j_u32x4_t *dst = reinterpret_cast<j_u32x4_t *>(a.nd + base);
            dst[0] = j_u32x4_t{o[0], o[1], o[2], o[3]};
            dst[1] = j_u32x4_t{o[4], o[5], o[6], o[7]};
            dst[2] = j_u32x4_t{o[8], o[9], o[10], o[11]};
            dst[3] = j_u32x4_t{o[12], o[13], o[14], o[15]};
        }
    }
}

// ---- pass C: the chain ------------------------------------------------------------------------------------------------------
namespace {

struct JumpWalk {
    const JumpDev &g;
    const JumpArgs &a;
    const ScanArgs &s;
    const uint16_t *ndp;      // ndp[r - begin] = ND of chain position r
    const uint32_t *hdp;      // hdp[r - begin] = HDEEP of start r
    ldsj_cu8 *l_cls;          // byte classes in LDS
    uint64_t len;             // end of the haystack

    __device__ __forceinline__ uint32_t byte4(uint64_t r) const {  // text bytes r .. r + 3 (zero behind the haystack's last granule)
        const uintptr_t p = reinterpret_cast<uintptr_t>(s.hay) + r, al = p & ~static_cast<uintptr_t>(3);
        const uintptr_t lim = (reinterpret_cast<uintptr_t>(s.hay) + len + 15u) & ~static_cast<uintptr_t>(15);
        const uint32_t lo = *reinterpret_cast<const uint32_t *>(al);
        const uint32_t hi = al + 4 < lim ? *reinterpret_cast<const uint32_t *>(al + 4) : 0u;
        return __builtin_amdgcn_alignbyte(hi, lo, static_cast<uint32_t>(p & 3u));
    }
};

// {count, S1, S2} of the matches of a chain; the h32 of a match takes two dependent loads (its text, then the table entry of its
// classes): asked for at one report and the next, folded in the one after — the chain's own load is the only thing waited for
struct JumpTally {
    const JumpWalk &w;
    unsigned long long cnt = 0;
    uint32_t s1 = 0, s2 = 0;
    uint32_t t_wait = 0, l_wait = 0, end_t = 0;   // stage 1: the match's first bytes (asked for), its length, its end
    uint64_t start_t = 0;
    uint32_t h_wait = 0, end_h = 0;               // stage 2: its h32 (asked for), its end
    __device__ __forceinline__ uint32_t table_load() const {
        if (l_wait == 0) return 0u;
        const uint32_t c0 = w.l_cls[t_wait & 0xffu], c1 = w.l_cls[(t_wait >> 8) & 0xffu], c2 = w.l_cls[(t_wait >> 16) & 0xffu];
        const uint32_t *p = l_wait == 1 ? w.g.h1 + c0 : l_wait == 2 ? w.g.h2 + (c0 * w.g.C + c1) : l_wait == 3 ? w.g.h3 + ((c0 * w.g.C + c1) * w.g.C + c2)
                                                                                                              : w.hdp + (start_t - w.s.begin);
        return *p;
    }
    __device__ __forceinline__ void operator()(uint32_t length, uint64_t end) {
        s1 += h_wait; s2 += h_wait * end_h;
        h_wait = table_load();
        end_h = end_t;
        start_t = end - length;
        t_wait = length <= 3 ? w.byte4(start_t) : 0u;
        l_wait = length;
        end_t = static_cast<uint32_t>(end);
        cnt += 1;
    }
    __device__ __forceinline__ uint4 packed() const {
        const uint32_t h = table_load();
        return uint4{static_cast<uint32_t>(cnt), static_cast<uint32_t>(cnt >> 32), s1 + h_wait + h, s2 + h_wait * end_h + h * end_t};
    }
};

// one link of the chain from position r: the next chain position (> r), the match reported on the way (if any)
template <class Emit>
__device__ __forceinline__ uint64_t jump_link(const JumpWalk &w, uint64_t r, Emit &&emit) {
    if (r >= w.len) return w.len;
    const uint32_t nd = w.ndp[r - w.s.begin];
    const uint32_t n = nd & 0xffu;
    if (n == 255u) {  // nothing starts in the next 255 - Lmax bytes (jump.hpp)
        const uint64_t r2 = r + (255u - w.g.max_len);
        return r2 < w.len ? r2 : w.len;
    }
    emit(n - (nd >> 8), r + n);
    return r + n;
}
template <class Emit>
__device__ __forceinline__ uint64_t jump_run(const JumpWalk &w, uint64_t entry, uint64_t hi, Emit &&emit) {
    uint64_t r = entry;
    while (r < hi) r = jump_link(w, r, emit);
    return r;
}

}  // namespace

// PASS 0: speculative exits, 1: a reconciliation round, 3: the sums (chain_scan.hpp)
template <int PASS>
__global__ __launch_bounds__(256) void jump_chain_kernel(const JumpDev g, const JumpArgs ja, const ScanArgs a, const ChainArgs c) {
    __shared__ __attribute__((aligned(16))) uint8_t l_cls[256];
    __shared__ unsigned long long scratch[3 * 4];
    if (PASS == 3) {
        chain_sum_body<0>(a, c, nullptr, scratch);
        return;
    }
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) l_cls[i] = g.cls[i];
    __syncthreads();
    const JumpWalk w{g, ja, a, ja.nd + ja.lead + 2, ja.hdeep + ja.lead + 2, (ldsj_cu8 *)l_cls, a.total_len};
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    if (PASS == 0) {
        for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
            const uint64_t lo = a.begin + seg * a.seg_bytes;
            const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
            JumpTally tally{w};
            c.x_out[seg] = jump_run(w, lo, hi, tally);
            c.tally_spec[seg] = tally.packed();
        }
    } else {
        bool changed = false;
        for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
            const uint64_t lo = a.begin + seg * a.seg_bytes;
            const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
            uint64_t exit = c.x_spec[seg];
            uint4 delta{0u, 0u, 0u, 0u};        // true chain minus speculative chain, in reported matches
            if (seg != 0) {
                const uint64_t entry = c.x_prev[seg - 1];
                uint64_t s = lo;
                if (entry >= hi) {
                    exit = entry;               // the chain jumps over this segment: nothing of it is reported
                    delta = tally_sub(delta, c.tally_spec[seg]);
                } else if (entry != s) {
                    uint64_t x = entry;         // the true chain; s follows the speculative one
                    JumpTally of_true{w}, of_spec{w};
                    bool merged = false;
                    while (x < hi) {
                        while (s < x && s < hi) s = jump_link(w, s, of_spec);
                        if (s == x) { merged = true; break; }
                        x = jump_link(w, x, of_true);
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
    }
}

// ---- launches ---------------------------------------------------------------------------------------------------------------
uint32_t jump_len_lds_bytes(const JumpDev &g) { return kJOffMS + g.ms_bytes + g.sdir_bytes + 16u * kJRing * 4u; }

hipError_t launch_jump_tables(const JumpDev &g, const JumpArgs &a, uint32_t num_cu, hipStream_t stream) {
    const uint32_t lds = jump_len_lds_bytes(g);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(jump_len_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    const uint32_t blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(num_cu, (a.nsteps + 15) / 16)));
    hipLaunchKernelGGL(jump_len_kernel, dim3(blocks), dim3(1024), lds, stream, g, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const uint32_t nd_blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(num_cu) * 8, (a.nd_chunks + 3) / 4)));
    hipLaunchKernelGGL(jump_nd_kernel, dim3(nd_blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_jump_chain(const JumpDev &g, const JumpArgs &ja, const ScanArgs &a, const ChainArgs &c, int pass, uint32_t blocks, hipStream_t stream) {
    const dim3 gr(blocks), b(256);
    if (pass == 0) hipLaunchKernelGGL(jump_chain_kernel<0>, gr, b, 0, stream, g, ja, a, c);
    else if (pass == 1) hipLaunchKernelGGL(jump_chain_kernel<1>, gr, b, 0, stream, g, ja, a, c);
    else hipLaunchKernelGGL(jump_chain_kernel<3>, gr, b, 0, stream, g, ja, a, c);
    return hipGetLastError();
}

}  // namespace daac
