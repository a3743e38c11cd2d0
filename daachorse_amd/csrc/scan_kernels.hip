// HIP kernels of the daachorse scan path for gfx950 (MI355X / CDNA4).
//
// One wavefront lane scans one haystack segment.  A segment [lo, hi) is entered (Lmax - 1) bytes
// early from ROOT: the Aho-Corasick state after position p is a suffix of the text of length
// <= Lmax, so the halo reproduces the sequential state at `lo` exactly (SURVEY.md §8a note A) and
// the lane reports precisely the matches with end in (lo, hi], in the reference's order
// (FindOverlappingIterator::next, reference src/bytewise/iter.rs:133-176).  Concatenating the
// segments in order gives the reference stream; matches are placed by a count -> exclusive scan
// -> write pass, so the output is deterministic and needs no sorting.
//
// The transition function exists twice (see repack.hpp):
//   TierEngine   bitmap-rank trie, dense fail-resolved rows + child bitmaps in LDS
//   DArrayEngine the reference's double array (bytewise.rs:1063-1088), root row in LDS
// Integer work throughout; no MFMA.  The roofline of these kernels is HBM bytes of haystack.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_tables.hpp"

namespace daac {

// ------------------------------------------------------------------------------------------ utils
__device__ __forceinline__ uint64_t mix64_dev(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t match_hash32_dev(uint32_t value, uint32_t length) {
    return static_cast<uint32_t>(mix64_dev((static_cast<uint64_t>(value) << 32) | length));
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Streaming (non-temporal) 16-byte haystack load: the haystack is read once and must not evict
// the automaton tables from L2.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t load_hay16(const uint8_t *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
}

// Cooperative global -> LDS copy of a table (16-byte granules; sizes are padded by the host).
__device__ __forceinline__ void copy_to_lds(void *dst, const void *src, uint32_t bytes) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

// ------------------------------------------------------------------------------------ TierEngine
template <bool ROW32>
struct TierEngine {
    using Dev = TierDev;
    using Row = typename std::conditional<ROW32, uint32_t, uint16_t>::type;
    struct State { uint32_t id; };

    const TierDev &d;
    const Row *l_rows;
    const uint32_t *l_bcmap;
    const uint32_t *l_bfail;
    const uint2 *l_ssum;
    const uint8_t *l_cls;

    __device__ TierEngine(const TierDev &dev, char *smem)
        : d(dev),
          l_rows(reinterpret_cast<const Row *>(smem)),
          l_bcmap(reinterpret_cast<const uint32_t *>(smem + dev.off_bcmap)),
          l_bfail(reinterpret_cast<const uint32_t *>(smem + dev.off_bfail)),
          l_ssum(reinterpret_cast<const uint2 *>(smem + dev.off_ssum)),
          l_cls(reinterpret_cast<const uint8_t *>(smem + dev.off_cls)) {}

    __device__ void load_lds(char *smem) const {
        copy_to_lds(smem, d.rows, d.off_bcmap);
        copy_to_lds(smem + d.off_bcmap, d.bcmap, d.off_bfail - d.off_bcmap);
        copy_to_lds(smem + d.off_bfail, d.bfail, d.off_ssum - d.off_bfail);
        copy_to_lds(smem + d.off_ssum, d.ssum, d.off_cls - d.off_ssum);
        copy_to_lds(smem + d.off_cls, d.cls, 256);
    }

    __device__ __forceinline__ State root() const { return State{0}; }
    __device__ __forceinline__ bool root_flag() const { return d.root_flag != 0; }

    // delta(state, byte) -> new state; returns whether the new state carries an output list.
    __device__ __forceinline__ bool step(State &st, uint32_t c) const {
        const uint32_t k = l_cls[c];
        uint32_t s = st.id;
        for (;;) {
            if (s < d.NA) {  // tier A: dense row, failure links already resolved
                const uint32_t e = l_rows[s * d.C + k];
                constexpr uint32_t kShift = ROW32 ? 31 : 15;
                st.id = e & ((1u << kShift) - 1u);
                return (e >> kShift) != 0;
            }
            uint32_t cmap, omap, first, fail;
            if (s < d.NB) {  // tier B: bitmap and fail in LDS, the rest only on a hit
                cmap = l_bcmap[s - d.NA];
                if (((cmap >> k) & 1u) == 0) { s = l_bfail[s - d.NA]; continue; }
                const uint4 r = d.grec[s];
                omap = r.y; first = r.z;
            } else {         // tier C: one 16-byte record from L2/HBM
                const uint4 r = d.grec[s];
                cmap = r.x; omap = r.y; first = r.z; fail = r.w;
                if (((cmap >> k) & 1u) == 0) { s = fail; continue; }
            }
            st.id = first + __popc(cmap & ((1u << k) - 1u));
            return ((omap >> k) & 1u) != 0;
        }
    }

    // {number of outputs, sum of their h32} of the state's output list
    __device__ __forceinline__ uint2 sum(const State &st) const {
        return st.id < d.NA ? l_ssum[st.id] : d.ssum[st.id];
    }
    __device__ __forceinline__ uint32_t opos(const State &st) const { return d.sopos[st.id]; }
    __device__ __forceinline__ const uint32_t *outputs() const { return d.outputs; }
};

// ---------------------------------------------------------------------------------- DArrayEngine
struct DArrayEngine {
    using Dev = DArrayDev;
    struct State { uint32_t idx, base, opos_ch; };

    const DArrayDev &d;
    const uint4 *l_root;

    __device__ DArrayEngine(const DArrayDev &dev, char *smem) : d(dev), l_root(reinterpret_cast<const uint4 *>(smem)) {}
    __device__ void load_lds(char *smem) const { copy_to_lds(smem, d.root, 256 * 16); }

    __device__ __forceinline__ State root() const {
        const uint2 h = d.hot[0];
        return State{0, h.x, h.y};
    }
    __device__ __forceinline__ bool root_flag() const { return d.root_flag != 0; }

    // next_state_id_unchecked, reference src/bytewise.rs:1063-1088
    __device__ __forceinline__ bool step(State &st, uint32_t c) const {
        for (;;) {
            if (st.idx == 0) {
                const uint4 r = l_root[c];  // {child, child.base, child.opos_ch}
                st = State{r.x, r.y, r.z};
                return (r.z >> 8) != 0;
            }
            if (st.base != 0) {
                const uint32_t child = st.base ^ c;
                const uint2 h = d.hot[child];
                if ((h.y & 0xffu) == c) {
                    st = State{child, h.x, h.y};
                    return (h.y >> 8) != 0;
                }
            }
            const uint32_t f = d.fail[st.idx];
            if (f == 0) { st.idx = 0; continue; }
            const uint2 h = d.hot[f];
            st = State{f, h.x, h.y};
        }
    }

    __device__ __forceinline__ uint2 sum(const State &st) const { return d.osum[(st.opos_ch >> 8) - 1]; }
    __device__ __forceinline__ uint32_t opos(const State &st) const { return st.opos_ch >> 8; }
    __device__ __forceinline__ const uint32_t *outputs() const { return d.outputs; }
};

// ------------------------------------------------------------------------------------ CharEngine
// CharwiseDoubleArrayAhoCorasick (reference src/charwise.rs:1022-1050): the lane is fed bytes and
// assembles UTF-8 scalars itself (charwise/iter.rs:64-98); a transition happens when a scalar is
// complete, so `e` in the scan loop is the byte offset of the END of the character, which is what
// the reference reports.  A segment may begin or end inside a character: continuation bytes seen
// before the first lead byte are skipped, and a character cut by the segment end is finished by
// the next lane (its halo starts earlier).  Unmapped scalars send the automaton to ROOT.
struct CharEngine {
    using Dev = CharDev;
    struct State { uint32_t idx, base, fail, opos, cp, need; };

    const CharDev &d;
    uint4 root_rec;

    __device__ CharEngine(const CharDev &dev, char *) : d(dev), root_rec(dev.states[0]) {}
    __device__ void load_lds(char *) const {}

    __device__ __forceinline__ State root() const { return State{0, root_rec.x, root_rec.z, root_rec.w, 0, 0}; }
    __device__ __forceinline__ bool root_flag() const { return d.root_flag != 0; }

    __device__ __forceinline__ bool step(State &st, uint32_t b) const {
        if (b < 0x80u) { st.cp = b; st.need = 0; }
        else if (b < 0xc0u) {
            if (st.need == 0) return false;  // inside a character that began before the lane's first byte
            st.cp = (st.cp << 6) | (b & 0x3fu);
            if (--st.need != 0) return false;
        } else {
            st.cp = b < 0xe0u ? (b & 0x1fu) : b < 0xf0u ? (b & 0x0fu) : (b & 0x07u);
            st.need = b < 0xe0u ? 1u : b < 0xf0u ? 2u : 3u;
            return false;
        }
        const uint32_t code = st.cp < d.table_len ? d.table[st.cp] : 0xffffffffu;
        if (code == 0xffffffffu) {  // charwise.rs:1031-1035
            st.idx = 0; st.base = root_rec.x; st.fail = root_rec.z; st.opos = root_rec.w;
            return root_rec.w != 0;
        }
        for (;;) {
            if (st.base != 0) {
                const uint32_t child = st.base ^ code;
                const uint4 r = d.states[child];
                if (r.y == st.idx) {
                    st.idx = child; st.base = r.x; st.fail = r.z; st.opos = r.w;
                    return r.w != 0;
                }
            }
            if (st.idx == 0) return root_rec.w != 0;
            const uint32_t f = st.fail;
            const uint4 r = d.states[f];
            st.idx = f; st.base = r.x; st.fail = r.z; st.opos = r.w;
        }
    }

    __device__ __forceinline__ uint2 sum(const State &st) const { return d.osum[st.opos - 1]; }
    __device__ __forceinline__ uint32_t opos(const State &st) const { return st.opos; }
    __device__ __forceinline__ const uint32_t *outputs() const { return d.outputs; }
};

// --------------------------------------------------------------------------------- the scan kernel
// MODE 0: count + checksum into a.result (atomics, one per workgroup)
// MODE 1: per-segment match counts into a.seg_counts
// MODE 2: write matches at a.out + a.seg_counts[seg] (exclusive offsets)
template <class Eng, int MODE, bool HEADS>
__global__ __launch_bounds__(1024) void scan_kernel(const typename Eng::Dev dev, const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Eng eng(dev, smem);
    eng.load_lds(smem);
    __syncthreads();

    const uint8_t *__restrict__ hay = a.hay;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long tot_cnt = 0;
    uint32_t tot_s1 = 0, tot_s2 = 0;

    for (uint64_t seg = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; seg < a.nseg; seg += stride) {
        const uint64_t lo = a.begin + seg * a.seg_bytes;
        const uint64_t hi = (lo + a.seg_bytes < a.len) ? lo + a.seg_bytes : a.len;
        uint64_t p = lo > a.halo ? lo - a.halo : 0;

        unsigned long long cnt = 0;
        uint32_t s1 = 0, s2 = 0, e = 0;  // e = end - lo of the byte just consumed
        daac_match *o = nullptr;
        if (MODE == 2) o = a.out + a.seg_counts[seg];

        typename Eng::State st = eng.root();

        auto emit = [&](const typename Eng::State &s) {
            if (MODE != 2) {
                if (HEADS) {  // FindOverlappingNoSuffixIterator: only the head of the list
                    const uint32_t *r = eng.outputs() + 3u * (eng.opos(s) - 1u);
                    const uint32_t h = match_hash32_dev(r[0], r[1]);
                    cnt += 1; s1 += h; s2 += h * e;
                } else {
                    const uint2 q = eng.sum(s);
                    cnt += q.x; s1 += q.y; s2 += q.y * e;
                }
            } else {
                uint32_t op = eng.opos(s);
                const uint64_t end = lo + e;
                do {
                    const uint32_t *r = eng.outputs() + 3u * (op - 1u);
                    const uint32_t value = r[0], length = r[1];
                    op = HEADS ? 0u : r[2];
                    daac_match m;
                    m.start = end - length; m.end = end; m.value = value; m._pad = 0;
                    *o++ = m;
                } while (op != 0);
            }
        };

        // ROOT's own list is drained once, at end = 0 (iter.rs:134-148 with pos = 0)
        if (lo == 0 && eng.root_flag()) emit(st);

        for (; p < lo; ++p) eng.step(st, hay[p]);  // halo warm-up, nothing reported

        auto on_byte = [&](uint32_t c) {
            ++e;
            if (eng.step(st, c)) emit(st);
        };

        while (p < hi && (reinterpret_cast<uintptr_t>(hay + p) & 15u) != 0) on_byte(hay[p++]);
        const uint64_t nvec = (hi - p) >> 4;
        if (nvec != 0) {
            const uint8_t *vp = hay + p;
            u32x4_t cur = load_hay16(vp);
            for (uint64_t i = 0; i < nvec; ++i) {
                const u32x4_t nxt = load_hay16(vp + 16 * (i + 1 < nvec ? i + 1 : i));
                const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    on_byte(w[j] & 0xffu);
                    on_byte((w[j] >> 8) & 0xffu);
                    on_byte((w[j] >> 16) & 0xffu);
                    on_byte(w[j] >> 24);
                }
                cur = nxt;
            }
            p += nvec << 4;
        }
        while (p < hi) on_byte(hay[p++]);

        if (MODE == 0) {
            tot_cnt += cnt;
            tot_s1 += s1;
            tot_s2 += s2 + s1 * static_cast<uint32_t>(lo);  // sum h * low32(lo + e)
        } else if (MODE == 1) {
            a.seg_counts[seg] = cnt;
        }
    }

    if (MODE == 0) {
        unsigned long long c = wave_sum_u64(tot_cnt);
        unsigned long long x1 = wave_sum_u64(tot_s1);
        unsigned long long x2 = wave_sum_u64(tot_s2);
        __syncthreads();  // the tables in LDS are dead from here on
        unsigned long long *scratch = reinterpret_cast<unsigned long long *>(smem);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { scratch[wave * 3 + 0] = c; scratch[wave * 3 + 1] = x1; scratch[wave * 3 + 2] = x2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            unsigned long long r0 = 0, r1 = 0, r2 = 0;
            for (int w = 0; w < nw; ++w) { r0 += scratch[w * 3]; r1 += scratch[w * 3 + 1]; r2 += scratch[w * 3 + 2]; }
            if (r0 | r1 | r2) {
                atomicAdd(a.result + 0, r0);
                atomicAdd(a.result + 1, r1);
                atomicAdd(a.result + 2, r2);
            }
        }
    }
}

// Exclusive scan of per-segment counts (in place), total into result[0].  One workgroup:
// nseg is a few hundred thousand at most, this is microseconds.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(unsigned long long *v, uint64_t n, unsigned long long *total) {
    __shared__ unsigned long long part[1024];
    const uint64_t per = (n + blockDim.x - 1) / blockDim.x;
    const uint64_t b = per * threadIdx.x;
    const uint64_t e = b + per < n ? b + per : n;
    unsigned long long s = 0;
    for (uint64_t i = b; i < e; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (unsigned i = 0; i < blockDim.x; ++i) { const unsigned long long t = part[i]; part[i] = run; run += t; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint64_t i = b; i < e; ++i) { const unsigned long long t = v[i]; v[i] = run; run += t; }
}

// ------------------------------------------------------------------------------------- launchers
template <class Eng>
static hipError_t launch_mode(const typename Eng::Dev &dev, const ScanArgs &a, int mode, bool heads, dim3 grid, dim3 block,
                              uint32_t lds, hipStream_t stream) {
#define DAAC_LAUNCH(M, H) hipLaunchKernelGGL((scan_kernel<Eng, M, H>), grid, block, lds, stream, dev, a)
    if (mode == 0) { if (heads) DAAC_LAUNCH(0, true); else DAAC_LAUNCH(0, false); }
    else if (mode == 1) { if (heads) DAAC_LAUNCH(1, true); else DAAC_LAUNCH(1, false); }
    else { if (heads) DAAC_LAUNCH(2, true); else DAAC_LAUNCH(2, false); }
#undef DAAC_LAUNCH
    return hipGetLastError();
}

template <class K>
static hipError_t allow_lds(K kernel, uint32_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
}

hipError_t launch_tier_scan(const TierDev &dev, const ScanArgs &a, int mode, bool heads, uint32_t blocks, uint32_t threads,
                            hipStream_t stream) {
    const uint32_t lds = dev.lds_bytes;
    hipError_t e;
    if (dev.row32) {
        using E = TierEngine<true>;
        if ((e = allow_lds(scan_kernel<E, 0, false>, lds)) != hipSuccess) return e;
        if ((e = allow_lds(scan_kernel<E, 0, true>, lds)) != hipSuccess) return e;
        if ((e = allow_lds(scan_kernel<E, 1, false>, lds)) != hipSuccess) return e;
        if ((e = allow_lds(scan_kernel<E, 1, true>, lds)) != hipSuccess) return e;
        if ((e = allow_lds(scan_kernel<E, 2, false>, lds)) != hipSuccess) return e;
        if ((e = allow_lds(scan_kernel<E, 2, true>, lds)) != hipSuccess) return e;
        return launch_mode<E>(dev, a, mode, heads, dim3(blocks), dim3(threads), lds, stream);
    }
    using E = TierEngine<false>;
    if ((e = allow_lds(scan_kernel<E, 0, false>, lds)) != hipSuccess) return e;
    if ((e = allow_lds(scan_kernel<E, 0, true>, lds)) != hipSuccess) return e;
    if ((e = allow_lds(scan_kernel<E, 1, false>, lds)) != hipSuccess) return e;
    if ((e = allow_lds(scan_kernel<E, 1, true>, lds)) != hipSuccess) return e;
    if ((e = allow_lds(scan_kernel<E, 2, false>, lds)) != hipSuccess) return e;
    if ((e = allow_lds(scan_kernel<E, 2, true>, lds)) != hipSuccess) return e;
    return launch_mode<E>(dev, a, mode, heads, dim3(blocks), dim3(threads), lds, stream);
}

hipError_t launch_darray_scan(const DArrayDev &dev, const ScanArgs &a, int mode, bool heads, uint32_t blocks, uint32_t threads,
                              hipStream_t stream) {
    return launch_mode<DArrayEngine>(dev, a, mode, heads, dim3(blocks), dim3(threads), 256 * 16, stream);
}

hipError_t launch_char_scan(const CharDev &dev, const ScanArgs &a, int mode, bool heads, uint32_t blocks, uint32_t threads,
                            hipStream_t stream) {
    return launch_mode<CharEngine>(dev, a, mode, heads, dim3(blocks), dim3(threads), 1024, stream);
}

// The same for millions of counts (the tuple emitter: one per tile of 1 KiB of text), in three launches: sums of chunks of
// kScanChunk elements, their exclusive scan by the one-workgroup kernel above, the scan inside every chunk on top of its sum.
constexpr uint32_t kScanChunk = 2048;  // 256 threads x 8 consecutive elements
__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const unsigned long long *__restrict__ v, uint64_t n, unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long part[4];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanChunk;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint64_t i = base + static_cast<uint64_t>(k) * 256 + threadIdx.x;
        if (i < n) s += v[i];
    }
    s = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void scan_chunks_kernel(unsigned long long *__restrict__ v, uint64_t n, const unsigned long long *__restrict__ sums_excl) {
    __shared__ unsigned long long part[256];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanChunk + static_cast<uint64_t>(threadIdx.x) * 8;
    unsigned long long x[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = base + k < n ? v[base + k] : 0ull; s += x[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64) {  // 256 thread sums: four per lane of the first wave
        unsigned long long q[4], t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { q[k] = part[threadIdx.x * 4 + k]; t += q[k]; }
        unsigned long long incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long y = __shfl_up(incl, off, 64);
            if (static_cast<int>(threadIdx.x) >= off) incl += y;
        }
        unsigned long long run = incl - t;
#pragma unroll
        for (int k = 0; k < 4; ++k) { part[threadIdx.x * 4 + k] = run; run += q[k]; }
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x] + sums_excl[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) v[base + k] = run;
        run += x[k];
    }
}

// `scratch`: room for n / kScanChunk + 2 more values (may be null for short arrays)
hipError_t launch_exclusive_scan(unsigned long long *v, uint64_t n, unsigned long long *total, unsigned long long *scratch, hipStream_t stream) {
    if (scratch == nullptr || n <= 4 * kScanChunk) {
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, v, n, total);
        return hipGetLastError();
    }
    const uint32_t nb = static_cast<uint32_t>((n + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nb), dim3(256), 0, stream, v, n, scratch);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, scratch, static_cast<uint64_t>(nb), total);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(nb), dim3(256), 0, stream, v, n, scratch);
    return hipGetLastError();
}

}  // namespace daac
