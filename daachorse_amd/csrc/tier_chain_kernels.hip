// FindIterator (reference src/bytewise/iter.rs:58-113) as a speculate / reconcile / emit chain (chain_scan.hpp) over the
// TIERED tables (repack.hpp) instead of the double array.
//
// On the double array every step of a chain is a dependent L2 gather ({base, check} of the child, failure links on the
// way), and with every lane of the chip chasing its own chain the scan runs at the rate at which the L2 takes uncoalesced
// requests (130-270 G/s: 73-76 GB/s of haystack on cfg3).  The TIERED re-pack keeps the shallow states — where a chain
// spends most of its bytes, all the more since FindIterator restarts at ROOT after every match — as dense, failure-resolved
// rows in LDS: a step there is one LDS read, and only the steps below the LDS tiers go to L2.
// Standard automata only (the leftmost kinds have no TIERED tables); "" among the patterns is left to the sync-point scanners.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "chain_scan.hpp"
#include "device_tables.hpp"

namespace daac {

namespace {

template <bool ROW32>
struct TierChainTables {
    using Row = typename std::conditional<ROW32, uint32_t, uint16_t>::type;
    static constexpr bool kMicro = false;
    using Stream = HayStream;
    struct State { uint32_t id, out; };  // out: the state carries an output list (known from the row entry / the parent's omap)

    const TierDev &d;
    const Row *l_rows;
    const uint32_t *l_bcmap;
    const uint32_t *l_bfail;
    const uint8_t *l_cls;
    const uint8_t *__restrict__ hay;

    __device__ TierChainTables(const TierDev &dev, char *smem, const uint8_t *h)
        : d(dev), l_rows(reinterpret_cast<const Row *>(smem)), l_bcmap(reinterpret_cast<const uint32_t *>(smem + dev.off_bcmap)),
          l_bfail(reinterpret_cast<const uint32_t *>(smem + dev.off_bfail)), l_cls(reinterpret_cast<const uint8_t *>(smem + dev.off_cls)), hay(h) {}

    __device__ __forceinline__ State root() const { return State{0, 0}; }
    __device__ __forceinline__ uint32_t symbol_at(HayWindow &win, uint64_t pos, uint32_t &clen) const { clen = 1; return win.byte_at(hay + pos); }
    __device__ __forceinline__ uint32_t opos(const State &st) const { return st.out ? d.sopos[st.id] : 0u; }
    __device__ __forceinline__ bool is_root(const State &st) const { return st.id == 0; }
    __device__ __forceinline__ uint64_t boundary_at_or_after(uint64_t x) const { return x; }
    __device__ __forceinline__ void step_leftmost(State &, uint32_t) const {}  // (never reached: Standard automata only)

    // delta with failure links (reference src/bytewise.rs:1063-1088) on the re-packed states: scan_kernels.hip, TierEngine::step
    __device__ __forceinline__ void step_plain(State &st, uint32_t c) const {
        const uint32_t k = l_cls[c];
        uint32_t s = st.id;
        for (;;) {
            if (s < d.NA) {  // dense row, failure links already resolved
                const uint32_t e = l_rows[s * d.C + k];
                constexpr uint32_t kShift = ROW32 ? 31 : 15;
                st = State{e & ((1u << kShift) - 1u), e >> kShift};
                return;
            }
            uint32_t cmap, omap, first, fail;
            if (s < d.NB) {  // child bitmap and failure link in LDS, the record only when the byte continues
                cmap = l_bcmap[s - d.NA];
                if (((cmap >> k) & 1u) == 0) { s = l_bfail[s - d.NA]; continue; }
                const uint4 r = d.grec[s];
                omap = r.y; first = r.z;
            } else {
                const uint4 r = d.grec[s];
                cmap = r.x; omap = r.y; first = r.z; fail = r.w;
                if (((cmap >> k) & 1u) == 0) { s = fail; continue; }
            }
            st = State{first + __popc(cmap & ((1u << k) - 1u)), (omap >> k) & 1u};
            return;
        }
    }
};

}  // namespace

template <bool ROW32, int PASS, int KMODE>
__global__ __launch_bounds__(1024) void tier_chain_kernel(const TierDev dev, const ScanArgs a, const ChainArgs c, unsigned long long *next_begin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned long long scratch[3 * 16];
    if (PASS != 3) {
        auto copy = [&](char *dst, const void *src, uint32_t bytes) {
            const uint4 *s = reinterpret_cast<const uint4 *>(src);
            uint4 *q = reinterpret_cast<uint4 *>(dst);
            for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) q[i] = s[i];
        };
        copy(smem, dev.rows, dev.off_bcmap);
        copy(smem + dev.off_bcmap, dev.bcmap, dev.off_bfail - dev.off_bcmap);
        copy(smem + dev.off_bfail, dev.bfail, dev.off_ssum - dev.off_bfail);
        copy(smem + dev.off_cls, dev.cls, 256);
        __syncthreads();
    }
    const TierChainTables<ROW32> T(dev, smem, a.hay);
    if (PASS == 0) chain_spec_body<TierChainTables<ROW32>, false>(T, a, c, dev.ohash);
    else if (PASS == 1) chain_fix_body<TierChainTables<ROW32>, false>(T, a, c, dev.ohash);
    else if (PASS == 3) chain_sum_body<KMODE>(a, c, next_begin, scratch);
    else chain_emit_body<TierChainTables<ROW32>, false, KMODE>(T, a, c, dev.outputs, next_begin, scratch);
}

template <bool ROW32>
static hipError_t launch_tc(const TierDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, unsigned long long *next_begin,
                            uint32_t blocks, hipStream_t stream) {
    const dim3 g(blocks), b(1024);
    const uint32_t lds = pass == 3 ? 0u : dev.lds_bytes;
#define DAAC_TC(P, M)                                                                                                                     \
    do {                                                                                                                                  \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tier_chain_kernel<ROW32, P, M>),                          \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dev.lds_bytes));            \
        if (e != hipSuccess) return e;                                                                                                    \
        hipLaunchKernelGGL((tier_chain_kernel<ROW32, P, M>), g, b, lds, stream, dev, a, c, next_begin);                                   \
    } while (0)
    if (pass == 0) DAAC_TC(0, 0);
    else if (pass == 1) DAAC_TC(1, 0);
    else if (pass == 3) { if (kmode == 0) DAAC_TC(3, 0); else DAAC_TC(3, 1); }
    else { if (kmode == 0) DAAC_TC(2, 0); else if (kmode == 1) DAAC_TC(2, 1); else DAAC_TC(2, 2); }
#undef DAAC_TC
    return hipGetLastError();
}

hipError_t launch_tier_chain(const TierDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, unsigned long long *next_begin,
                             uint32_t blocks, hipStream_t stream) {
    return dev.row32 ? launch_tc<true>(dev, a, c, pass, kmode, next_begin, blocks, stream)
                     : launch_tc<false>(dev, a, c, pass, kmode, next_begin, blocks, stream);
}

}  // namespace daac
