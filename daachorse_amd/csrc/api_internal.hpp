// What the translation units of the C ABI share (include/daachorse_amd.h; api_upload.hip, api_scan.hip, api_select.hip, api_iter.hip,
// api_options.hip): the options, the handle, its device tables, the call-scoped buffers and the drivers one unit calls in another.
// Internal: nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "charwise.hpp"
#include "device_tables.hpp"
#include "gram.hpp"
#include "gram2.hpp"
#include "gram4.hpp"
#include "gram2w.hpp"
#include "pfx.hpp"
#include "pma.hpp"
#include "repack.hpp"

namespace daac {
const char *last_error_cstr();

// ------------------------------------------------------------------------------------ options
// One list (X-macro): field name = user-facing option name, default.  `upload` marks the options that are read when a handle's tables
// are laid out (daac_pma_set_option tells a caller who sets one on a handle that already has tables on a device).
//      X(name, default, read at upload)
#define DAAC_OPTIONS(X)                                                                                                                      \
    X(seg_bytes, 0, 0)                   /* 0 = auto */                                                                                      \
    X(lds_budget, 96 * 1024, 1)                                                                                                              \
    X(dense_depth, -1, 1)                                                                                                                    \
    X(rows_share_pct, 45, 1)                                                                                                                 \
    X(blocks_per_cu, 0, 0)               /* 0 = auto */                                                                                      \
    X(threads, 1024, 0)                                                                                                                      \
    X(iter_window, 64ll << 20, 0)        /* lazy iterator: haystack bytes per window (the first windows are smaller: 16, 32 MiB) */          \
    X(max_result_bytes, 8ll << 30, 0)                                                                                                        \
    X(gram_lds_budget, 158 * 1024, 1)                                                                                                        \
    X(gram_region, 0, 0)                 /* 0 = auto: 16 KiB for the first table set, 64 / 256 KiB for the second */                         \
    X(gram_slab, 4096, 0)                                                                                                                    \
    X(gram_ppl, 0, 0)                    /* 0 = auto, 16, 32 positions per lane and step */                                                  \
    X(gram_dense, -1, 0)                 /* -1 = decide per automaton */                                                                     \
    X(gram_rank_in_lds, -1, 1)           /* -1 = decide per automaton */                                                                     \
    X(gram_version, 0, 0)                /* 0 = auto (count + checksum: v1 where it applies, else v2; `.count()`: gram4_kernels.hip), 1 = v1 \
                                            only, 2 = v2 tables with gram2_kernels.hip, 4 (3: its name until ABI 5) = gram4 or an error */   \
    X(gram4_arith, 1, 0)                 /* gram4: byte classes by arithmetic where the dictionary's bytes are one range (0: class table) */ \
    X(gram4_filter, 1, 0)                /* gram4: the LDS filter in front of rank + gather where it fits (0: per-word rank directory) */      \
    X(gram2_dpp, 1, 0)                   /* v2: neighbour exchange through DPP wave shifts (0: ds_bpermute) */                               \
    X(find3, 1, 0)                       /* find_iter's count through find3_kernels.hip where the dictionary allows (2: whatever the text,   \
                                            0: the chain walkers always) */                                                                  \
    X(pfx_probe, 16384, 0)               /* AUTO: the micro-step walker takes over where more than this many of 65 536 sampled positions     \
                                            survive PFX's filter (0 = never ask, always PFX) */                                              \
    X(pfx, 1, 1)                         /* PFX engine: 1 = built for automata the GRAM tables do not serve, 2 = always, 0 = never */        \
    X(gram_tail, -1, 0)                  /* gram4: tail records from the hit record on (-1 = decide per launch); also as gram3_tail */       \
    X(gram2_rfull, 1, 0)                 /* one rank-directory entry per M word when LDS allows */                                           \
    X(emit, 1, 0)                        /* materialising overlapping scans: GRAM tuple emission where it applies (0: segment scanners) */    \
    X(emit_rec_per_kib, 32, 0)           /* emit3: deep-match records the list is first sized for, per KiB of haystack */                    \
    X(restart_bpc, 8, 0)                 /* 256-thread workgroups per CU of the chain walkers */                                             \
    X(restart_chain, 1, 0)               /* find_iter / leftmost_find_iter: speculate-reconcile-emit (0 = sync-point scanners only) */       \
    X(chain_rounds, 24, 0)                                                                                                                   \
    X(overlap_micro, 1, 0)               /* counts of overlapping scans outside GRAM: 1 micro-step walker, 2 also instead of TIERED, 0 off */ \
    X(pool, 1, 0)                        /* scratch / result buffers from the stream-ordered pool */                                         \
    X(pool_keep, 0, 0)                   /* bytes the pool keeps between calls (0 = auto) */                                                 \
    X(left3, 1, 0)                       /* leftmost_find_iter's count through left3_kernels.hip (as find3: 2 = whatever the text, 0 = off) */ \
    X(select_emit, 1, 0)                 /* the restart iterators' tuple list from find3 / left3 (0: the chain walkers') */                  \
    X(find3_window, 1ll << 30, 0)        /* find3: end positions per window (tests: small windows = many restarts) */                        \
    X(workspace_keep, 8ll << 30, 0)      /* bytes of scratch a handle may keep for its emitter / find3 calls (0: none) */                    \
    X(char_map_lds, 1, 1)                /* charwise chain scans: the populated stretch of the code mapper in LDS */                         \
    X(char_row_lds, 1, 1)                /* ... and ROOT's row of children beside it */

enum OptionId : int {
#define X(NAME, DEF, UP) OPT_##NAME,
    DAAC_OPTIONS(X)
#undef X
    OPT_COUNT
};
static_assert(OPT_COUNT <= 64, "per-handle override mask is one 64-bit word");
struct Options {
    std::atomic<int64_t> v[OPT_COUNT];
    Options() {
#define X(NAME, DEF, UP) v[OPT_##NAME].store(static_cast<int64_t>(DEF));
        DAAC_OPTIONS(X)
#undef X
    }
};
inline Options g_opt;
// Options are process-wide defaults (daac_set_option) that a HANDLE may override (daac_pma_set_option): a scan looks an option up through OPT(),
// which takes the override of the handle the calling thread is working for (PmaScope, set by every entry point that is given a handle, an
// iterator or a stream — and by the worker threads of the iterator and of daac_scan_count_multi) and the process-wide value otherwise.  An
// override is a slot of an array in the handle + a bit of a mask: a lookup takes no lock and builds no string (round-5 advisor).
struct OptionOverrides {
    std::atomic<uint64_t> mask{0};
    std::atomic<int64_t> v[OPT_COUNT];
    OptionOverrides() { for (auto &x : v) x.store(0); }
};
inline thread_local const OptionOverrides *tl_ov = nullptr;
inline thread_local const ::daac_pma *tl_pma = nullptr;
const OptionOverrides *overrides_of(const ::daac_pma *p);   // (defined below daac_pma)
struct PmaScope {
    const ::daac_pma *prev;
    const OptionOverrides *prev_ov;
    explicit PmaScope(const ::daac_pma *p) : prev(tl_pma), prev_ov(tl_ov) { tl_pma = p; tl_ov = p ? overrides_of(p) : nullptr; }
    ~PmaScope() { tl_pma = prev; tl_ov = prev_ov; }
    PmaScope(const PmaScope &) = delete;
    PmaScope &operator=(const PmaScope &) = delete;
};
static inline int64_t opt_get(int id) {
    const OptionOverrides *o = tl_ov;
    if (o && ((o->mask.load(std::memory_order_acquire) >> id) & 1ull)) return o->v[id].load(std::memory_order_relaxed);
    return g_opt.v[id].load(std::memory_order_relaxed);
}
#define OPT(X) opt_get(OPT_##X)
inline thread_local int g_last_engine = DAAC_ENGINE_AUTO;  // engine of this thread's most recent scan (daac_last_engine)
inline thread_local std::string g_last_kernel;               // ... and the kernel + launch shape of its most recent count (daac_last_kernel)

inline daac_status hip_fail(hipError_t e, const char *what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    (void)hipGetLastError();  // the runtime also remembers the error: left there, the next successful launch would report it as its own
    return DAAC_ERR_DEVICE;
}
#define HIP_TRY(expr)                                                      \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) return hip_fail(_e, #expr);                  \
    } while (0)

// Scratch and result buffers of the scans come from the device's stream-ordered pool (hipMallocAsync): a scan that needs
// tens of MB of scratch, or hands back GBs of tuples, does not pay the driver's map / unmap each time — the pool keeps up
// to `pool_keep` bytes (default 1/8 of the device memory, at most 32 GiB) for the next call.  Option pool = 0: plain hipMalloc.
// decided per device (a process may scan on several): -1 undecided, 0 hipMalloc / hipFree, 1 stream-ordered pool
constexpr int kMaxDevices = 64;
inline std::atomic<int> g_pool_mode[kMaxDevices];
inline struct PoolModeInit { PoolModeInit() { for (auto &m : g_pool_mode) m.store(-1); } } g_pool_mode_init;
inline int pool_mode_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    int mode = g_pool_mode[dev].load();
    if (mode >= 0) return mode;
    mode = 0;
    if (g_opt.v[OPT_pool].load() != 0) {
        int supported = 0;
        hipMemPool_t pool;
        if (hipDeviceGetAttribute(&supported, hipDeviceAttributeMemoryPoolsSupported, dev) == hipSuccess && supported &&
            hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            uint64_t keep = static_cast<uint64_t>(g_opt.v[OPT_pool_keep].load());
            if (keep == 0) keep = std::min<uint64_t>(32ull << 30, tot / 8);
            if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) mode = 1;
        }
        (void)hipGetLastError();
    }
    g_pool_mode[dev].store(mode);
    return mode;
}
// DAAC_DEBUG_TIMING=1: host wall time between marks of one call, to stderr
inline void dbg_mark(const char *what) {
    static const bool on = std::getenv("DAAC_DEBUG_TIMING") != nullptr;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mark] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - last).count());
    last = now;
}
inline hipError_t dev_malloc(void **p, size_t bytes, hipStream_t s) {
    if (bytes == 0) bytes = 16;
    return pool_mode_of_current_device() == 1 ? hipMallocAsync(p, bytes, s) : hipMalloc(p, bytes);
}
inline void dev_free(void *p, hipStream_t s) {
    if (!p) return;
    if (pool_mode_of_current_device() == 1) (void)hipFreeAsync(p, s); else (void)hipFree(p);
}
struct DevBuf {  // scratch that lives as long as the call
    void *p = nullptr;
    hipStream_t s = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { dev_free(p, s); }
    hipError_t alloc(size_t bytes, hipStream_t stream) { s = stream; return dev_malloc(&p, bytes, stream); }
};

// ----------------------------------------------------------------------------- device tables
struct DeviceTables {
    int device = -1;
    int num_cu = 0;
    std::vector<void *> allocs;
    bool tier_ok = false;
    TierDev tier{};
    DArrayDev da{};
    TierTables tier_host_meta;  // sizes only (vectors cleared after upload)
    bool gram_ok = false;
    GramDev gram{};
    bool gram2_ok = false;     // second table set (gram2.hpp)
    Gram2Dev gram2{};
    bool gram4_ok = false;     // `.count()` tables of round 5 (gram4.hpp), derived from the second table set
    Gram4Dev gram4{};
    bool gramw_ok = false;     // wide alphabets (gram2w.hpp)
    bool pfx_ok = false;       // any byte alphabet, `.count()` (pfx.hpp)
    uint32_t n_distinct_bytes = 0;  // distinct pattern bytes (known when the PFX builder ran)
    PfxDev pfx{};
    Gram2WDev gramw{};
    Gram2EmitDev emit{};
    bool emit3_ok = false;     // ... with detection done once (emit3_kernels.hip)
    Gram3Lds emit3_lds{};
    bool emit3_has_len1 = false;   // some pattern is a single byte
    const uint32_t *pfx_probe_word = nullptr;   // device word the probe kernel leaves its count in
    std::atomic<int> pfx_dense{-1};  // the last probe's verdict on the text (scan_count_impl): 1 = most positions survive the filter
    bool find3_ok = false;         // find_iter's count / checksum without a state chain (find3_kernels.hip): K = 3, no pattern beyond 19 bytes
    Find3Dev find3{};
    Find3Dev find3v{};             // the same tables with the patterns' VALUES (the emitter's V1 / V2 / V3 rank structure): the selection's tuple list
    bool left3_ok = false;         // a leftmost handle whose patterns, as a Standard automaton, got the emitter's and find3's tables: left3_kernels.hip serves leftmost_find_iter
    std::atomic<uint32_t> find3_gave_up{0};
    std::atomic<uint32_t> find3_retry{0}, emit3_retry{0};   // large requests turned away since the engines gave up (every sixteenth tries again)
    std::atomic<uint32_t> find3_skips{0};
    std::atomic<uint32_t> find3_rec_per_kib{0};   // deep matches per KiB the last find3 request met, + 1 (0: none yet): text made of the dictionary's
                                                   // own words keeps DETECT's walkers busy (3.4 ms per GiB against 1.4) and the chain walkers are faster there
    bool pfx_emit_ok = false;      // PFX tuples: pfx_emit_kernel + EXPAND over the raw haystack (no pattern registered twice)
    Gram2EmitDev pfx_emit{};       // what that EXPAND needs: V1 by byte + the 256 flag bytes (v1, v1_bytes = 1280), K = 1
    std::atomic<uint32_t> emit3_rec_per_kib{0};  // deep-match records per KiB the last scans met (sizes the next scan's list)
    // scans in a row on which an emitter gave up on the TEXT (more deep matches or extras than it places: known only after its detection
    // has run): from the second on the handle stops trying and the plan says so (a served scan resets the count)
    std::atomic<uint32_t> emit3_gave_up{0};
    CharDev chr{};  // charwise automata only
    // The emitter's and find3's scratch (annotated stream, record list, scan arrays: ~2 bytes per haystack byte), kept by the handle from
    // one call to the next — the stream-ordered pool's calls cost host time in proportion to the bytes asked for, and what a call frees is
    // handed back at the next synchronisation (tools/micro/pool_ops.hip: 0.3 + 0.45 ms per GiB).  One call at a time borrows it
    // (Scratch below); a second concurrent call on the handle goes to the pool.  Option workspace_keep bounds it; 0 = none.
    std::atomic<bool> ws_busy{false};
    void *ws_p = nullptr;
    size_t ws_bytes = 0;
    std::atomic<uint64_t> ws_want{0};   // what the largest call so far needed

    ~DeviceTables() {
        for (void *p : allocs) (void)hipFree(p);
        if (ws_p) (void)hipFree(ws_p);
    }
    template <class T>
    daac_status put(const std::vector<T> &v, const T *&out) {
        // padded so that 16-byte granule copies into LDS never run past the allocation
        const size_t bytes = v.size() * sizeof(T);
        const size_t padded = ((bytes + 15) & ~size_t(15)) + 16;
        void *d = nullptr;
        HIP_TRY(hipMalloc(&d, padded));
        allocs.push_back(d);
        HIP_TRY(hipMemset(d, 0, padded));
        if (bytes) HIP_TRY(hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
        out = static_cast<const T *>(d);
        return DAAC_OK;
    }
};

}  // namespace daac

using namespace daac;

// daac_scan_count_multi: one persistent host thread per (handle, device), with a stream of its own — created the first time a shard names
// the device, joined when the handle is freed.  A job makes no assumption about the calling thread: the worker has made its device current
// and put the handle's options in scope (PmaScope) once, at its start.  (Round 5 started and joined one std::thread per shard and call,
// on the device's default stream, without the handle's options in scope on any shard but the first.)
struct ShardWorker {
    int device = 0;
    const ::daac_pma *pma = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long *d_res = nullptr;    // 3 u64 per shard of the job in hand, device
    unsigned long long *h_res = nullptr;    // ... and page-locked host
    size_t res_cap = 0;                     // shards the two buffers hold
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void(ShardWorker &)>> jobs;
    bool stop = false, failed = false;      // failed: the device could not be made current / no stream
    ShardWorker(int dev, const ::daac_pma *p);
    ~ShardWorker();
    ShardWorker(const ShardWorker &) = delete;
    ShardWorker &operator=(const ShardWorker &) = delete;
    void post(std::function<void(ShardWorker &)> job);
    bool reserve(size_t shards);            // (worker thread only)
};

struct daac_pma {
    OptionOverrides opt_ov;   // per-handle option overrides (daac_pma_set_option)
    bool charwise = false;  // which of the two containers is populated
    HostPma host;           // DoubleArrayAhoCorasick<u32>
    HostCharPma chost;      // CharwiseDoubleArrayAhoCorasick<u32>
    std::mutex mu;
    std::map<int, std::unique_ptr<DeviceTables>> dev;
    std::mutex workers_mu;
    std::map<int, std::unique_ptr<ShardWorker>> workers;   // daac_scan_count_multi's, by device (declared after `dev`: joined before the tables go)

    bool is_standard() const { return charwise ? chost.is_standard() : host.is_standard(); }
    bool root_has_output() const { return charwise ? chost.states[kRoot].output_pos != 0 : output_pos_of(host.opos_ch(kRoot)) != 0; }
    uint32_t max_pattern_len() const { return charwise ? chost.max_pattern_len() : host.max_pattern_len(); }
    // Bytes a lane reads ahead of its segment: a state is a suffix of the text of at most Lmax bytes, and a
    // match that ends inside the segment starts at most Lmax - 1 bytes before it.  Charwise lanes (and bytewise
    // leftmost automata with "" in the set) take Lmax whole bytes — the leftmost iterator with "" in the set must also see a pattern that ends exactly
    // at a cut (charwise/iter.rs:351-353, skip_empty) — and never less than 3, the distance to the lead byte
    // of a character that straddles the cut.
    uint32_t halo() const {
        const uint32_t lmax = max_pattern_len();
        if (charwise) return std::max(lmax, 3u);
        if (!is_standard() && root_has_output()) return lmax;
        return lmax > 0 ? lmax - 1 : 0;
    }
};

inline const daac::OptionOverrides *daac::overrides_of(const ::daac_pma *p) { return &p->opt_ov; }

// Host-side list of match tuples.  Page-locked memory: the device writes tuples at HBM speed and a pageable
// destination (plus its zero fill) turned the copy back into the slowest part of a materialising scan.
// One released page-locked block is kept for the next list (pinning a GB costs ~50 ms, unpinning ~60 ms).
struct PinnedSpare {
    std::mutex mu;
    void *p = nullptr;
    size_t cap = 0;  // in tuples
    // never freed at exit: the HIP runtime may already be gone when static destructors run
};
inline PinnedSpare g_spare;

struct MatchBuf {
    daac_match *p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;
    MatchBuf() = default;
    MatchBuf(const MatchBuf &) = delete;
    MatchBuf &operator=(const MatchBuf &) = delete;
    ~MatchBuf() { release(); }
    void release() {
        if (p && pinned) {
            std::lock_guard<std::mutex> g(g_spare.mu);
            if (cap > g_spare.cap) { std::swap(g_spare.p, reinterpret_cast<void *&>(p)); std::swap(g_spare.cap, cap); }
        }
        if (p) { if (pinned) (void)hipHostFree(p); else std::free(p); }
        p = nullptr; n = cap = 0;
    }
    void clear() { n = 0; }
    size_t size() const { return n; }
    bool reserve(size_t want) {  // contents are not kept
        if (want <= cap) return true;
        release();
        const size_t c = std::max<size_t>(want, 4096);
        void *q = nullptr;
        {
            std::lock_guard<std::mutex> g(g_spare.mu);
            if (g_spare.p && g_spare.cap >= want && g_spare.cap <= 4 * c) {
                p = static_cast<daac_match *>(g_spare.p);
                cap = g_spare.cap;
                pinned = true;
                g_spare.p = nullptr;
                g_spare.cap = 0;
                return true;
            }
        }
        if (hipHostMalloc(&q, c * sizeof(daac_match), hipHostMallocDefault) == hipSuccess) pinned = true;
        else { (void)hipGetLastError(); q = std::malloc(c * sizeof(daac_match)); pinned = false; }
        if (!q) return false;
        p = static_cast<daac_match *>(q);
        cap = c;
        return true;
    }
};

struct daac_matches {
    MatchBuf v;
};


// ---- what the translation units of the C ABI share (api_upload.hip, api_scan.hip, api_select.hip, api_iter.hip, api_options.hip)
namespace daac {
namespace api {

struct Plan {
    ChainArgs chain{};      // restart scans whose chain has been resolved (chain.x_prev != nullptr): the emit pass runs
    bool tier;
    bool charwise = false;  // the charwise engine (scan_kernel<CharEngine> / char_restart_kernel)
    bool restart = false;   // find_iter / leftmost_find_iter: the restart scanners (DARRAY tables)
    bool leftmost = false;
    uint32_t blocks, threads;
    ScanArgs a;
};

// Resolves where the chain of a restart iterator enters every segment (chain_scan.hpp): speculative exits,
// then rounds of reconciliation until no exit moves.  On success pl.chain names the final exits and the
// emit passes may run; otherwise ("" in the pattern set, a link that would not end, no convergence) pl.chain
// stays empty and the sync-point scanners of restart_kernels.hip / charwise_kernels.hip do the scan.
struct ChainBuffers {
    void *buf = nullptr;
    hipStream_t s = nullptr;
    ~ChainBuffers() { dev_free(buf, s); }
};

// a match list in device memory (daac_scan_device, and the first half of every materialising scan)
struct DevMatches {
    daac_match *p = nullptr;   // (or 16-byte tuples when f16)
    uint64_t n = 0;
    bool f16 = false;          // in: the caller wants {end u64, length u32, value u32} tuples; out: that is what p holds
    DevMatches() = default;
    DevMatches(const DevMatches &) = delete;
    DevMatches &operator=(const DevMatches &) = delete;
    hipStream_t s = nullptr;
    bool f16_done = false;     // the emitter wrote 16-byte tuples itself
    ~DevMatches() { dev_free(p, s); }
    daac_match *release() { daac_match *q = p; p = nullptr; n = 0; return q; }
    daac_match *release_keep_n() { daac_match *q = p; p = nullptr; return q; }
};

// A call's scratch, carved from the handle's kept workspace when nobody else is using it (DeviceTables::ws_*), from the pool otherwise.
// The borrower gives it back only when its stream has drained (every caller below has read its results back by then: the wait is a formality).
struct Scratch {
    DeviceTables *t;
    hipStream_t s;
    bool borrowed = false;
    size_t used = 0, pool_bytes = 0;
    std::vector<void *> pool_allocs;
    Scratch(DeviceTables *t_, hipStream_t s_, size_t expect) : t(t_), s(s_) {
        const uint64_t keep = static_cast<uint64_t>(OPT(workspace_keep));
        const uint64_t want = std::max<uint64_t>(expect, t->ws_want.load());
        if (keep == 0 || expect > keep || t->ws_busy.exchange(true)) return;
        borrowed = true;
        const uint64_t target = std::min<uint64_t>(keep, want + want / 8);
        if (t->ws_bytes < expect || (t->ws_bytes < want && target > t->ws_bytes)) {
            if (t->ws_p) (void)hipFree(t->ws_p);
            t->ws_p = nullptr; t->ws_bytes = 0;
            if (hipMalloc(&t->ws_p, target) == hipSuccess) t->ws_bytes = target;
            else { (void)hipGetLastError(); t->ws_p = nullptr; }
        }
    }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() {
        for (void *p : pool_allocs) dev_free(p, s);
        if (!borrowed) return;
        if (used != 0) (void)hipStreamSynchronize(s);
        uint64_t need = used + pool_bytes, seen = t->ws_want.load();
        while (need > seen && !t->ws_want.compare_exchange_weak(seen, need)) {}
        t->ws_busy.store(false);
    }
    hipError_t alloc(void **p, size_t bytes) {
        bytes = (std::max<size_t>(bytes, 16) + 255) & ~size_t(255);
        if (borrowed && used + bytes <= t->ws_bytes) { *p = static_cast<char *>(t->ws_p) + used; used += bytes; return hipSuccess; }
        const hipError_t e = dev_malloc(p, bytes, s);
        if (e == hipSuccess) { pool_allocs.push_back(*p); pool_bytes += bytes; }
        return e;
    }
    size_t mark() const { return used; }
    void rewind(size_t m) { used = m; }   // (what went to the pool after the mark stays until the call ends: the rare rerun's business)
};

// api_upload.hip
daac_status upload_locked(daac_pma *pma, int device, DeviceTables **out);   // pma->mu held
daac_status get_tables(daac_pma *pma, DeviceTables **out);                  // of the current device, uploading on first use
// api_scan.hip
struct CountRoute { bool g1_can, g2_can, gw_can, gram, pfx; };   // the GRAM table sets that can serve; GRAM at all; PFX (before its probe of the text)
CountRoute count_route(const daac_pma *pma, const DeviceTables *t, int mode, int engine, bool want_checksum, uint64_t span);
daac_status check_mode_kind(const daac_pma *pma, int mode);
daac_status diverged();
daac_status make_plan(const daac_pma *pma, const DeviceTables *t, int mode, int engine, uint64_t begin, uint64_t end, Plan &pl, bool &heads);
hipError_t launch(const DeviceTables *t, const Plan &pl, int kmode, bool heads, hipStream_t s, unsigned long long *next_begin = nullptr);
unsigned int *pinned_words();   // a few page-locked words per host thread for flags read back between passes
daac_status chain_resolve(const daac_pma *pma, const DeviceTables *t, Plan &pl, hipStream_t stream, ChainBuffers &cb);
daac_status scan_range_device(daac_pma *pma, DeviceTables *t, int mode, int engine, const uint8_t *dev_hay, uint64_t begin,
                              uint64_t end, uint64_t total_len, hipStream_t stream, DevMatches &out, uint64_t *next_begin);
daac_status scan_range_materialize(daac_pma *pma, DeviceTables *t, int mode, int engine, const uint8_t *dev_hay, uint64_t begin,
                                   uint64_t end, uint64_t total_len, hipStream_t stream, MatchBuf &out, uint64_t *next_begin);
daac_status stage_window(const uint8_t *host_hay, uint64_t copy_from, uint64_t end, hipStream_t stream, void **dbuf,
                         const uint8_t **virt_base);
// daac_match {start, end, value} -> {end u64, length u32, value u32}; 16-byte tuples -> daac_match8 {value, (end - base) | length << end_bits}
hipError_t launch_repack16(const daac_match *in, void *out, unsigned long long n, hipStream_t stream);
hipError_t launch_repack8(const void *in, void *out, unsigned long long n, unsigned long long base, uint32_t end_bits, hipStream_t stream);
// api_select.hip
daac_status emit_overlapping3(daac_pma *pma, DeviceTables *t, const uint8_t *dev_hay, uint64_t begin, uint64_t end, hipStream_t stream,
                              DevMatches &out, bool *served, bool raw = false, void *dest = nullptr, uint64_t dest_cap = 0);
daac_status find_count3(daac_pma *pma, DeviceTables *t, const uint8_t *dev_hay, uint64_t begin, uint64_t len, hipStream_t stream,
                        unsigned long long *d_res, bool want_checksum, bool leftmost, unsigned long long acc[3], bool *served);
daac_status select_emit(daac_pma *pma, DeviceTables *t, int mode, const uint8_t *dev_hay, uint64_t begin, uint64_t end, uint64_t total_len,
                        hipStream_t stream, DevMatches &out, uint64_t *next_begin, bool *served);

}  // namespace api
}  // namespace daac

using namespace daac::api;
