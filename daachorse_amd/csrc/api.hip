// C ABI of the MI355X-native daachorse scan path (include/daachorse_amd.h): handle management,
// device upload of the re-packed automaton, scan drivers.  No CPU scan fallback lives here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
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
struct Options {
    std::atomic<int64_t> seg_bytes{0};       // 0 = auto
    std::atomic<int64_t> lds_budget{96 * 1024};
    std::atomic<int64_t> dense_depth{-1};
    std::atomic<int64_t> rows_share_pct{45};
    std::atomic<int64_t> blocks_per_cu{0};   // 0 = auto
    std::atomic<int64_t> threads{1024};
    std::atomic<int64_t> iter_window{64ll << 20};   // lazy iterator: haystack bytes per window (the first windows are smaller: 16, 32 MiB)
    std::atomic<int64_t> max_result_bytes{8ll << 30};
    std::atomic<int64_t> gram_lds_budget{158 * 1024};
    std::atomic<int64_t> gram_region{0};           // 0 = auto: 16 KiB for the first table set, 64 KiB for the second
    std::atomic<int64_t> gram_slab{4096};
    std::atomic<int64_t> gram_ppl{0};           // 0 = auto (32 positions per lane for automata without short patterns), 16, 32
    std::atomic<int64_t> gram_dense{-1};        // -1 = decide per automaton
    std::atomic<int64_t> gram_rank_in_lds{-1};  // -1 = decide per automaton
    std::atomic<int64_t> gram_version{0};       // 0 = auto (count + checksum: v1 where it applies, else v2; `.count()`: gram4_kernels.hip), 1 = v1 only,
                                                // 2 = v2 tables with gram2_kernels.hip, 4 = gram4_kernels.hip for `.count()` or an error
    std::atomic<int64_t> gram4_arith{1};        // gram4: byte classes by arithmetic where the dictionary's bytes are one range (0: the class table in LDS)
    std::atomic<int64_t> gram2_dpp{1};
    std::atomic<int64_t> find3{1};              // find_iter's count (+ checksum) of a whole haystack of at most 1 GiB through find3_kernels.hip (selection over the
                                                // emitter's per-position flags, no state chain) where the dictionary allows; 0: the chain walkers always
    std::atomic<int64_t> pfx_probe{16384};      // AUTO, `.count()` / count + checksum of a dictionary PFX serves: the micro-step walker takes over where more than
                                                // this many of 65 536 sampled positions survive PFX's filter (0 = never ask, always PFX)
    std::atomic<int64_t> pfx{1};                // PFX engine: 1 = built for automata the GRAM tables do not serve, 2 = always, 0 = never (read at upload)
    std::atomic<int64_t> gram3_tail{-1};        // gram4 (option names gram_tail / gram3_tail): tail records from the hit record on (-1 = decide per launch)
    std::atomic<int64_t> gram2_rfull{1};        // v2 count-only: one directory entry per M word when LDS allows          // v2: neighbour exchange through DPP wave shifts (0: ds_bpermute)
    std::atomic<int64_t> emit{1};               // materialising overlapping scans: GRAM tuple emission where it applies (0: segment scanners)
    std::atomic<int64_t> emit_v3_lds{1};        // emit3 EXPAND: values of the 3-byte patterns from a rank structure in LDS when it fits (0: from L2)
    std::atomic<int64_t> emit_stagger{0};       // emit3 EXPAND: the waves of a CU start this many x 1024 cycles apart (0: together)
    std::atomic<int64_t> emit_rec_per_kib{32};  // emit3: deep-match records the list is first sized for, per KiB of haystack (a rerun sizes it exactly)
    std::atomic<int64_t> restart_bpc{8};        // 256-thread workgroups per CU of the chain walkers
    std::atomic<int64_t> restart_chain{1};      // find_iter / leftmost_find_iter: speculate-reconcile-emit (0 = sync-point scanners only)
    std::atomic<int64_t> chain_rounds{24};
    std::atomic<int64_t> overlap_micro{1};      // counts of overlapping scans outside GRAM: 1 micro-step walker (charwise, DARRAY), 2 also instead of TIERED, 0 off
    std::atomic<int64_t> pool{1};               // scratch / result buffers from the stream-ordered pool
    std::atomic<int64_t> pool_keep{0};          // bytes the pool keeps between calls (0 = auto)
    std::atomic<int64_t> left3{1};                    // leftmost_find_iter's count (+ checksum) through left3_kernels.hip (as find3: 2 = whatever the text, 0 = off)
    std::atomic<int64_t> select_emit{1};              // the restart iterators' tuple list from find3 / left3 (0: the chain walkers')
    std::atomic<int64_t> find3_window{1ll << 30};     // find3: end positions per window (tests: small windows = many restarts)
    std::atomic<int64_t> workspace_keep{8ll << 30};   // bytes of scratch a handle may keep for its emitter / find3 calls (0: none)
    std::atomic<int64_t> char_map_lds{1};
    std::atomic<int64_t> char_row_lds{1};       // ... and ROOT's row of children beside it       // charwise chain scans: stage the populated stretch of the code mapper in LDS
                                                // (off: measured slower on cfg5, 88 vs 104 GB/s — the stretch is L1-resident anyway)
};
static Options g_opt;
// Options are process-wide defaults (daac_set_option) that a HANDLE may override (daac_pma_set_option): a scan looks an option up through OPT(),
// which takes the override of the handle the calling thread is working for (PmaScope, set by every entry point that is given a handle, an
// iterator or a stream — and by the iterator's worker thread) and the process-wide value otherwise.  Two threads that scan two handles with
// different settings no longer share one set of atomics.
static thread_local const ::daac_pma *tl_pma = nullptr;
struct PmaScope {
    const ::daac_pma *prev;
    explicit PmaScope(const ::daac_pma *p) : prev(tl_pma) { tl_pma = p; }
    ~PmaScope() { tl_pma = prev; }
    PmaScope(const PmaScope &) = delete;
    PmaScope &operator=(const PmaScope &) = delete;
};
bool pma_override(const ::daac_pma *p, const char *field, int64_t *value);   // (defined below daac_pma)
static inline int64_t opt_get(const char *field, const std::atomic<int64_t> &global) {
    int64_t v;
    if (tl_pma && pma_override(tl_pma, field, &v)) return v;
    return global.load();
}
#define OPT(X) opt_get(#X, g_opt.X)
static thread_local int g_last_engine = DAAC_ENGINE_AUTO;  // engine of this thread's most recent scan (daac_last_engine)

static daac_status hip_fail(hipError_t e, const char *what) {
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
static std::atomic<int> g_pool_mode[kMaxDevices];
static struct PoolModeInit { PoolModeInit() { for (auto &m : g_pool_mode) m.store(-1); } } g_pool_mode_init;
static int pool_mode_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    int mode = g_pool_mode[dev].load();
    if (mode >= 0) return mode;
    mode = 0;
    if (g_opt.pool.load() != 0) {
        int supported = 0;
        hipMemPool_t pool;
        if (hipDeviceGetAttribute(&supported, hipDeviceAttributeMemoryPoolsSupported, dev) == hipSuccess && supported &&
            hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            uint64_t keep = static_cast<uint64_t>(g_opt.pool_keep.load());
            if (keep == 0) keep = std::min<uint64_t>(32ull << 30, tot / 8);
            if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) mode = 1;
        }
        (void)hipGetLastError();
    }
    g_pool_mode[dev].store(mode);
    return mode;
}
// DAAC_DEBUG_TIMING=1: host wall time between marks of one call, to stderr
static void dbg_mark(const char *what) {
    static const bool on = std::getenv("DAAC_DEBUG_TIMING") != nullptr;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mark] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - last).count());
    last = now;
}
static hipError_t dev_malloc(void **p, size_t bytes, hipStream_t s) {
    if (bytes == 0) bytes = 16;
    return pool_mode_of_current_device() == 1 ? hipMallocAsync(p, bytes, s) : hipMalloc(p, bytes);
}
static void dev_free(void *p, hipStream_t s) {
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

// The patterns a bytewise automaton of either kind was built from, read back from its trie (goto edges of the double array, bytewise.rs:1070-1077)
// with their values: a state's own pattern is the output as long as the state is deep.  (LeftmostFirst: the builder never inserted what lies
// below an earlier-registered pattern, nfa_builder.rs:60-66 — what is read back is what can be reported.)  false: not a tree / "" / too large.
static bool recover_patterns(const HostPma &p, std::vector<uint8_t> &blob, std::vector<uint64_t> &offs, std::vector<uint32_t> &vals) {
    const uint32_t n = static_cast<uint32_t>(p.states_len());
    if (n == 0 || output_pos_of(p.opos_ch(kRoot)) != 0) return false;
    constexpr uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> depth(n, kNone), parent(n, kNone), order{kRoot};
    std::vector<uint8_t> label(n, 0);
    depth[kRoot] = 0;
    for (size_t qi = 0; qi < order.size(); ++qi) {
        const uint32_t s = order[qi], base = p.base(s);
        if (base == 0) continue;
        for (uint32_t c = 0; c < 256; ++c) {
            const uint32_t t = base ^ c;
            if (t >= n || t == kRoot || t == kDead || check_of(p.opos_ch(t)) != c) continue;
            if (depth[t] != kNone) return false;
            depth[t] = depth[s] + 1; parent[t] = s; label[t] = static_cast<uint8_t>(c);
            order.push_back(t);
        }
    }
    blob.clear(); offs.assign(1, 0); vals.clear();
    std::vector<uint8_t> tmp;
    for (const uint32_t s : order) {
        if (s == kRoot) continue;
        uint32_t op = output_pos_of(p.opos_ch(s));
        bool own = false; uint32_t value = 0;
        for (int hops = 0; op != 0 && hops < 4 && op - 1 < p.outputs.size(); ++hops) {
            if (p.outputs[op - 1].length == depth[s]) { own = true; value = p.outputs[op - 1].value; break; }
            if (p.outputs[op - 1].length < depth[s]) break;
            op = p.outputs[op - 1].parent;
        }
        if (!own) continue;
        tmp.clear();
        for (uint32_t x = s; x != kRoot; x = parent[x]) tmp.push_back(label[x]);
        blob.insert(blob.end(), tmp.rbegin(), tmp.rend());
        offs.push_back(blob.size());
        vals.push_back(value);
        if (blob.size() >= (1ull << 31)) return false;
    }
    return !vals.empty();
}

// hit records with the first child beside them: one request instead of a dependent second one
static std::vector<U32x4> zip_first_child(const std::vector<U32x2> &hit, const std::vector<uint32_t> &first) {
    std::vector<U32x4> out(hit.size());
    for (size_t i = 0; i < hit.size(); ++i) out[i] = U32x4{hit[i].x, hit[i].y, first[i], 0u};
    return out;
}

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

struct daac_pma {
    // per-handle option overrides (daac_pma_set_option): field name of Options -> value
    mutable std::mutex opt_mu;
    std::map<std::string, int64_t> opt_ov;
    std::atomic<int> opt_n{0};
    bool charwise = false;  // which of the two containers is populated
    HostPma host;           // DoubleArrayAhoCorasick<u32>
    HostCharPma chost;      // CharwiseDoubleArrayAhoCorasick<u32>
    std::mutex mu;
    std::map<int, std::unique_ptr<DeviceTables>> dev;

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

bool daac::pma_override(const ::daac_pma *p, const char *field, int64_t *value) {
    if (p->opt_n.load(std::memory_order_relaxed) == 0) return false;
    std::lock_guard<std::mutex> g(p->opt_mu);
    const auto it = p->opt_ov.find(field);
    if (it == p->opt_ov.end()) return false;
    *value = it->second;
    return true;
}

// Host-side list of match tuples.  Page-locked memory: the device writes tuples at HBM speed and a pageable
// destination (plus its zero fill) turned the copy back into the slowest part of a materialising scan.
// One released page-locked block is kept for the next list (pinning a GB costs ~50 ms, unpinning ~60 ms).
struct PinnedSpare {
    std::mutex mu;
    void *p = nullptr;
    size_t cap = 0;  // in tuples
    // never freed at exit: the HIP runtime may already be gone when static destructors run
};
static PinnedSpare g_spare;

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

// --------------------------------------------------------------------------------------- upload
static daac_status upload_locked(daac_pma *pma, int device, DeviceTables **out) {
    auto it = pma->dev.find(device);
    if (it != pma->dev.end()) { *out = it->second.get(); return DAAC_OK; }
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    struct DeviceGuard {  // the caller's current device comes back on every return path, errors included
        int prev;
        ~DeviceGuard() { (void)hipSetDevice(prev); }
    } device_guard{prev};
    std::unique_ptr<DeviceTables> t(new DeviceTables);
    t->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    t->num_cu = prop.multiProcessorCount;

    const HostPma &h = pma->host;
    daac_status st;
    // outputs, shared by all engines
    const uint32_t *d_outputs = nullptr, *d_ohash = nullptr;
    {
        const std::vector<OutputRec> &outs = pma->charwise ? pma->chost.outputs : h.outputs;
        std::vector<uint32_t> flat(outs.size() * 3);
        for (size_t i = 0; i < outs.size(); ++i) {
            flat[3 * i] = outs[i].value; flat[3 * i + 1] = outs[i].length; flat[3 * i + 2] = outs[i].parent;
        }
        if ((st = t->put(flat, d_outputs)) != DAAC_OK) return st;
        std::vector<uint32_t> oh(outs.size());
        for (size_t i = 0; i < outs.size(); ++i) oh[i] = match_hash32(outs[i].value, outs[i].length);
        if ((st = t->put(oh, d_ohash)) != DAAC_OK) return st;
    }
    if (pma->charwise) {
        CharTables ct;
        build_char_tables(pma->chost, ct);
        CharDev &c = t->chr;
        const CStateRec *states; const OutSum *osum;
        if ((st = t->put(ct.states, states)) != DAAC_OK) return st;
        if ((st = t->put(ct.table, c.table)) != DAAC_OK) return st;
        if ((st = t->put(ct.osum, osum)) != DAAC_OK) return st;
        c.fail_plain = nullptr;
        if (!ct.fail_plain.empty() && (st = t->put(ct.fail_plain, c.fail_plain)) != DAAC_OK) return st;
        c.states = reinterpret_cast<const uint4 *>(states);
        c.osum = reinterpret_cast<const uint2 *>(osum);
        c.outputs = d_outputs;
        c.ohash = d_ohash;
        c.table_len = static_cast<uint32_t>(ct.table.size());
        c.n = static_cast<uint32_t>(ct.states.size());
        c.root_flag = ct.root_flag;
        c.leftmost = !pma->chost.is_standard();
        // Stage [map_lo, table_len) of the mapper in LDS when that stretch is small: map_lo = the lowest start for which
        // it fits 32 KB, moved up to the first mapped code point at or above it (CJK text: the table is dense from the
        // kana up, ASCII below stays in L2).
        {
            const uint32_t cap = 16u * 1024u - 128u;  // u16 entries (128 more hold ASCII)
            uint32_t lo = c.table_len > cap ? c.table_len - cap : 0u;
            while (lo < c.table_len && ct.table[lo] == kInvalidCode) ++lo;
            c.map_lo = lo;
            uint32_t staged = 0;
            for (uint32_t i = lo; i < c.table_len; ++i) staged += ct.table[i] != kInvalidCode;
            // worth it only if most of the alphabet lives in the stretch
            c.map_in_lds = OPT(char_map_lds) != 0 && lo < c.table_len && pma->chost.alphabet_size < 0xffffu &&
                           staged * 4u >= pma->chost.alphabet_size * 3u;
        }
        // the walkers' records: the output_pos word also carries the state's child filter (device_tables.hpp, CharDev::wstates)
        std::vector<uint32_t> filt(ct.states.size(), 0u);
        {
            const size_t n_out = pma->chost.outputs.size();
            c.obits = n_out < (1u << 16) ? 16u : n_out < (1u << 24) ? 24u : 0u;
            c.fbits = c.obits == 16u ? 16u : c.obits == 24u ? 8u : 0u;
            std::vector<CStateRec> ws(ct.states);
            if (c.fbits != 0) {
                // a slot t >= 2 whose CHECK names a state p other than DEAD is p's child on code t ^ base(p) (vacant slots: CHECK = DEAD,
                // reference src/charwise.rs:1103-1112)
                for (size_t tt = 2; tt < ct.states.size(); ++tt) {
                    const uint32_t pp = ct.states[tt].check;
                    if (pp == 1u || pp >= ct.states.size() || ct.states[pp].base == 0) continue;
                    const uint32_t code = static_cast<uint32_t>(tt) ^ ct.states[pp].base;
                    if (code >= pma->chost.alphabet_size) continue;
                    filt[pp] |= 1u << (code & (c.fbits - 1u));
                }
                for (size_t i = 0; i < ws.size(); ++i) ws[i].output_pos |= filt[i] << c.obits;
            }
            const CStateRec *dws;
            if ((st = t->put(ws, dws)) != DAAC_OK) return st;
            c.wstates = reinterpret_cast<const uint4 *>(dws);
        }
        // ROOT's row of children for the chain walkers: a lane at ROOT (where failed walks end) then needs no memory at all.
        // Staged beside the mapper when both fit 80 KB (two 1024-lane workgroups per CU) and every child packs into 8 bytes.
        {
            const uint32_t A = pma->chost.alphabet_size;
            std::vector<U32x2> row(A, U32x2{2u << 30, 0u});
            bool ok = c.map_in_lds != 0 && OPT(char_row_lds) != 0 && A != 0;
            const CStateRec &rt = ct.states[0];
            for (uint32_t code = 0; ok && code < A; ++code) {
                if (rt.base == 0) break;
                const uint32_t child = rt.base ^ code;
                if (child >= ct.states.size() || ct.states[child].check != 0) continue;
                const CStateRec &ch = ct.states[child];
                const uint32_t fl = (c.leftmost || ct.fail_plain.empty()) ? ch.fail : ct.fail_plain[child];
                if (fl > 1u || (fl == 1u && !c.leftmost) || ch.base >= (1u << 30)) ok = false;
                row[code] = U32x2{ch.base | (fl << 30), ch.output_pos | (c.fbits ? filt[child] << c.obits : 0u)};
            }
            const uint32_t map_bytes = ((128u + c.table_len - c.map_lo) * 2u + 15u) & ~15u;
            ok = ok && map_bytes + A * 8u <= 80u * 1024u;
            c.alphabet = A;
            c.row_in_lds = ok;
            c.root_row = nullptr;
            if (ok) {
                const U32x2 *drow;
                if ((st = t->put(row, drow)) != DAAC_OK) return st;
                c.root_row = reinterpret_cast<const uint2 *>(drow);
            }
        }
        HIP_TRY(hipDeviceSynchronize());
        *out = t.get();
        pma->dev[device] = std::move(t);
        return DAAC_OK;
    }
    // DARRAY engine: always available
    {
        DArrayTables da;
        build_darray_tables(h, da);
        const U32x2 *hot; const uint32_t *fail; const U32x4 *root; const OutSum *osum;
        if ((st = t->put(da.hot, hot)) != DAAC_OK) return st;
        if ((st = t->put(da.fail, fail)) != DAAC_OK) return st;
        t->da.fail_plain = fail;
        if (!da.fail_plain.empty() && (st = t->put(da.fail_plain, t->da.fail_plain)) != DAAC_OK) return st;
        t->da.leftmost = !h.is_standard();
        if ((st = t->put(da.root, root)) != DAAC_OK) return st;
        if ((st = t->put(da.osum, osum)) != DAAC_OK) return st;
        {
            std::vector<U32x4> rec(da.hot.size());
            for (size_t i = 0; i < da.hot.size(); ++i) rec[i] = U32x4{da.hot[i].x, da.hot[i].y, da.fail[i], da.fmap[i]};
            const U32x4 *drec;
            if ((st = t->put(rec, drec)) != DAAC_OK) return st;
            t->da.rec = reinterpret_cast<const uint4 *>(drec);
            const U32x4 *droot;
            if ((st = t->put(da.root_chain, droot)) != DAAC_OK) return st;
            t->da.root_chain = reinterpret_cast<const uint4 *>(droot);
        }
        t->da.hot = reinterpret_cast<const uint2 *>(hot);
        t->da.fail = fail;
        t->da.root = reinterpret_cast<const uint4 *>(root);
        t->da.osum = reinterpret_cast<const uint2 *>(osum);
        t->da.outputs = d_outputs;
        t->da.ohash = d_ohash;
        t->da.n = static_cast<uint32_t>(h.states_len());
        t->da.root_flag = output_pos_of(h.opos_ch(kRoot)) != 0;
    }
    // TIERED engine
    {
        RepackOptions ro;
        ro.lds_budget = static_cast<uint32_t>(OPT(lds_budget));
        ro.dense_depth = static_cast<int>(OPT(dense_depth));
        ro.rows_share_pct = static_cast<uint32_t>(OPT(rows_share_pct));
        TierTables tt;
        if (build_tier_tables(h, ro, tt)) {
            TierDev &d = t->tier;
            const uint16_t *r16 = nullptr; const uint32_t *r32 = nullptr;
            if (tt.row32) { if ((st = t->put(tt.rows32, r32)) != DAAC_OK) return st; d.rows = r32; }
            else { if ((st = t->put(tt.rows16, r16)) != DAAC_OK) return st; d.rows = r16; }
            const U32x4 *grec; const OutSum *ssum;
            if ((st = t->put(tt.bcmap, d.bcmap)) != DAAC_OK) return st;
            if ((st = t->put(tt.bfail, d.bfail)) != DAAC_OK) return st;
            if ((st = t->put(tt.ssum, ssum)) != DAAC_OK) return st;
            if ((st = t->put(tt.cls, d.cls)) != DAAC_OK) return st;
            if ((st = t->put(tt.grec, grec)) != DAAC_OK) return st;
            if ((st = t->put(tt.sopos, d.sopos)) != DAAC_OK) return st;
            d.ssum = reinterpret_cast<const uint2 *>(ssum);
            d.grec = reinterpret_cast<const uint4 *>(grec);
            d.outputs = d_outputs;
            d.ohash = d_ohash;
            d.C = tt.C; d.NA = tt.NA; d.NB = tt.NB; d.N = tt.N;
            auto pad16 = [](uint32_t x) { return (x + 15u) & ~15u; };
            d.off_bcmap = pad16(tt.NA * tt.C * (tt.row32 ? 4u : 2u));
            d.off_bfail = d.off_bcmap + pad16((tt.NB - tt.NA) * 4u);
            d.off_ssum = d.off_bfail + pad16((tt.NB - tt.NA) * 4u);
            d.off_cls = d.off_ssum + pad16(tt.NA * 8u);
            d.lds_bytes = std::max<uint32_t>(d.off_cls + 256u, 1024u);
            d.row32 = tt.row32;
            d.root_flag = tt.root_flag;
            t->tier_ok = true;
            // GRAM count engine, derived from the tier tables
            GramTables gt;
            // (the option bounds the tables; tables AND the hit rings of a 1024-thread workgroup have to fit the 160 KB a workgroup can have)
            const int64_t g1_budget = std::min<int64_t>(OPT(gram_lds_budget), 160 * 1024 - 16 * 128 * 8);
            if (tt.N < (1u << 27) && g1_budget > 0 && build_gram_tables(h, tt, static_cast<uint32_t>(g1_budget), gt)) {
                GramDev &g = t->gram;
                const U32x2 *combo; const U32x4 *drec; const U32x2 *dhit;
                std::vector<uint32_t> cls32(gt.cls.begin(), gt.cls.end());
                if ((st = t->put(cls32, g.cls32)) != DAAC_OK) return st;
                if ((st = t->put(gt.cid, g.cid)) != DAAC_OK) return st;
                if ((st = t->put(gt.combo, combo)) != DAAC_OK) return st;
                if ((st = t->put(gt.bbits, g.bbits)) != DAAC_OK) return st;
                if ((st = t->put(gt.brank, g.brank)) != DAAC_OK) return st;
                if ((st = t->put(gt.bsuper, g.bsuper)) != DAAC_OK) return st;
                if ((st = t->put(gt.drec, drec)) != DAAC_OK) return st;
                if ((st = t->put(gt.dhit, dhit)) != DAAC_OK) return st;
                { const U32x4 *h4; if ((st = t->put(zip_first_child(gt.dhit, gt.cfirst), h4)) != DAAC_OK) return st; g.dhit4 = reinterpret_cast<const uint4 *>(h4); }
                if ((st = t->put(gt.cfirst, g.cfirst)) != DAAC_OK) return st;
                g.combo = reinterpret_cast<const uint2 *>(combo);
                g.drec = reinterpret_cast<const uint4 *>(drec);
                g.dhit = reinterpret_cast<const uint2 *>(dhit);
                auto p16 = [](size_t x) { return static_cast<uint32_t>((x + 15) & ~size_t(15)); };
                g.has_short = gt.has_short;
                // LDS layout: [classes as u32 x 256][B bitmap][CID][COMBO][rank directory][hit stacks]; the first two sit at
                // fixed offsets so that the kernel addresses them with immediates
                g.off_bbits = 1024;
                g.off_cid = g.off_bbits + p16(gt.bbits.size() * 4);
                g.off_combo = g.off_cid + (gt.has_short ? p16(gt.cid.size() * 2) : 0u);   // not staged when unused
                g.off_brank = g.off_combo + (gt.has_short ? p16(gt.combo.size() * 8) : 0u);
                // Without the rank directory two workgroups may fit one CU (<= 80 KB each); worth it when
                // the level is small, i.e. B hits are rare whatever the text.
                g.rank_in_lds = !(g.off_brank + 16u * 1024u <= 80u * 1024u && gt.dhit.size() <= 8192);
                if (OPT(gram_rank_in_lds) >= 0) g.rank_in_lds = OPT(gram_rank_in_lds) != 0;
                if (g.rank_in_lds) {
                    g.off_bsuper = g.off_brank + p16(gt.brank.size());
                    g.off_scratch = g.off_bsuper + p16(gt.bsuper.size() * 4);
                } else {
                    g.off_bsuper = g.off_brank;
                    g.off_scratch = g.off_brank;
                }
                // + one 128-entry x 8-byte hit ring per wave of a 1024-thread workgroup
                g.lds_bytes = std::max<uint32_t>(g.off_scratch + 16u * 128u * 8u, 1024u);
                g.K = gt.K; g.C = gt.C; g.CC = gt.C * gt.C; g.CCC = gt.C * gt.C * gt.C;
                g.level_start = gt.level_start;
                g.unused_byte = gt.unused_byte;
                g.n_deep = static_cast<uint32_t>(gt.dhit.size());
                t->gram_ok = g.lds_bytes <= 160u * 1024u;  // (what a workgroup can have on gfx950)
            }
            // keep the sizes for daac_pma_info
            tt.rows16.clear(); tt.rows32.clear(); tt.bcmap.clear(); tt.bfail.clear(); tt.grec.clear(); tt.ssum.clear(); tt.sopos.clear(); tt.old_of_new.clear();
            t->tier_host_meta = tt;
        }
    }
    // GRAM engine, second table set: built from the automaton itself.  A leftmost handle gets the tables of a Standard automaton of ITS
    // patterns (read back from its trie) — not for any Standard scan (its kind forbids them) but for left3_kernels.hip, which selects
    // leftmost_find_iter's matches among the ones the emitter's detection finds.
    HostPma shadow;
    bool have_shadow = false;
    // (the shadow is only ever used by left3, which takes dictionaries of at most 19-byte patterns over at most 29 distinct bytes:
    // neither a second automaton nor its tables are built for a handle that cannot qualify)
    const bool shadow_can = !h.is_standard() && OPT(left3) != 0 && !pma->root_has_output() && h.max_pattern_len() <= 19;
    if (shadow_can) {
        std::vector<uint8_t> blob; std::vector<uint64_t> offs; std::vector<uint32_t> vals;
        if (recover_patterns(h, blob, offs, vals)) {
            bool seen[256] = {false};
            uint32_t distinct = 0;
            for (uint8_t c : blob) if (!seen[c]) { seen[c] = true; ++distinct; }
            if (distinct <= 29 && build_bytewise(blob.data(), offs.data(), vals.data(), vals.size(), DAAC_STANDARD, 16, shadow) == DAAC_OK) have_shadow = true;
        }
    }
    const HostPma &hg2 = have_shadow ? shadow : h;
    {
        Gram2Tables g2;
        const uint32_t ring_bytes = 16u * 128u * 8u;  // one 128-entry x 8-byte hit ring per wave of a 1024-thread workgroup
        const int64_t budget = OPT(gram_lds_budget) - static_cast<int64_t>(ring_bytes);
        if (budget > 0 && build_gram2_tables(hg2, static_cast<uint32_t>(budget), g2)) {
            Gram2Dev &d = t->gram2;
            auto p16 = [](size_t x) { return static_cast<uint32_t>((x + 15) & ~size_t(15)); };
            const U32x4 *drec; const U32x2 *dhit;
            if ((st = t->put(g2.cls, d.cls)) != DAAC_OK) return st;
            if ((st = t->put(g2.m, d.m)) != DAAC_OK) return st;
            if (g2.s16) {
                std::vector<uint16_t> s16(g2.sdir.begin(), g2.sdir.end());
                const uint16_t *ps;
                if ((st = t->put(s16, ps)) != DAAC_OK) return st;
                d.sdir = ps;
                d.s_bytes = p16(s16.size() * 2);
            } else {
                const uint32_t *ps;
                if ((st = t->put(g2.sdir, ps)) != DAAC_OK) return st;
                d.sdir = ps;
                d.s_bytes = p16(g2.sdir.size() * 4);
            }
            // CID entries are the LDS addresses of their H words (H sits at a fixed offset)
            std::vector<uint16_t> cid(g2.cid4.size());
            bool exact_ok = g2.exact_available && kGram2OffH + g2.hsum.size() * 4 <= 65536;
            for (size_t i = 0; i < cid.size(); ++i) cid[i] = static_cast<uint16_t>(kGram2OffH + g2.cid4[i]);
            if ((st = t->put(cid, d.cid4)) != DAAC_OK) return st;
            if ((st = t->put(g2.hsum, d.hsum)) != DAAC_OK) return st;
            if ((st = t->put(g2.drec, drec)) != DAAC_OK) return st;
            if ((st = t->put(g2.dhit, dhit)) != DAAC_OK) return st;
            { const U32x4 *h4; if ((st = t->put(zip_first_child(g2.dhit, g2.cfirst), h4)) != DAAC_OK) return st; d.dhit4 = reinterpret_cast<const uint4 *>(h4); }
            if ((st = t->put(g2.cfirst, d.cfirst)) != DAAC_OK) return st;
            d.drec = reinterpret_cast<const uint4 *>(drec);
            d.dhit = reinterpret_cast<const uint2 *>(dhit);
            d.m_bytes = p16(g2.m.size() * 4);
            d.cid_bytes = p16(cid.size() * 2);
            d.h_bytes = p16(g2.hsum.size() * 4);
            d.off_m_count = kGram2OffM;
            d.off_s_count = d.off_m_count + d.m_bytes;
            d.off_ring_count = d.off_s_count + d.s_bytes;
            d.lds_count = d.off_ring_count + ring_bytes;
            d.rfull = nullptr;
            d.rfull_ok = 0;
            if (g2.s16) {  // per-word directory for count-only launches (they have the LDS for it)
                std::vector<uint16_t> rf(g2.m.size());
                uint32_t run = 0;
                for (size_t i = 0; i < g2.m.size(); ++i) { rf[i] = static_cast<uint16_t>(run); run += static_cast<uint32_t>(__builtin_popcount(g2.m[i] & kGram2MaskBits)); }
                if ((st = t->put(rf, d.rfull)) != DAAC_OK) return st;
                d.rfull_bytes = p16(rf.size() * 2);
                d.off_ring_rfull = d.off_s_count + d.rfull_bytes;
                d.lds_rfull = d.off_ring_rfull + ring_bytes;
                d.rfull_ok = d.lds_rfull <= static_cast<uint32_t>(OPT(gram_lds_budget)) && OPT(gram2_rfull) != 0;
            }
            d.off_m_exact = kGram2OffH + d.h_bytes;
            d.off_s_exact = d.off_m_exact + d.m_bytes;
            d.off_cid = d.off_s_exact + d.s_bytes;
            d.off_ring_exact = d.off_cid + d.cid_bytes;
            d.lds_exact = d.off_ring_exact + ring_bytes;
            // the hit queue keeps the LDS address of an M word in 17 bits
            if (d.off_m_exact + d.m_bytes > (1u << 17) || d.lds_exact > 160u * 1024u) exact_ok = false;
            d.K = g2.K; d.C = g2.C; d.s16 = g2.s16; d.unused_byte = g2.unused_byte;
            d.n_deep = static_cast<uint32_t>(g2.dhit.size());
            d.exact_ok = exact_ok;
            d.xlane_dpp = OPT(gram2_dpp) != 0;
            t->gram2_ok = d.off_m_count + d.m_bytes <= (1u << 17) && d.lds_count <= 160u * 1024u;
            if (t->gram2_ok) {   // the same tables in the numbering of gram4_kernels.hip ("no pattern" last: arithmetic byte classes)
                Gram4Tables g4;
                build_gram4_tables(g2, g4);
                Gram4Dev &q = t->gram4;
                q = Gram4Dev{};
                if ((st = t->put(g4.cls, q.cls)) != DAAC_OK) return st;
                if ((st = t->put(g4.m, q.m)) != DAAC_OK) return st;
                q.m_bytes = p16(g4.m.size() * 4);
                if (g4.s16) {
                    std::vector<uint16_t> s16(g4.sdir.begin(), g4.sdir.end());
                    const uint16_t *ps;
                    if ((st = t->put(s16, ps)) != DAAC_OK) return st;
                    q.sdir = ps;
                    q.s_bytes = p16(s16.size() * 2);
                    if ((st = t->put(g4.rfull, q.rfull)) != DAAC_OK) return st;
                    q.rfull_bytes = p16(g4.rfull.size() * 2);
                } else {
                    const uint32_t *ps;
                    if ((st = t->put(g4.sdir, ps)) != DAAC_OK) return st;
                    q.sdir = ps;
                    q.s_bytes = p16(g4.sdir.size() * 4);
                }
                { const U32x2 *x; if ((st = t->put(g4.dhit_c, x)) != DAAC_OK) return st; q.dhit_c = reinterpret_cast<const uint2 *>(x); }
                { const U32x4 *x; if ((st = t->put(g4.dhit_t, x)) != DAAC_OK) return st; q.dhit_t = reinterpret_cast<const uint4 *>(x); }
                { const U32x4 *x; if ((st = t->put(g4.drec_c, x)) != DAAC_OK) return st; q.drec_c = reinterpret_cast<const uint4 *>(x); }
                { const U32x4 *x; if ((st = t->put(g4.drec_t, x)) != DAAC_OK) return st; q.drec_t = reinterpret_cast<const uint4 *>(x); }
                q.K = g4.K; q.C = g4.C; q.s16 = g4.s16 ? 1u : 0u; q.arith = g4.arith ? 1u : 0u; q.lo = g4.lo; q.unused_byte = g4.unused_byte;
                q.n_deep = static_cast<uint32_t>(g4.dhit_c.size());
                t->gram4_ok = g4.available;
            }
            if (t->gram2_ok && g2.emit_available) {
                Gram2EmitDev &e = t->emit;
                const U32x4 *erec; const U32x2 *ehit;
                e.cls = d.cls;
                e.sdir = d.sdir;
                e.cfirst = d.cfirst;
                if ((st = t->put(g2.me, e.me)) != DAAC_OK) return st;
                if ((st = t->put(g2.v1, e.v1)) != DAAC_OK) return st;
                if ((st = t->put(g2.v2, e.v2)) != DAAC_OK) return st;
                if ((st = t->put(g2.v3, e.v3)) != DAAC_OK) return st;
                if ((st = t->put(g2.erec, erec)) != DAAC_OK) return st;
                if ((st = t->put(g2.ehit, ehit)) != DAAC_OK) return st;
                {
                    std::vector<U32x4> h4v = zip_first_child(g2.ehit, g2.cfirst);
                    for (size_t i = 0; i < h4v.size(); ++i) h4v[i].w = g2.ecopies[i] << 24;
                    const U32x4 *h4;
                    if ((st = t->put(h4v, h4)) != DAAC_OK) return st;
                    e.ehit4 = reinterpret_cast<const uint4 *>(h4);
                }
                if ((st = t->put(g2.dupo, e.dupo)) != DAAC_OK) return st;
                if ((st = t->put(g2.dupv, e.dupv)) != DAAC_OK) return st;
                e.level_start = g2.level_start;
                e.erec = reinterpret_cast<const uint4 *>(erec);
                e.ehit = reinterpret_cast<const uint2 *>(ehit);
                e.m_bytes = d.m_bytes; e.s_bytes = d.s_bytes;
                e.v1_bytes = p16(g2.v1.size() * 4); e.v2_bytes = p16(g2.v2.size() * 4);
                e.off_s = kGram2OffM + e.m_bytes;
                e.off_v1 = e.off_s + e.s_bytes;
                e.off_v2 = e.off_v1 + e.v1_bytes;
                e.off_ring = e.off_v2 + e.v2_bytes;
                e.off_wave = e.off_ring + ring_bytes;
                e.K = g2.K; e.C = g2.C; e.s16 = g2.s16; e.unused_byte = g2.unused_byte;
                // emit3: DETECT as a 16-wave workgroup when the tables leave room for the text slots, else 8 waves
                t->emit3_ok = emit3_plan(e, 16, 160u * 1024u, t->emit3_lds) || emit3_plan(e, 8, 160u * 1024u, t->emit3_lds);
                // the values of the 3-byte patterns as a rank structure for EXPAND's LDS (device_tables.hpp: v3c)
                e.v3c = nullptr; e.v3c_bytes = e.v3c_dir = e.v3c_val = 0;
                if (g2.K == 3) {
                    const uint32_t n3 = static_cast<uint32_t>(g2.v3.size()), nw = (n3 + 31) / 32;
                    std::vector<uint32_t> bm(nw, 0), vals;
                    std::vector<uint16_t> dir(nw + (nw & 1), 0);
                    for (uint32_t i = 0; i < n3; ++i) {
                        if ((i & 31) == 0) dir[i >> 5] = static_cast<uint16_t>(vals.size());
                        if ((g2.me[i] >> 31) & 1u) { bm[i >> 5] |= 1u << (i & 31); vals.push_back(g2.v3[i]); }
                    }
                    if (vals.size() < 65536) {
                        std::vector<uint32_t> blob(bm);
                        e.v3c_dir = static_cast<uint32_t>(blob.size() * 4);
                        for (size_t i = 0; i < dir.size(); i += 2) blob.push_back(dir[i] | (static_cast<uint32_t>(dir[i + 1]) << 16));
                        e.v3c_val = static_cast<uint32_t>(blob.size() * 4);
                        blob.insert(blob.end(), vals.begin(), vals.end());
                        while (blob.size() & 3) blob.push_back(0);
                        e.v3c_bytes = static_cast<uint32_t>(blob.size() * 4);
                        if ((st = t->put(blob, e.v3c)) != DAAC_OK) return st;
                        // find3: the same three tables with h32 of the pattern in place of its value
                        if (g2.max_len <= 19) {   // (duplicates: find_iter reports a state's FIRST output, which is the record's own value)
                            std::vector<uint32_t> h1(g2.v1.size()), h2(g2.v2.size()), hb(blob);
                            for (size_t i = 0; i < h1.size(); ++i) h1[i] = match_hash32(g2.v1[i], 1);
                            for (size_t i = 0; i < h2.size(); ++i) h2[i] = match_hash32(g2.v2[i], 2);
                            for (size_t i = 0; i < vals.size(); ++i) hb[e.v3c_val / 4 + i] = match_hash32(vals[i], 3);
                            h1.resize(e.v1_bytes / 4, 0);
                            h2.resize(e.v2_bytes / 4, 0);
                            Find3Dev &f = t->find3;
                            if ((st = t->put(h1, f.h1)) != DAAC_OK) return st;
                            if ((st = t->put(h2, f.h2)) != DAAC_OK) return st;
                            if ((st = t->put(hb, f.h3c)) != DAAC_OK) return st;
                            f.h1_bytes = e.v1_bytes; f.h2_bytes = e.v2_bytes; f.h3c_bytes = e.v3c_bytes; f.h3c_dir = e.v3c_dir; f.h3c_val = e.v3c_val; f.C = g2.C;
                            t->find3v = f;
                            t->find3v.h1 = e.v1; t->find3v.h2 = e.v2; t->find3v.h3c = e.v3c;
                            t->find3_ok = true;   // (&& emit3_ok, decided below)
                        }
                    }
                }
                // (a staged tuple keeps its length in 22 bits)
                t->emit3_ok = t->emit3_ok && emit3_expand_lds_bytes(e, 4, false, false) <= 64u * 1024u && emit3_expand_lds_bytes(e, 8, true, false) <= 80u * 1024u && g2.max_len < (1u << 22);
                for (uint32_t w : g2.me) t->emit3_has_len1 = t->emit3_has_len1 || ((w >> 29) & 1u) != 0;
                t->find3_ok = t->find3_ok && t->emit3_ok && find3_lds_bytes(t->find3, true) <= 160u * 1024u;
                if (have_shadow) {   // (nothing Standard is ever asked of a leftmost handle; said explicitly all the same)
                    t->left3_ok = t->find3_ok && left3_lds_bytes(t->find3, true) <= 160u * 1024u;
                    t->find3_ok = false;
                }
            }
        }
    }
    // GRAM engine for wide alphabets: only where the 32-bit tables do not apply
    if (!t->gram_ok && !t->gram2_ok) {
        Gram2WTables gw;
        const uint32_t ring_bytes = 16u * 128u * 8u;
        const int64_t budget = OPT(gram_lds_budget) - static_cast<int64_t>(ring_bytes);
        if (budget > 0 && build_gram2w_tables(h, static_cast<uint32_t>(budget), gw)) {
            Gram2WDev &d = t->gramw;
            auto p16 = [](size_t x) { return static_cast<uint32_t>((x + 15) & ~size_t(15)); };
            const U32x4 *drec; const U32x4 *dhit; const uint64_t *pm;
            if ((st = t->put(gw.cls, d.cls)) != DAAC_OK) return st;
            if ((st = t->put(gw.m, pm)) != DAAC_OK) return st;
            d.m = reinterpret_cast<const unsigned long long *>(pm);
            if ((st = t->put(gw.sdir, d.sdir)) != DAAC_OK) return st;
            std::vector<uint16_t> cid(gw.cid4.size());
            for (size_t i = 0; i < cid.size(); ++i) cid[i] = static_cast<uint16_t>(kGram2OffH + gw.cid4[i]);
            if ((st = t->put(cid, d.cid4)) != DAAC_OK) return st;
            if ((st = t->put(gw.hsum, d.hsum)) != DAAC_OK) return st;
            if ((st = t->put(gw.drec, drec)) != DAAC_OK) return st;
            if ((st = t->put(gw.dhit, dhit)) != DAAC_OK) return st;
            d.drec = reinterpret_cast<const uint4 *>(drec);
            d.dhit = reinterpret_cast<const uint4 *>(dhit);
            d.m_bytes = p16(gw.m.size() * 8); d.s_bytes = p16(gw.sdir.size() * 4);
            d.cid_bytes = p16(cid.size() * 2); d.h_bytes = p16(gw.hsum.size() * 4);
            d.off_m_count = kGram2OffM;
            d.off_s_count = d.off_m_count + d.m_bytes;
            d.off_ring_count = d.off_s_count + d.s_bytes;
            d.lds_count = d.off_ring_count + ring_bytes;
            d.off_m_exact = kGram2OffH + d.h_bytes;
            d.off_s_exact = d.off_m_exact + d.m_bytes;
            d.off_cid = d.off_s_exact + d.s_bytes;
            d.off_ring_exact = d.off_cid + d.cid_bytes;
            d.lds_exact = d.off_ring_exact + ring_bytes;
            d.C = gw.C; d.unused_byte = gw.unused_byte; d.n_deep = static_cast<uint32_t>(gw.dhit.size());
            d.exact_ok = gw.exact_available && kGram2OffH + gw.hsum.size() * 4 <= 65536 && d.lds_exact <= 160u * 1024u;
            t->gramw_ok = d.lds_count <= 160u * 1024u;
        }
    }
    // PFX engine: `.count()` for every bytewise Standard automaton the GRAM tables do not serve (any alphabet); pfx = 2 builds it always
    if (OPT(pfx) == 2 || (OPT(pfx) == 1 && !t->gram_ok && !t->gram2_ok && !t->gramw_ok)) {
        PfxTables px;
        const bool px_ok = build_pfx_tables(h, 160u * 1024u - 16u * (2u * 1056u + 512u) - 64u - 1024u, px);
        t->n_distinct_bytes = px.n_distinct_bytes;
        if (px_ok) {
            PfxDev &d = t->pfx;
            auto p16 = [](size_t x) { return static_cast<uint32_t>((x + 15) & ~size_t(15)); };
            px.disp.resize((px.disp.size() + 7) & ~size_t(7), 0);
            const U32x4 *sl; const U32x2 *wr;
            if ((st = t->put(px.bloom, d.bloom)) != DAAC_OK) return st;
            if ((st = t->put(px.cnt1, d.cnt1)) != DAAC_OK) return st;
            if ((st = t->put(px.disp, d.disp)) != DAAC_OK) return st;
            if ((st = t->put(px.slots, sl)) != DAAC_OK) return st;
            if ((st = t->put(px.wrec, wr)) != DAAC_OK) return st;
            { const U32x4 *x; if ((st = t->put(px.slots_x, x)) != DAAC_OK) return st; d.slots_x = reinterpret_cast<const uint4 *>(x); }
            { const U32x4 *x; if ((st = t->put(px.wrec_x, x)) != DAAC_OK) return st; d.wrec_x = reinterpret_cast<const uint4 *>(x); }
            if ((st = t->put(px.hs1, d.hs1)) != DAAC_OK) return st;
            { const std::vector<uint32_t> z(4, 0); if ((st = t->put(z, t->pfx_probe_word)) != DAAC_OK) return st; }
            if (px.emit_ok) {
                { const U32x4 *x; if ((st = t->put(px.slots_e, x)) != DAAC_OK) return st; d.slots_e = reinterpret_cast<const uint4 *>(x); }
                std::vector<uint32_t> v1f(320, 0);   // V1 by byte, then the flag bytes
                std::memcpy(v1f.data(), px.v1.data(), 1024);
                std::memcpy(v1f.data() + 256, px.has1.data(), 256);
                if ((st = t->put(v1f, t->pfx_emit.v1)) != DAAC_OK) return st;
                t->pfx_emit.v1_bytes = 1280;
                t->pfx_emit.K = 1;
            }
            d.slots = reinterpret_cast<const uint4 *>(sl);
            d.wrec = reinterpret_cast<const uint2 *>(wr);
            d.G = px.G; d.has_len1 = px.has_len1; d.bloom_words = px.bloom_words; d.buckets = px.buckets; d.n_slots = px.n_slots;
            d.seed = px.seed; d.n_keys = px.n_keys;
            d.bloom_bytes = p16(px.bloom.size() * 4);
            d.disp_bytes = p16(px.disp.size() * 2);
            t->pfx_ok = pfx_plan(d, 160u * 1024u);
            t->pfx_emit_ok = t->pfx_ok && px.emit_ok && h.max_pattern_len() < (1u << 22);
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    *out = t.get();
    pma->dev[device] = std::move(t);
    return DAAC_OK;
}

static daac_status get_tables(daac_pma *pma, DeviceTables **out) {
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> g(pma->mu);
    return upload_locked(pma, device, out);
}

// ---------------------------------------------------------------------------------- scan driver
static daac_status scan_count_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin, int hay_is_device,
                                   void *stream_, uint64_t *count, uint64_t *checksum, uint64_t *result_dev, bool want_checksum);
namespace {

// The reference panics when a query does not fit the automaton's MatchKind (bytewise.rs:194-197,
// 299-302, 551-554); checked before anything touches the device.
daac_status check_mode_kind(const daac_pma *pma, int mode) {
    const bool standard = pma->is_standard();
    if (mode == DAAC_FIND_OVERLAPPING || mode == DAAC_FIND_OVERLAPPING_NO_SUFFIX || mode == DAAC_FIND) {
        if (!standard) { set_error("Error: match_kind must be standard."); return DAAC_ERR_MATCH_KIND; }
    } else if (mode == DAAC_LEFTMOST_FIND) {
        if (standard) { set_error("Error: match_kind must be leftmost."); return DAAC_ERR_MATCH_KIND; }
    } else {
        set_error("unknown scan mode");
        return DAAC_ERR_INVALID_ARGUMENT;
    }
    return DAAC_OK;
}

// SURVEY.md 8a note D: with "" in the set, a leftmost iterator whose haystack ends inside a longer pattern
// yields the same empty match forever in the reference (charwise/iter.rs:385-398)
daac_status diverged() {
    set_error("the reference iterator does not terminate on this input (leftmost kind, empty pattern, haystack ends inside a pattern)");
    return DAAC_ERR_UNSUPPORTED;
}

struct Plan {
    ChainArgs chain{};      // restart scans whose chain has been resolved (chain.x_prev != nullptr): the emit pass runs
    bool tier;
    bool charwise = false;  // the charwise engine (scan_kernel<CharEngine> / char_restart_kernel)
    bool restart = false;   // find_iter / leftmost_find_iter: the restart scanners (DARRAY tables)
    bool leftmost = false;
    uint32_t blocks, threads;
    ScanArgs a;
};

daac_status make_plan(const daac_pma *pma, const DeviceTables *t, int mode, int engine, uint64_t begin, uint64_t end, Plan &pl,
                      bool &heads) {
    daac_status kst = check_mode_kind(pma, mode);
    if (kst != DAAC_OK) return kst;
    pl.charwise = pma->charwise;
    if (pl.charwise && engine != DAAC_ENGINE_AUTO && engine != DAAC_ENGINE_DARRAY) {
        set_error("charwise automata run on their double array only (engine AUTO or DARRAY)");
        return DAAC_ERR_UNSUPPORTED;
    }
    pl.restart = mode == DAAC_FIND || mode == DAAC_LEFTMOST_FIND;
    pl.leftmost = mode == DAAC_LEFTMOST_FIND;
    if (pl.restart) {
        if (engine != DAAC_ENGINE_AUTO && engine != DAAC_ENGINE_DARRAY) {
            set_error("find_iter / leftmost_find_iter run on the DARRAY tables only");
            return DAAC_ERR_UNSUPPORTED;
        }
    }
    heads = mode == DAAC_FIND_OVERLAPPING_NO_SUFFIX;
    if (engine == DAAC_ENGINE_TIERED && !t->tier_ok) {
        set_error("TIERED engine not available for this automaton (more than 31 distinct pattern bytes, or not standard)");
        return DAAC_ERR_UNSUPPORTED;
    }
    if (engine == DAAC_ENGINE_GRAM) {
        set_error("the GRAM engine only serves daac_scan_count(DAAC_FIND_OVERLAPPING)");
        return DAAC_ERR_UNSUPPORTED;
    }
    if (engine != DAAC_ENGINE_AUTO && engine != DAAC_ENGINE_TIERED && engine != DAAC_ENGINE_DARRAY) {
        // (PFX counts; an unknown number is nobody's engine: a scan that silently ran on the double array instead would be ten times
        // slower than what the caller asked for, with daac_last_engine() saying DARRAY)
        set_error(engine == DAAC_ENGINE_PFX ? "the PFX engine only serves count (+ checksum) of DAAC_FIND_OVERLAPPING" : "unknown engine");
        return DAAC_ERR_UNSUPPORTED;
    }
    pl.tier = !pl.charwise && !pl.restart && (engine == DAAC_ENGINE_TIERED || (engine == DAAC_ENGINE_AUTO && t->tier_ok));
    uint32_t halo = pma->halo();
    // The sync-point scanners decide "is the classic state ROOT here" from a warm-up over the halo, and every lane
    // must reach the same verdict as a lane that has been following the text for longer: Lmax whole bytes, so that a
    // pattern of maximal length ending exactly at the cut is seen too.
    if (pl.restart) halo = std::max(halo, pma->max_pattern_len());
    uint32_t threads = static_cast<uint32_t>(OPT(threads));
    threads = std::min(1024u, std::max(64u, threads & ~63u));
    uint32_t bpc = static_cast<uint32_t>(OPT(blocks_per_cu));
    const uint32_t lds = pl.tier ? t->tier.lds_bytes : 4096u;
    if (bpc == 0) bpc = std::max(1u, std::min(2048u / threads, (160u * 1024u) / std::max(lds, 1u)));
    const uint64_t lanes = static_cast<uint64_t>(t->num_cu) * bpc * threads;
    const uint64_t len = end - begin;
    uint64_t S = static_cast<uint64_t>(OPT(seg_bytes));
    if (S == 0) {
        S = (len + lanes - 1) / lanes;
        const uint64_t min_seg = std::max<uint64_t>(pl.restart ? 1024 : 256, 16ull * halo);  // restart scans keep 56 B of chain state per segment
        S = std::max(S, min_seg);
    }
    S = (std::max<uint64_t>(S, 16) + 15) & ~15ull;
    const uint64_t nseg = len ? (len + S - 1) / S : 0;
    pl.threads = threads;
    pl.blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu) * bpc, (nseg + threads - 1) / threads)));
    pl.a = ScanArgs{};
    pl.a.begin = begin;
    pl.a.len = end;
    pl.a.seg_bytes = S;
    pl.a.nseg = nseg;
    pl.a.halo = halo;
    pl.a.total_len = end;
    if (pl.restart) {
        pl.threads = 256;
        pl.blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu) * static_cast<uint64_t>(std::max<int64_t>(1, OPT(restart_bpc))), (nseg + 255) / 256)));
    }
    return DAAC_OK;
}

hipError_t launch(const DeviceTables *t, const Plan &pl, int kmode, bool heads, hipStream_t s, unsigned long long *next_begin = nullptr) {
    if (pl.restart && pl.chain.x_prev != nullptr) {  // totals and per-segment counts are sums of tallies; only writing re-scans
        const int pass = kmode == 2 ? 2 : 3;
        return pl.charwise ? launch_char_chain(t->chr, pl.a, pl.chain, pass, kmode, pl.leftmost, next_begin, pl.blocks, s)
                           : launch_chain(t->da, pl.a, pl.chain, pass, kmode, pl.leftmost, next_begin, pl.blocks, s);
    }
    // count (+ checksum) of an overlapping scan the GRAM tables do not serve: the micro-step walker over segments (2048 lanes
    // per CU, a segment each) instead of the byte-at-a-time segment scanners
    if (kmode == 0 && !pl.restart && OPT(overlap_micro) != 0 && (pl.charwise || !pl.tier || OPT(overlap_micro) == 2) &&
        pl.a.seg_bytes + pl.a.halo < (1ull << 30)) {  // (the walker counts in 32-bit offsets from where it enters its segment)
        const uint32_t blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu) * 8, (pl.a.nseg + 255) / 256)));
        if (pl.charwise && t->chr.root_flag == 0) return launch_char_overlap_count(t->chr, pl.a, heads, blocks, s);
        if (!pl.charwise && t->da.root_flag == 0) return launch_overlap_count(t->da, pl.a, heads, blocks, s);
    }
    if (pl.charwise) {
        return pl.restart ? launch_char_restart_scan(t->chr, pl.a, kmode, pl.leftmost, next_begin, pl.blocks, pl.threads, s)
                          : launch_char_scan(t->chr, pl.a, kmode, heads, pl.blocks, pl.threads, s);
    }
    if (pl.restart) return launch_restart_scan(t->da, pl.a, kmode, pl.leftmost, next_begin, pl.blocks, pl.threads, s);
    return pl.tier ? launch_tier_scan(t->tier, pl.a, kmode, heads, pl.blocks, pl.threads, s)
                   : launch_darray_scan(t->da, pl.a, kmode, heads, pl.blocks, pl.threads, s);
}

// Resolves where the chain of a restart iterator enters every segment (chain_scan.hpp): speculative exits,
// then rounds of reconciliation until no exit moves.  On success pl.chain names the final exits and the
// emit passes may run; otherwise ("" in the pattern set, a link that would not end, no convergence) pl.chain
// stays empty and the sync-point scanners of restart_kernels.hip / charwise_kernels.hip do the scan.
struct ChainBuffers {
    void *buf = nullptr;
    hipStream_t s = nullptr;
    ~ChainBuffers() { dev_free(buf, s); }
};

// a few page-locked words per host thread for flags read back between passes
static unsigned int *pinned_words() {
    struct Holder {
        unsigned int *p = nullptr;
        Holder() { if (hipHostMalloc(reinterpret_cast<void **>(&p), 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } }
        ~Holder() { if (p) (void)hipHostFree(p); }
    };
    static thread_local Holder h;
    return h.p;
}

daac_status chain_resolve(const daac_pma *pma, const DeviceTables *t, Plan &pl, hipStream_t stream, ChainBuffers &cb) {
    pl.chain = ChainArgs{};
    if (!pl.restart || pma->root_has_output() || OPT(restart_chain) == 0 || pl.a.nseg == 0) return DAAC_OK;
    const uint64_t n = pl.a.nseg;
    cb.s = stream;
    HIP_TRY(dev_malloc(&cb.buf, (3 * n + 2) * sizeof(unsigned long long) + 2 * n * sizeof(uint4), stream));
    uint4 *tallies = static_cast<uint4 *>(cb.buf);  // 16-byte records first (alignment), then the exits
    unsigned long long *x_spec = reinterpret_cast<unsigned long long *>(tallies + 2 * n), *xa = x_spec + n, *xb = xa + n;
    unsigned int *flags = reinterpret_cast<unsigned int *>(xb + n);
    HIP_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(unsigned int), stream));
    ChainArgs c{};
    c.cap = std::max<uint64_t>(4096, 8 * pl.a.seg_bytes);
    c.flags = flags;
    c.tally_spec = tallies;
    c.tally_delta = tallies + n;
    c.x_out = x_spec;
    auto run = [&](int pass) {
        return pl.charwise ? launch_char_chain(t->chr, pl.a, c, pass, 0, pl.leftmost, nullptr, pl.blocks, stream)
                           : launch_chain(t->da, pl.a, c, pass, 0, pl.leftmost, nullptr, pl.blocks, stream);
    };
    HIP_TRY(run(0));
    const unsigned long long *prev = x_spec;
    const int max_rounds = static_cast<int>(std::max<int64_t>(1, OPT(chain_rounds)));
    for (int round = 0; round < max_rounds; ++round) {
        unsigned long long *out = (round & 1) ? xb : xa;
        c.x_spec = x_spec; c.x_prev = prev; c.x_out = out;
        HIP_TRY(hipMemsetAsync(flags, 0, sizeof(unsigned int), stream));
        HIP_TRY(run(1));
        unsigned int *f = pinned_words();  // page-locked: the copy is a plain DMA, not a staged one
        unsigned int f_local[2] = {0, 0};
        if (!f) f = f_local;
        HIP_TRY(hipMemcpyAsync(f, flags, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (f[1] != 0) return DAAC_OK;  // a link ran away: not this method's text
        prev = out;
        if (f[0] == 0) {                // nothing moved: `out` holds the true exits
            pl.chain = c;
            pl.chain.x_prev = out;
            pl.chain.x_out = nullptr;
            return DAAC_OK;
        }
    }
    return DAAC_OK;
}

// {count, S1, S2} of a shard scanned with shard-relative ends -> absolute ends, plus tuples counted on the host
// daac_match {start, end, value} -> {end u64, length u32, value u32}
// daac_match16 -> daac_match8 {value, (end - base) | length << end_bits}: what the compact lazy iterator sends over PCIe (half)
__global__ void repack8_kernel(const uint4 *in, uint2 *out, unsigned long long n, unsigned long long base, uint32_t end_bits) {
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
        const uint4 t = in[i];
        const unsigned long long end = (static_cast<unsigned long long>(t.y) << 32) | t.x;
        out[i] = uint2{t.w, static_cast<uint32_t>(end - base) | (t.z << end_bits)};
    }
}
__global__ void repack16_kernel(const daac_match *in, uint4 *out, unsigned long long n) {
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
        const daac_match m = in[i];
        out[i] = uint4{static_cast<uint32_t>(m.end), static_cast<uint32_t>(m.end >> 32), static_cast<uint32_t>(m.end - m.start), m.value};
    }
}

// r = {count, S1, S2} of [from, len) scanned as a haystack of its own, h = the same of [from, begin): what ends in (begin, len], ends re-based
__global__ void shard_subtract_kernel(unsigned long long *r, const unsigned long long *h, unsigned long long from32) {
    r[0] -= h[0];
    r[1] -= h[1];
    r[2] -= h[2];
    r[2] += (r[1] & 0xffffffffull) * from32;
}

__global__ void shard_fixup_kernel(unsigned long long *r, unsigned long long begin32, unsigned long long c, unsigned long long s1,
                                   unsigned long long s2) {
    r[2] += (r[1] & 0xffffffffull) * begin32 + s2;
    r[1] += s1;
    r[0] += c;
}

// Scans [begin, end) of a haystack whose byte 0 is at `dev_hay` (device pointer; only bytes
// >= begin - halo are dereferenced) and returns the matches with end in (begin, end] — plus
// ROOT's list at end = 0 when begin == 0 — in reference order.
// For the restart scanners (find_iter / leftmost_find_iter) `begin` must be a sync point (0, or the
// `next_begin` of the previous window), `total_len` is the real end of the haystack, and the scan runs
// on to the first sync point >= end, which is returned in *next_begin.
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

// FindOverlappingIterator of a bytewise Standard automaton through the one-detection tuple emitter (emit3_kernels.hip):
// DETECT (annotated class stream, tile counts, deep-match records) -> scans of the tile counts -> BIN (records by tile) -> EXPAND.
// *served = false when the automaton / request does not qualify or the haystack is of the adversarial kind the kernels give up on
// (then nothing is returned and the other engines take over).
// `dest`: the tuples go to this place (room for dest_cap of them) instead of a buffer of the call's own: out.n says how many, out.p stays null.
// `raw`: the PFX engine's tuples (any byte alphabet): pfx_emit_kernel logs every match of two or more bytes as a record, EXPAND runs over the
// haystack itself (one-byte patterns by table) — same glue, same record list, no annotated stream.
daac_status emit_overlapping3(daac_pma *pma, DeviceTables *t, const uint8_t *dev_hay, uint64_t begin, uint64_t end, hipStream_t stream,
                              DevMatches &out, bool *served, bool raw = false, void *dest = nullptr, uint64_t dest_cap = 0) {
    *served = false;
    if (!(raw ? t->pfx_emit_ok : t->emit3_ok) || OPT(emit) == 0 || end <= begin) return DAAC_OK;
    // (short scans may still try: they cost little; of the large ones every sixteenth looks again — one pair of adversarial haystacks
    // is not the text of a long-lived handle for ever)
    if (t->emit3_gave_up.load() >= 2 && end - begin >= (1u << 20) && (t->emit3_retry.fetch_add(1) & 15u) != 15u) return DAAC_OK;
    const Gram2EmitDev &e = raw ? t->pfx_emit : t->emit;
    const Gram3Lds &L = t->emit3_lds;
    const uint64_t halo = pma->halo();
    // windows of at most 1 GiB of end positions: virtual positions inside a window fit 32 bits
    const uint64_t kWin = 1ull << 30;
    constexpr uint32_t kStep = 2048;   // bytes of a DETECT wave-step
    struct Win { uint64_t wb, we, from; uint32_t lead, vlen, emit_from, nsteps, ntiles; uint64_t tile0, ann0; const uint8_t *hay_al; };
    std::vector<Win> wins;
    uint64_t tiles_total = 0, ann_total = 0;
    for (uint64_t wb = begin; wb < end; wb += kWin) {
        Win w{};
        w.wb = wb; w.we = std::min(end, wb + kWin);
        w.from = wb > halo ? wb - halo : 0;
        const uint8_t *first = dev_hay + w.from;
        w.lead = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(first) & 15u);
        w.hay_al = first - w.lead;
        const uint64_t vlen64 = w.lead + (w.we - w.from);
        if (vlen64 >= (1ull << 31)) return DAAC_OK;   // (a dictionary with a pattern of a GiB: not this engine's business)
        w.vlen = static_cast<uint32_t>(vlen64);
        w.emit_from = static_cast<uint32_t>(w.lead + (wb - w.from));
        w.nsteps = (w.vlen + kStep - 1) / kStep;
        w.ntiles = w.nsteps * (kStep / kEmit3Tile);
        w.tile0 = tiles_total;
        w.ann0 = ann_total;
        tiles_total += w.ntiles;
        if (!raw) ann_total += static_cast<uint64_t>(w.nsteps) * kStep;
        wins.push_back(w);
    }
    if (tiles_total >= (1ull << 32)) return DAAC_OK;
    // DETECT geometry (gram3's): regions of 64 KiB (256 KiB for the large windows), one 16- or 8-wave workgroup per CU
    uint32_t region = (end - begin) >= (1ull << 31) ? 262144u : 65536u;
    if (OPT(gram_region) >= 2048) { region = 2048; while (region * 2 <= static_cast<uint64_t>(OPT(gram_region)) && region < (1u << 20)) region *= 2; }
    const uint32_t wpb = raw ? t->pfx.threads / 64 : L.threads / 64;
    uint64_t max_regions = 0;
    for (const Win &w : wins) max_regions = std::max<uint64_t>(max_regions, (static_cast<uint64_t>(w.vlen) + region - 1) / region);
    const uint32_t blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu), (max_regions + wpb - 1) / wpb)));
    const uint64_t nwaves = static_cast<uint64_t>(blocks) * wpb;
    const uint32_t wq_slab = static_cast<uint32_t>(std::max<int64_t>(64 * 32 + 128 + 64, OPT(gram_slab)));
    const size_t wq_entry = raw ? sizeof(uint4) : sizeof(uint2);

    const size_t scan_words = tiles_total + 2 + exclusive_scan_scratch(tiles_total);
    const size_t off_short = 0, off_deep = off_short + ((tiles_total * 4 + 255) & ~size_t(255));
    const size_t off_a = off_deep + ((tiles_total * 4 + 255) & ~size_t(255)), off_b = off_a + ((scan_words * 8 + 255) & ~size_t(255));
    const size_t off_ctl = off_b + ((scan_words * 8 + 255) & ~size_t(255));   // {chunk_next, fail}
    const size_t off_wq = off_ctl + 256, off_ann = off_wq + ((nwaves * wq_slab * wq_entry + 255) & ~size_t(255));
    // the record list: sized for what the last scans of this automaton met (or the option's guess), rerun once with the exact number
    uint32_t per_kib = t->emit3_rec_per_kib.load();
    if (per_kib == 0) per_kib = static_cast<uint32_t>(std::max<int64_t>(1, OPT(emit_rec_per_kib)));
    uint64_t chunk_cap = ((end - begin) / 1024 + 1) * per_kib / kEmit3Chunk * 2 + 2 * nwaves + 16;
    const size_t g1_bytes = off_ann + ann_total + 256;
    Scratch sc(t, stream, g1_bytes + chunk_cap * (static_cast<size_t>(kEmit3Chunk) * sizeof(uint4) * 3 / 2 + 4) + 4096);   // (+ the binned copy: the list is at most half empty)
    void *g1_p = nullptr, *g_recs_p = nullptr, *g_bins_p = nullptr;
    HIP_TRY(sc.alloc(&g1_p, g1_bytes));
    char *base = static_cast<char *>(g1_p);
    uint32_t *d_short = reinterpret_cast<uint32_t *>(base + off_short), *d_deep = reinterpret_cast<uint32_t *>(base + off_deep);
    unsigned long long *d_a = reinterpret_cast<unsigned long long *>(base + off_a), *d_b = reinterpret_cast<unsigned long long *>(base + off_b);
    uint32_t *d_ctl = reinterpret_cast<uint32_t *>(base + off_ctl);
    uint8_t *d_ann = reinterpret_cast<uint8_t *>(base + off_ann);
    const size_t sc_mark = sc.mark();
    dbg_mark("emit: scratch");
    unsigned long long total = 0, deep_total = 0;
    uint32_t ctl[2] = {0, 0};
    for (int attempt = 0;; ++attempt) {
        if (chunk_cap >= (1ull << 32) / kEmit3Chunk) return DAAC_OK;
        sc.rewind(sc_mark);
        HIP_TRY(sc.alloc(&g_recs_p, chunk_cap * (static_cast<size_t>(kEmit3Chunk) * sizeof(uint4) + 4)));
        uint4 *d_recs = static_cast<uint4 *>(g_recs_p);
        uint32_t *d_fill = reinterpret_cast<uint32_t *>(d_recs + chunk_cap * kEmit3Chunk);
        HIP_TRY(hipMemsetAsync(d_fill, 0, chunk_cap * 4, stream));
        HIP_TRY(hipMemsetAsync(d_deep, 0, tiles_total * 4, stream));
        if (raw) HIP_TRY(hipMemsetAsync(d_short, 0, tiles_total * 4, stream));   // (its DETECT writes the tiles it meets one-byte patterns in)
        HIP_TRY(hipMemsetAsync(d_ctl, 0, 256, stream));
        for (const Win &w : wins) {
            Emit3Args a{};
            a.hay_al = w.hay_al; a.lead = w.lead; a.vlen = w.vlen; a.emit_from = w.emit_from;
            a.ann = d_ann + w.ann0;
            a.tile_short = d_short + w.tile0; a.tile_deep = d_deep + w.tile0; a.tile0 = static_cast<uint32_t>(w.tile0);
            a.recs = d_recs; a.chunk_fill = d_fill; a.chunk_next = d_ctl; a.chunk_cap = static_cast<uint32_t>(chunk_cap);
            a.wq = reinterpret_cast<uint2 *>(base + off_wq); a.wq_slab = wq_slab;
            a.region_bytes = region; a.nregions = static_cast<uint32_t>((static_cast<uint64_t>(w.vlen) + region - 1) / region);
            a.fail = d_ctl + 1;
            if (raw) HIP_TRY(launch_pfx_emit_detect(t->pfx, a, blocks, stream));
            else HIP_TRY(launch_emit3_detect(e, a, L, blocks, stream));
        }
        HIP_TRY(launch_emit3_combine(d_short, d_deep, d_a, d_b, tiles_total, stream));
        HIP_TRY(launch_exclusive_scan(d_a, tiles_total, d_a + tiles_total, d_a + tiles_total + 2, stream));
        HIP_TRY(launch_exclusive_scan(d_b, tiles_total, d_b + tiles_total, d_b + tiles_total + 2, stream));
        {
            unsigned long long *pin = reinterpret_cast<unsigned long long *>(pinned_words());
            HIP_TRY(hipMemcpyAsync(pin ? pin : &total, d_a + tiles_total, 8, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(pin ? pin + 1 : &deep_total, d_b + tiles_total, 8, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(pin ? reinterpret_cast<uint32_t *>(pin + 2) : ctl, d_ctl, 8, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (pin) { total = pin[0]; deep_total = pin[1]; std::memcpy(ctl, pin + 2, 8); }
        }
        if (ctl[1] != 0) {
            t->emit3_gave_up.fetch_add(1);
            set_error("GRAM emitter: a wave met more deep matches between two checkpoints than a chunk holds (code " + std::to_string(ctl[1]) + ")");
            return DAAC_OK;
        }
        if (ctl[0] <= chunk_cap) break;
        if (attempt != 0) { set_error("GRAM emitter: the record list overflowed twice"); return DAAC_OK; }
        chunk_cap = static_cast<uint64_t>(ctl[0]) + 2 * nwaves + 16;   // (chunks are closed at least half full: the rerun takes no more of them)
    }
    t->emit3_rec_per_kib.store(static_cast<uint32_t>(std::min<uint64_t>(1u << 20, deep_total * 5 / 4 / ((end - begin) / 1024 + 1) + 1)));
    g_last_engine = raw ? DAAC_ENGINE_PFX : DAAC_ENGINE_GRAM;
    const size_t tuple_bytes = out.f16 ? 16 : sizeof(daac_match);
    if (total == 0) { *served = true; return DAAC_OK; }
    if (!dest && total * tuple_bytes > static_cast<unsigned long long>(OPT(max_result_bytes))) {
        set_error("match list of " + std::to_string(total) + " tuples exceeds max_result_bytes; iterate with daac_iter_* instead");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
    if (dest && total > dest_cap) { set_error("tuple emitter: more tuples than the count pass announced"); return DAAC_ERR_DEVICE; }
    dbg_mark("emit: DETECT + scans read");
    HIP_TRY(sc.alloc(&g_bins_p, static_cast<size_t>(deep_total + 1) * sizeof(uint4)));
    if (deep_total != 0) {
        uint4 *d_recs = static_cast<uint4 *>(g_recs_p);
        const uint32_t *d_fill = reinterpret_cast<const uint32_t *>(d_recs + chunk_cap * kEmit3Chunk);
        HIP_TRY(launch_emit3_bin(d_recs, d_fill, d_ctl, static_cast<uint32_t>(chunk_cap), d_b, d_deep, static_cast<uint4 *>(g_bins_p), tiles_total, deep_total,
                                 static_cast<uint32_t>(std::min<uint64_t>(ctl[0], static_cast<uint64_t>(t->num_cu) * 16)), stream));
    }
    daac_match *d_out = static_cast<daac_match *>(dest);
    if (!dest) {
        HIP_TRY(dev_malloc(reinterpret_cast<void **>(&d_out), total * tuple_bytes, stream));
        out.p = d_out;
    }
    out.s = stream;
    out.n = total;
    dbg_mark("emit: BIN asked + out alloc");
    const char *dbg_env = std::getenv("DAAC_DEBUG_TIMING");
    const bool dbg_sync = dbg_env && dbg_env[0] == '1';
    if (dbg_sync) { fprintf(stderr, "[emit] BIN sync: %s  total=%llu deep=%llu chunks=%u cap=%llu\n", hipGetErrorString(hipStreamSynchronize(stream)), total, deep_total, ctl[0], (unsigned long long)chunk_cap); }
    HIP_TRY(hipMemsetAsync(d_ctl + 1, 0, 4, stream));
    for (const Win &w : wins) {
        Expand3Args a{};
        a.ann = d_ann + w.ann0;
        a.ntiles = (w.vlen + kEmit3Tile - 1) / kEmit3Tile;
        a.tile_off = d_a + w.tile0; a.bin_off = d_b + w.tile0;
        a.binned = static_cast<const uint4 *>(g_bins_p);
        a.out = d_out;
        a.pos_base = w.from - w.lead + 1;  // (mod 2^64: a match ends one past its last byte)
        a.has_len1 = (raw ? t->pfx.has_len1 != 0 : t->emit3_has_len1) ? 1u : 0u;
        if (raw) { a.ann = w.hay_al; a.vlen = w.vlen; a.emit_from = w.emit_from; }
        // the rank structure goes to LDS when the workgroups still fit with it: two of eight waves (16-byte tuples), three of four (24-byte)
        const uint32_t xwaves = (out.f16 && !raw) ? 8u : 4u;
        a.v3_in_lds = (!raw && e.v3c != nullptr && OPT(emit_v3_lds) != 0 &&
                       emit3_expand_lds_bytes(e, xwaves, out.f16, true) <= (160u * 1024u) / (out.f16 ? 2u : 3u)) ? 1u : 0u;
        a.off_wave = e.v1_bytes + e.v2_bytes + (a.v3_in_lds ? e.v3c_bytes : 0u);
        a.stagger = a.ntiles >= 65536u ? static_cast<uint32_t>(std::max<int64_t>(0, std::min<int64_t>(64, OPT(emit_stagger)))) : 0u;
        a.fail = d_ctl + 1;
        const uint32_t xblocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu) * ((out.f16 && !raw) ? 2u : 4u), (a.ntiles + xwaves - 1) / xwaves)));
        if (raw) HIP_TRY(launch_emit3_expand_raw(e, a, out.f16, xblocks, stream));
        else HIP_TRY(launch_emit3_expand(e, a, out.f16, xblocks, stream));
        if (dbg_sync) fprintf(stderr, "[emit] EXPAND window at %llu sync: %s\n", (unsigned long long)w.wb, hipGetErrorString(hipStreamSynchronize(stream)));
    }
    {
        unsigned int fail = 0;
        unsigned int *pin = pinned_words();
        HIP_TRY(hipMemcpyAsync(pin ? pin : &fail, d_ctl + 1, sizeof(fail), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (pin) fail = *pin;
        if (fail != 0) {  // more extras in one tile than EXPAND places: left to the other engines
            t->emit3_gave_up.fetch_add(1);
            set_error("GRAM emitter: the expansion gave up (code " + std::to_string(fail) + ")");
            dev_free(out.release(), stream);
            return DAAC_OK;
        }
    }
    out.f16_done = out.f16;
    t->emit3_gave_up.store(0);
    *served = true;
    dbg_mark("emit: EXPAND read");
    return DAAC_OK;
}

// FindIterator's count (+ checksum) over [begin, len) of a haystack that ends at `len`, without a state chain (find3_kernels.hip): DETECT and
// BIN of the tuple emitter, then SELECT passes over tiles of 2 048 positions until no tile's last word moves.  The result is left in
// d_res {count, S1, S2}.  *served = false: the dictionary / request does not qualify, or the text is of the kind the relaxation gives up
// on (then d_res holds nothing of value and the chain walkers take the request).
// What a window is asked for beside its sums: the tuples themselves
struct SelectEmit {
    bool f16 = false;
    void *dest = nullptr;      // in: write here (room for dest_cap tuples); null: a buffer of the call's own, handed back in p
    uint64_t dest_cap = 0;
    void *p = nullptr;         // out (dest == null): the list (dev_malloc on the call's stream)
    uint64_t n = 0;            // out: tuples written
};

// One window: matches with end in (begin, len], len - begin <= 1 GiB; begin is a restart point (0, or the end of a match the iterator
// returned).  r = {count, S1, S2} of the window; *next_begin = where a window behind this one restarts: the end of the last match
// selected here, or — none within the last two tiles — 64 bytes before the end (no match ends in between, and the longest pattern is
// shorter: the restart changes nothing).
// `leftmost`: the handle is a leftmost one and the selection is left3_kernels.hip's (by starts; begin = the first start that counts).
// There the window's matches START in [begin, sel_end) and the detection runs on to `len` (a match may end behind sel_end); *next_begin = the
// end of the window's last match, at least sel_end.
static daac_status find_count3_window(daac_pma *pma, DeviceTables *t, const uint8_t *dev_hay, uint64_t begin, uint64_t len, uint64_t sel_end, hipStream_t stream,
                                      bool want_checksum, bool leftmost, unsigned long long r[3], uint64_t *next_begin, bool *served,
                                      SelectEmit *em = nullptr) {
    *served = false;
    const int64_t optv = leftmost ? OPT(left3) : OPT(find3);
    // (DAAC_DEBUG_TIMING=1: the stream is waited for at every lap — kernel times; =2: host time between the laps as the call really runs)
    const char *dbg_env = std::getenv("DAAC_DEBUG_TIMING");
    const bool dbg_sync = dbg_env && dbg_env[0] == '1';
    auto lap = [&](const char *what) { if (dbg_env) { if (dbg_sync) (void)hipStreamSynchronize(stream); dbg_mark(what); } };
    if (!(leftmost ? t->left3_ok : t->find3_ok) || optv == 0 || len <= begin || len - begin > (1ull << 30)) return DAAC_OK;
    if (t->find3_gave_up.load() >= 2 && len - begin >= (1u << 20) && (t->find3_retry.fetch_add(1) & 15u) != 15u) return DAAC_OK;
    // (option find3 = 2: whatever the text)
    const uint32_t kDenseRecPerKib = 26;
    if (optv < 2 && t->find3_rec_per_kib.load() > kDenseRecPerKib + 1 && len - begin >= (1u << 20)) {
        // Text of dictionary words goes to the chain walkers without a detection.  Every sixteenth such request looks again — at a SAMPLE:
        // the first 4 MiB go through this function (detection, selection, result thrown away: ~20 us), which refreshes the handle's
        // records-per-KiB; the whole request is only detected when the sample says the text has changed.  (Round 4 ran the full
        // detection on those requests: 3.5 ms per GiB spent and discarded, profiles/r04_leftmost_dense_kernel_stats.csv.)
        static thread_local bool probing = false;
        if (probing || (t->find3_skips.fetch_add(1) & 15u) != 15u) return DAAC_OK;
        constexpr uint64_t kSample = 4ull << 20;
        if (len - begin > 2 * kSample) {
            unsigned long long sr[3];
            uint64_t snext = 0;
            bool sserved = false;
            probing = true;
            t->find3_rec_per_kib.store(0);   // (the sample itself must not be turned away by the gate)
            const daac_status sst = find_count3_window(pma, t, dev_hay, begin, begin + kSample, begin + kSample, stream, false, leftmost, sr, &snext, &sserved);
            probing = false;
            if (sst != DAAC_OK) return sst;
            if (t->find3_rec_per_kib.load() == 0) t->find3_rec_per_kib.store(kDenseRecPerKib + 2);   // (the sample gave no verdict: as before)
            if (t->find3_rec_per_kib.load() > kDenseRecPerKib + 1) return DAAC_OK;
        }
    }
    const Gram2EmitDev &e = t->emit;
    const Gram3Lds &L = t->emit3_lds;
    const uint64_t halo = pma->halo();
    constexpr uint32_t kStep = 2048;
    // (a restart point: nothing that begins before it is ever reported, so the detection begins THERE — the chunk-fed steppers hand over a buffer
    // whose first byte is the restart point, and reading a halo in front of it read in front of the allocation)
    const uint64_t from = begin;
    const uint8_t *first = dev_hay + from;
    const uint32_t lead = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(first) & 15u);
    const uint8_t *hay_al = first - lead;
    const uint64_t vlen64 = lead + (len - from);
    if (vlen64 >= (1ull << 31)) return DAAC_OK;
    const uint32_t vlen = static_cast<uint32_t>(vlen64), emit_from = static_cast<uint32_t>(lead + (begin - from));
    const uint32_t nsteps = (vlen + kStep - 1) / kStep, n1k = nsteps * (kStep / kEmit3Tile);
    const uint32_t region = 65536u;
    const uint32_t wpb = L.threads / 64;
    const uint64_t nregions = (static_cast<uint64_t>(vlen) + region - 1) / region;
    const uint32_t blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu), (nregions + wpb - 1) / wpb)));
    const uint64_t nwaves = static_cast<uint64_t>(blocks) * wpb;
    const uint32_t wq_slab = static_cast<uint32_t>(std::max<int64_t>(64 * 32 + 128 + 64, OPT(gram_slab)));
    const size_t scan_words = n1k + 2 + exclusive_scan_scratch(n1k);
    const size_t off_short = 0, off_deep = off_short + ((static_cast<size_t>(n1k) * 4 + 255) & ~size_t(255));
    const size_t off_a = off_deep + ((static_cast<size_t>(n1k) * 4 + 255) & ~size_t(255)), off_b = off_a + ((scan_words * 8 + 255) & ~size_t(255));
    const size_t off_ctl = off_b + ((scan_words * 8 + 255) & ~size_t(255));
    const size_t tcnt_words = em ? nsteps + 2 + exclusive_scan_scratch(nsteps) : 0;
    const size_t off_ex = off_ctl + 256, off_tcnt = off_ex + 2 * ((static_cast<size_t>(nsteps) * 4 + 255) & ~size_t(255));
    const size_t off_wq = off_tcnt + ((tcnt_words * 8 + 255) & ~size_t(255));
    const size_t off_ann = off_wq + ((nwaves * wq_slab * sizeof(uint2) + 255) & ~size_t(255));
    uint32_t per_kib = t->emit3_rec_per_kib.load();
    if (per_kib == 0) per_kib = static_cast<uint32_t>(std::max<int64_t>(1, OPT(emit_rec_per_kib)));
    uint64_t chunk_cap = ((len - begin) / 1024 + 1) * per_kib / kEmit3Chunk * 2 + 2 * nwaves + 16;
    const size_t g1_bytes = off_ann + static_cast<size_t>(nsteps) * kStep + 256;
    Scratch sc(t, stream, g1_bytes + chunk_cap * (static_cast<size_t>(kEmit3Chunk) * sizeof(uint4) * 2 + 4) + 4096);
    void *g1_p = nullptr, *g_recs_p = nullptr, *g_bins_p = nullptr;
    HIP_TRY(sc.alloc(&g1_p, g1_bytes));
    const size_t sc_mark = sc.mark();
    char *base = static_cast<char *>(g1_p);
    lap("scratch");
    uint32_t *d_short = reinterpret_cast<uint32_t *>(base + off_short), *d_deep = reinterpret_cast<uint32_t *>(base + off_deep);
    unsigned long long *d_a = reinterpret_cast<unsigned long long *>(base + off_a), *d_b = reinterpret_cast<unsigned long long *>(base + off_b);
    uint32_t *d_ctl = reinterpret_cast<uint32_t *>(base + off_ctl);
    uint32_t *d_ex[2] = {reinterpret_cast<uint32_t *>(base + off_ex), reinterpret_cast<uint32_t *>(base + off_ex + ((static_cast<size_t>(nsteps) * 4 + 255) & ~size_t(255)))};
    uint8_t *d_ann = reinterpret_cast<uint8_t *>(base + off_ann);
    // ---- DETECT (emit3_kernels.hip) with its record list, sized from what the handle's last scans met (rerun once if too short), and
    // behind it — without the host looking in between — BIN, the tiles' tails and the first SELECT pass; those do nothing when the list
    // overflowed or holds more than the chain walkers' text would (find3_detect_usable) ----
    const uint64_t kib = (len - begin) / 1024 + 1;
    const bool gate = optv < 2 && len - begin >= (1u << 20);
    const unsigned long long rec_gate = gate ? static_cast<unsigned long long>(kib) * (kDenseRecPerKib + 1) : ~0ull;
    Find3Args f{};
    f.ann = d_ann; f.ntiles = nsteps; f.n1k = n1k;
    f.bin_off = d_b;
    f.force_pos = emit_from == 0 ? 0xffffffffu : emit_from - 1u;
    f.pos_base = from - lead + 1;  // (mod 2^64: a match ends one past its last byte)
    f.result = reinterpret_cast<unsigned long long *>(d_ctl + 4);   // d_ctl: {chunks, DETECT's failure, flag, last selection + 1, - count, S1, S2 -}
    f.flag = d_ctl + 2;
    f.last_sel = d_ctl + 3;
    unsigned long long *d_tcnt = em ? reinterpret_cast<unsigned long long *>(base + off_tcnt) : nullptr;
    f.tile_cnt = d_tcnt;
    f.first_start = emit_from;
    f.last_start = (leftmost && sel_end < len) ? static_cast<uint32_t>(emit_from + (sel_end - begin)) : 0xffffffffu;
    f.ctl = d_ctl;
    f.count_only = want_checksum ? 0u : 1u;
    const uint32_t sblocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu), (nsteps + 15) / 16)));
    const uint32_t tblocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu) * 8, (nsteps + 63) / 64)));
    const uint32_t lds_tables = t->find3.h1_bytes + t->find3.h2_bytes + t->find3.h3c_bytes;
    unsigned long long deep_total = 0;
    uint32_t ctl[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int attempt = 0;; ++attempt) {
        if (chunk_cap >= (1ull << 32) / kEmit3Chunk) return DAAC_OK;
        sc.rewind(sc_mark);
        HIP_TRY(sc.alloc(&g_recs_p, chunk_cap * (static_cast<size_t>(kEmit3Chunk) * sizeof(uint4) + 4)));
        const unsigned long long rec_limit = std::min<unsigned long long>(rec_gate, chunk_cap * kEmit3Chunk);
        HIP_TRY(sc.alloc(&g_bins_p, static_cast<size_t>(rec_limit + 1) * sizeof(uint4)));
        uint4 *d_recs = static_cast<uint4 *>(g_recs_p);
        uint32_t *d_fill = reinterpret_cast<uint32_t *>(d_recs + chunk_cap * kEmit3Chunk);
        HIP_TRY(hipMemsetAsync(d_fill, 0, chunk_cap * 4, stream));
        HIP_TRY(hipMemsetAsync(d_deep, 0, static_cast<size_t>(n1k) * 4, stream));
        HIP_TRY(hipMemsetAsync(d_ctl, 0, 256, stream));
        Emit3Args a{};
        a.hay_al = hay_al; a.lead = lead; a.vlen = vlen; a.emit_from = emit_from;
        a.ann = d_ann;
        a.tile_short = d_short; a.tile_deep = d_deep; a.tile0 = 0;
        a.recs = d_recs; a.chunk_fill = d_fill; a.chunk_next = d_ctl; a.chunk_cap = static_cast<uint32_t>(chunk_cap);
        a.wq = reinterpret_cast<uint2 *>(base + off_wq); a.wq_slab = wq_slab;
        a.region_bytes = region; a.nregions = static_cast<uint32_t>(nregions);
        a.fail = d_ctl + 1;
        lap("allocs + memsets");
        HIP_TRY(launch_emit3_detect(e, a, L, blocks, stream));
        lap("DETECT");
        HIP_TRY(launch_emit3_combine(d_short, d_deep, d_a, d_b, n1k, stream));
        HIP_TRY(launch_exclusive_scan(d_b, n1k, d_b + n1k, d_b + n1k + 2, stream));
        HIP_TRY(launch_emit3_bin(d_recs, d_fill, d_ctl, static_cast<uint32_t>(chunk_cap), d_b, d_deep, static_cast<uint4 *>(g_bins_p), n1k, rec_limit,
                                 static_cast<uint32_t>(std::min<uint64_t>(chunk_cap, static_cast<uint64_t>(t->num_cu) * 16)), stream));
        lap("combine + scan + BIN");
        // SELECT: the tails' kernel leaves every tile's last word; a tallying pass enters with the words of the pass before
        f.binned = static_cast<const uint4 *>(g_bins_p);
        f.chunk_cap = static_cast<uint32_t>(chunk_cap); f.rec_limit = rec_limit;
        f.entry_in = nullptr; f.exit_out = d_ex[0]; f.off_wave = 0;
        if (leftmost) HIP_TRY(launch_left3_tail(f, t->emit3_has_len1, tblocks, stream));
        else HIP_TRY(launch_find3_tail(f, t->emit3_has_len1, tblocks, stream));
        f.off_wave = lds_tables;
        f.entry_in = d_ex[0]; f.exit_out = d_ex[1];
        if (leftmost) HIP_TRY(launch_left3_select(t->find3, f, t->emit3_has_len1, true, sblocks, stream));
        else HIP_TRY(launch_find3_select(t->find3, f, t->emit3_has_len1, true, sblocks, stream));
        {
            unsigned long long *pin = reinterpret_cast<unsigned long long *>(pinned_words());
            HIP_TRY(hipMemcpyAsync(pin ? pin : &deep_total, d_b + n1k, 8, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(pin ? reinterpret_cast<uint32_t *>(pin + 1) : ctl, d_ctl, 40, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (pin) { deep_total = pin[0]; std::memcpy(ctl, pin + 1, 40); }
        }
        lap("tails + SELECT + read");
        if (ctl[1] != 0) { t->find3_gave_up.fetch_add(1); return DAAC_OK; }
        if (ctl[0] <= chunk_cap) break;
        if (attempt != 0) return DAAC_OK;
        chunk_cap = static_cast<uint64_t>(ctl[0]) + 2 * nwaves + 16;
    }
    t->emit3_rec_per_kib.store(static_cast<uint32_t>(std::min<uint64_t>(1u << 20, deep_total * 5 / 4 / kib + 1)));
    {   // text that is mostly dictionary words: this detection has cost more than the chain walkers' whole scan — theirs from here on
        const uint64_t rk = deep_total / kib;
        t->find3_rec_per_kib.store(static_cast<uint32_t>(std::min<uint64_t>(1u << 20, rk + 1)));
        if (gate && rk > kDenseRecPerKib) return DAAC_OK;
    }
    if (deep_total > std::min<unsigned long long>(rec_gate, chunk_cap * kEmit3Chunk)) return DAAC_OK;   // (the kernels behind DETECT did nothing)
    const uint32_t *verified = d_ex[1];   // the exits of the last pass (= the entries it was given, once no tile's exit moved)
    for (int pass = 0;; ++pass) {
        const unsigned int flag = ctl[2];
        if (flag & 6u) { t->find3_gave_up.fetch_add(1); return DAAC_OK; }
        if ((flag & 1u) == 0) break;
        if (pass == 5) { t->find3_gave_up.fetch_add(1); return DAAC_OK; }   // (chains that will not fall in step: the walkers' business)
        HIP_TRY(hipMemsetAsync(d_ctl + 2, 0, 32, stream));   // flag, last selection, the three sums
        f.entry_in = d_ex[(pass & 1) ^ 1]; f.exit_out = d_ex[pass & 1];
        verified = f.exit_out;
        if (leftmost) HIP_TRY(launch_left3_select(t->find3, f, t->emit3_has_len1, true, sblocks, stream));
        else HIP_TRY(launch_find3_select(t->find3, f, t->emit3_has_len1, true, sblocks, stream));
        unsigned int *pin = pinned_words();
        HIP_TRY(hipMemcpyAsync(pin ? pin : &ctl[2], d_ctl + 2, 32, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (pin) std::memcpy(&ctl[2], pin, 32);
        lap("one more SELECT");
    }
    std::memcpy(r, &ctl[4], 24);
    if (em) {   // ---- the list: offsets = a scan over the tiles' counts, then the selection once more, writing ----
        const uint64_t n = r[0];
        const size_t tb = em->f16 ? 16 : sizeof(daac_match);
        em->n = 0; em->p = nullptr;
        void *dst = em->dest;
        if (dst) {
            if (n > em->dest_cap) { set_error("selection emitter: more tuples than the count pass announced"); return DAAC_ERR_DEVICE; }
        } else {
            if (n * tb > static_cast<unsigned long long>(OPT(max_result_bytes))) {
                set_error("match list of " + std::to_string(n) + " tuples exceeds max_result_bytes; iterate with daac_iter_* instead");
                return DAAC_ERR_AUTOMATON_SCALE;
            }
            if (n != 0) { HIP_TRY(dev_malloc(&em->p, n * tb, stream)); dst = em->p; }
        }
        if (n != 0) {
            HIP_TRY(launch_exclusive_scan(d_tcnt, nsteps, d_tcnt + nsteps, d_tcnt + nsteps + 2, stream));
            HIP_TRY(hipMemsetAsync(d_ctl + 2, 0, 4, stream));
            f.tile_cnt = nullptr; f.tile_off = d_tcnt; f.out = dst; f.f16 = em->f16 ? 1u : 0u;
            f.entry_in = verified; f.exit_out = const_cast<uint32_t *>(verified == d_ex[0] ? d_ex[1] : d_ex[0]);
            if (leftmost) HIP_TRY(launch_left3_emit(t->find3v, f, t->emit3_has_len1, sblocks, stream));
            else HIP_TRY(launch_find3_emit(t->find3v, f, t->emit3_has_len1, sblocks, stream));
            unsigned int flag = 0;
            unsigned int *pin = pinned_words();
            unsigned long long tot = 0;
            HIP_TRY(hipMemcpyAsync(pin ? pin : &flag, d_ctl + 2, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(pin ? reinterpret_cast<unsigned long long *>(pin + 2) : &tot, d_tcnt + nsteps, 8, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (pin) { flag = *pin; tot = *reinterpret_cast<unsigned long long *>(pin + 2); }
            lap("list");
            if (flag != 0 || tot != n) {   // (cannot happen after a verified tally; the walkers then)
                if (em->p) { dev_free(em->p, stream); em->p = nullptr; }
                t->find3_gave_up.fetch_add(1);
                return DAAC_OK;
            }
        }
        em->n = n;
    }
    if (leftmost) *next_begin = std::max<uint64_t>(sel_end, ctl[3] != 0 ? f.pos_base - 1u + ctl[3] : 0);
    else *next_begin = ctl[3] != 0 ? f.pos_base + (ctl[3] - 1u) : (len > 64 ? len - 64 : 0);
    *served = true;
    return DAAC_OK;
}

__global__ void set_result_kernel(unsigned long long *res, unsigned long long c, unsigned long long s1, unsigned long long s2) { res[0] = c; res[1] = s1; res[2] = s2; }

// The request in windows of 1 GiB of end positions, each restarting where the one before it selected its last match; the sums are left in
// d_res {count, S1, S2} (stream order) and in acc.
daac_status find_count3(daac_pma *pma, DeviceTables *t, const uint8_t *dev_hay, uint64_t begin, uint64_t len, hipStream_t stream,
                        unsigned long long *d_res, bool want_checksum, bool leftmost, unsigned long long acc[3], bool *served) {
    *served = false;
    acc[0] = acc[1] = acc[2] = 0;
    const uint64_t kWin = static_cast<uint64_t>(OPT(find3_window));
    for (uint64_t cur = begin;;) {
        // (leftmost: a window's matches START in it; the detection looks 32 bytes further so that the last ones are whole)
        const uint64_t wend = len - cur <= kWin ? len : cur + kWin - (leftmost ? 64 : 0);
        const uint64_t dend = leftmost ? std::min<uint64_t>(len, wend + 32) : wend;
        unsigned long long r[3] = {0, 0, 0};
        uint64_t next = wend;
        bool ok = false;
        const daac_status st = find_count3_window(pma, t, dev_hay, cur, dend, wend, stream, want_checksum, leftmost, r, &next, &ok);
        if (st != DAAC_OK || !ok) return st;
        for (int k = 0; k < 3; ++k) acc[k] += r[k];
        if (wend >= len || next >= len) break;
        if (next <= cur || next > dend) return DAAC_OK;   // (cannot happen; the walkers then)
        cur = next;
    }
    hipLaunchKernelGGL(set_result_kernel, dim3(1), dim3(1), 0, stream, d_res, acc[0], acc[1], acc[2]);
    HIP_TRY(hipGetLastError());
    t->find3_gave_up.store(0);
    g_last_engine = DAAC_ENGINE_GRAM;
    *served = true;
    return DAAC_OK;
}

// The restart iterators' tuple LIST from the selection kernels: scan_range_device's contract (begin = a restart point; the list holds the
// matches up to *next_begin, where the next window restarts).  One window of at most 1 GiB as it comes; a longer range only as a whole
// haystack (end == total_len): counted first, allocated once, then window by window straight into its place.
daac_status select_emit(daac_pma *pma, DeviceTables *t, int mode, const uint8_t *dev_hay, uint64_t begin, uint64_t end, uint64_t total_len,
                        hipStream_t stream, DevMatches &out, uint64_t *next_begin, bool *served) {
    *served = false;
    const bool leftmost = mode == DAAC_LEFTMOST_FIND;
    if (pma->charwise || pma->root_has_output() || end <= begin || pma->host.is_standard() == leftmost) return DAAC_OK;
    if (!(leftmost ? t->left3_ok : t->find3_ok) || OPT(select_emit) == 0) return DAAC_OK;
    const uint64_t kWin = static_cast<uint64_t>(OPT(find3_window));
    daac_status st;
    if (end - begin <= kWin) {
        SelectEmit em;
        em.f16 = out.f16;
        unsigned long long r[3];
        uint64_t next = end;
        bool ok = false;
        const uint64_t dend = leftmost ? std::min<uint64_t>(total_len, end + 32) : end;
        if ((st = find_count3_window(pma, t, dev_hay, begin, dend, end, stream, false, leftmost, r, &next, &ok, &em)) != DAAC_OK) return st;
        if (!ok) return DAAC_OK;
        if (end < total_len && next <= begin) { if (em.p) dev_free(em.p, stream); return DAAC_OK; }   // (a window without progress: the walkers')
        out.p = static_cast<daac_match *>(em.p); out.n = em.n; out.s = stream; out.f16_done = out.f16;
        if (next_begin) *next_begin = end >= total_len ? end : next;
        t->find3_gave_up.store(0);
        g_last_engine = DAAC_ENGINE_GRAM;
        *served = true;
        return DAAC_OK;
    }
    if (end != total_len) return DAAC_OK;
    unsigned long long acc[3];
    bool counted = false;
    DevBuf tmp;
    HIP_TRY(tmp.alloc(3 * sizeof(unsigned long long), stream));
    if ((st = find_count3(pma, t, dev_hay, begin, end, stream, static_cast<unsigned long long *>(tmp.p), false, leftmost, acc, &counted)) != DAAC_OK) return st;
    if (!counted) return DAAC_OK;
    const uint64_t total = acc[0];
    const size_t tb = out.f16 ? 16 : sizeof(daac_match);
    if (total * tb > static_cast<unsigned long long>(OPT(max_result_bytes))) {
        set_error("match list of " + std::to_string(total) + " tuples exceeds max_result_bytes; iterate with daac_iter_* instead");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
    void *d_out = nullptr;
    HIP_TRY(dev_malloc(&d_out, std::max<size_t>(16, total * tb), stream));
    uint64_t at = 0;
    for (uint64_t cur = begin;;) {
        const uint64_t wend = end - cur <= kWin ? end : cur + kWin - (leftmost ? 64 : 0);
        const uint64_t dend = leftmost ? std::min<uint64_t>(end, wend + 32) : wend;
        SelectEmit em;
        em.f16 = out.f16; em.dest = static_cast<char *>(d_out) + at * tb; em.dest_cap = total - at;
        unsigned long long r[3];
        uint64_t next = wend;
        bool ok = false;
        st = find_count3_window(pma, t, dev_hay, cur, dend, wend, stream, false, leftmost, r, &next, &ok, &em);
        if (st != DAAC_OK || !ok) { dev_free(d_out, stream); return st; }
        at += em.n;
        if (wend >= end || next >= end) break;
        if (next <= cur || next > dend) { dev_free(d_out, stream); return DAAC_OK; }
        cur = next;
    }
    if (at != total) { dev_free(d_out, stream); return DAAC_OK; }
    out.p = static_cast<daac_match *>(d_out); out.n = total; out.s = stream; out.f16_done = out.f16;
    if (next_begin) *next_begin = end;
    t->find3_gave_up.store(0);
    g_last_engine = DAAC_ENGINE_GRAM;
    *served = true;
    return DAAC_OK;
}

// Scans [begin, end) of a haystack whose byte 0 is at `dev_hay` (device pointer; only bytes
// >= begin - halo are dereferenced) and leaves the matches with end in (begin, end] — plus
// ROOT's list at end = 0 when begin == 0 — in device memory, in reference order.
// For the restart scanners (find_iter / leftmost_find_iter) `begin` must be a sync point (0, or the
// `next_begin` of the previous window), `total_len` is the real end of the haystack, and the scan runs
// on to the first sync point >= end, which is returned in *next_begin.
daac_status scan_range_device(daac_pma *pma, DeviceTables *t, int mode, int engine, const uint8_t *dev_hay, uint64_t begin,
                              uint64_t end, uint64_t total_len, hipStream_t stream, DevMatches &out, uint64_t *next_begin) {
    Plan pl;
    bool heads = false;
    const bool want_gram = engine == DAAC_ENGINE_GRAM, want_pfx = engine == DAAC_ENGINE_PFX;
    daac_status st = make_plan(pma, t, mode, (want_gram || want_pfx) ? DAAC_ENGINE_AUTO : engine, begin, end, pl, heads);
    if (st != DAAC_OK) return st;
    if (next_begin) *next_begin = end;
    // (AUTO: the GRAM tables' emitter where the dictionary has them — option pfx = 2 builds both —, then PFX's)
    if (!pma->charwise && mode == DAAC_FIND_OVERLAPPING && (engine == DAAC_ENGINE_AUTO || want_gram)) {
        bool served = false;
        if ((st = emit_overlapping3(pma, t, dev_hay, begin, end, stream, out, &served)) != DAAC_OK) return st;
        if (served) return DAAC_OK;
    }
    if (engine == DAAC_ENGINE_AUTO && (mode == DAAC_FIND || mode == DAAC_LEFTMOST_FIND)) {   // the restart iterators' list from the selection kernels
        bool served = false;
        if ((st = select_emit(pma, t, mode, dev_hay, begin, end, total_len, stream, out, next_begin, &served)) != DAAC_OK) return st;
        if (served) return DAAC_OK;
        if (next_begin) *next_begin = end;
    }
    if (want_gram) {
        set_error(std::string("the GRAM engine cannot emit tuples for this automaton / request [") + last_error_cstr() + "]");
        return DAAC_ERR_UNSUPPORTED;
    }
    if (!pma->charwise && mode == DAAC_FIND_OVERLAPPING && (engine == DAAC_ENGINE_AUTO || want_pfx) && t->pfx_emit_ok) {
        bool served = false;
        if (end <= begin && want_pfx) { g_last_engine = DAAC_ENGINE_PFX; return DAAC_OK; }   // (no "" among the patterns: nothing ends at 0)
        // Every tuple of this path is first a record, then a binned record, then a tuple: three and a half times the list's size in flight
        // (1 GiB of the Unidic-like text in one piece asked the driver for 30 GB per call and took 1.2 s for it).  A range whose scratch
        // would pass 8 GB — by what the handle's last scan met; a first scan beyond 256 MiB counts as such — is therefore COUNTED first
        // (`.count()`, one pass), the list allocated once, and emitted piece by piece straight into its place.
        const uint64_t kPiece = 256ull << 20;
        const uint64_t hint = t->emit3_rec_per_kib.load();
        const uint64_t est = hint ? (end - begin) / 1024 * hint * 58 : ~0ull;   // 16 B per record x 3.6
        if (end - begin <= kPiece || est <= (8ull << 30)) {
            if ((st = emit_overlapping3(pma, t, dev_hay, begin, end, stream, out, &served, true)) != DAAC_OK) return st;
            if (served) return DAAC_OK;
        } else {
            uint64_t total = 0;
            if ((st = scan_count_impl(pma, DAAC_FIND_OVERLAPPING, DAAC_ENGINE_AUTO, dev_hay, end, begin, 1, stream, &total, nullptr, nullptr, false)) != DAAC_OK) return st;
            const size_t tb = out.f16 ? 16 : sizeof(daac_match);
            if (total * tb > static_cast<unsigned long long>(OPT(max_result_bytes))) {
                set_error("match list of " + std::to_string(total) + " tuples exceeds max_result_bytes; iterate with daac_iter_* instead");
                return DAAC_ERR_AUTOMATON_SCALE;
            }
            daac_match *d_out = nullptr;
            HIP_TRY(dev_malloc(reinterpret_cast<void **>(&d_out), std::max<size_t>(16, total * tb), stream));
            uint64_t at = 0;
            served = true;
            for (uint64_t b = begin; b < end && served; b += kPiece) {
                DevMatches part;
                part.f16 = out.f16;
                if ((st = emit_overlapping3(pma, t, dev_hay, b, std::min(end, b + kPiece), stream, part, &served, true,
                                            reinterpret_cast<char *>(d_out) + at * tb, total - at)) != DAAC_OK) { dev_free(d_out, stream); return st; }
                at += part.n;
            }
            if (served && at == total) {
                out.p = d_out; out.s = stream; out.n = total; out.f16_done = out.f16;
                g_last_engine = DAAC_ENGINE_PFX;
                return DAAC_OK;
            }
            dev_free(d_out, stream);   // (a piece was given up: the other engines take the whole range)
            served = false;
        }
    }
    if (want_pfx) {
        set_error(std::string("the PFX engine cannot emit tuples for this automaton / request [") + last_error_cstr() + "]");
        return DAAC_ERR_UNSUPPORTED;
    }
    g_last_engine = pl.tier ? DAAC_ENGINE_TIERED : DAAC_ENGINE_DARRAY;
    // an empty range still has to report ROOT's list at end = 0: run one (empty) segment
    if (pl.a.nseg == 0) { if (begin != 0) return DAAC_OK; pl.a.nseg = 1; }
    pl.a.hay = dev_hay;
    pl.a.total_len = total_len;
    DevBuf g1;
    HIP_TRY(g1.alloc((pl.a.nseg + 3 + exclusive_scan_scratch(pl.a.nseg)) * sizeof(unsigned long long), stream));
    unsigned long long *d_counts = static_cast<unsigned long long *>(g1.p);
    pl.a.seg_counts = d_counts;
    pl.a.result = d_counts + pl.a.nseg;
    unsigned long long *d_next = d_counts + pl.a.nseg + 1;
    HIP_TRY(hipMemsetAsync(d_next, 0, 2 * sizeof(unsigned long long), stream));
    pl.a.flags = d_next + 1;
    ChainBuffers chain_buffers;
    if ((st = chain_resolve(pma, t, pl, stream, chain_buffers)) != DAAC_OK) return st;
    HIP_TRY(launch(t, pl, 1, heads, stream, d_next));
    HIP_TRY(launch_exclusive_scan(d_counts, pl.a.nseg, d_counts + pl.a.nseg, d_counts + pl.a.nseg + 3, stream));
    unsigned long long total = 0, nbf[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&total, d_counts + pl.a.nseg, sizeof(total), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(nbf, d_next, sizeof(nbf), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const unsigned long long nb = nbf[0];
    if (nbf[1] & 1ull) return diverged();
    // FindIterator with "" in the set reports every position whatever the text: windows need no sync point
    const bool positional = pl.restart && !pl.leftmost && pma->root_has_output();
    if (pl.restart && !positional && next_begin) *next_begin = std::max<uint64_t>(nb, end);
    if (total == 0) return DAAC_OK;
    if (total * sizeof(daac_match) > static_cast<unsigned long long>(OPT(max_result_bytes))) {
        set_error("match list of " + std::to_string(total) + " tuples exceeds max_result_bytes; iterate with daac_iter_* instead");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
    daac_match *d_out = nullptr;
    HIP_TRY(dev_malloc(reinterpret_cast<void **>(&d_out), total * sizeof(daac_match), stream));
    out.p = d_out;
    out.s = stream;
    out.n = total;
    pl.a.out = d_out;
    HIP_TRY(launch(t, pl, 2, heads, stream, nullptr));
    return DAAC_OK;
}

// The same, copied to the host (page-locked) for daac_scan / the lazy iterator / the steppers.
daac_status scan_range_materialize(daac_pma *pma, DeviceTables *t, int mode, int engine, const uint8_t *dev_hay, uint64_t begin,
                                   uint64_t end, uint64_t total_len, hipStream_t stream, MatchBuf &out,
                                   uint64_t *next_begin) {
    out.clear();
    DevMatches dm;
    const daac_status st = scan_range_device(pma, t, mode, engine, dev_hay, begin, end, total_len, stream, dm, next_begin);
    if (st != DAAC_OK) return st;
    if (dm.n == 0) return DAAC_OK;
    if (!out.reserve(dm.n)) { set_error("out of host memory for the match list"); return DAAC_ERR_AUTOMATON_SCALE; }
    out.n = dm.n;
    HIP_TRY(hipMemcpyAsync(out.p, dm.p, dm.n * sizeof(daac_match), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return DAAC_OK;
}

// Host haystack window -> device buffer holding bytes [copy_from, end); returns the pointer that
// byte 0 of the haystack would have.
daac_status stage_window(const uint8_t *host_hay, uint64_t copy_from, uint64_t end, hipStream_t stream, void **dbuf,
                         const uint8_t **virt_base) {
    const uint64_t n = end - copy_from;
    const uint64_t skew = copy_from & 15;  // keep the haystack's 16-byte phase for the vector loop
    HIP_TRY(hipMalloc(dbuf, n + skew + 32));
    if (n) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(*dbuf) + skew, host_hay + copy_from, n, hipMemcpyHostToDevice, stream));
    *virt_base = static_cast<const uint8_t *>(*dbuf) + skew - copy_from;
    return DAAC_OK;
}

}  // namespace

// ============================================================================== exported C ABI
extern "C" {

const char *daac_last_error(void) { return last_error_cstr(); }
int daac_last_engine(void) { return g_last_engine; }
void daac_free(void *p) { std::free(p); }

daac_status daac_bytewise_from_serialized(const uint8_t *blob, size_t len, daac_pma **out, size_t *consumed) {
    if (!blob || !out) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::unique_ptr<daac_pma> p(new daac_pma);
    const daac_status st = HostPma::deserialize(blob, len, p->host, consumed);
    if (st != DAAC_OK) return st;
    *out = p.release();
    return DAAC_OK;
}

daac_status daac_bytewise_from_parts(const uint32_t *states, size_t n_states, const uint32_t *lstates, const uint32_t *fails,
                                     size_t n_lstates, const uint32_t *outputs, size_t n_outputs, uint8_t match_kind,
                                     uint32_t num_states, daac_pma **out) {
    if (!out || match_kind > 2) { set_error("bad argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    if ((n_states && !states) || (n_lstates && (!lstates || !fails)) || (n_outputs && !outputs)) {
        set_error("null array with a non-zero count (`fails` must hold n_lstates entries)");
        return DAAC_ERR_INVALID_ARGUMENT;
    }
    std::unique_ptr<daac_pma> p(new daac_pma);
    HostPma &h = p->host;
    h.match_kind = match_kind;
    h.num_states = num_states;
    h.states.resize(n_states);
    if (n_states) std::memcpy(h.states.data(), states, n_states * sizeof(StateRec));
    h.lstates.resize(n_lstates);
    h.fails.resize(n_lstates);
    if (n_lstates) {
        std::memcpy(h.lstates.data(), lstates, n_lstates * sizeof(LStateRec));
        std::memcpy(h.fails.data(), fails, n_lstates * sizeof(uint32_t));
    }
    h.outputs.resize(n_outputs);
    if (n_outputs) std::memcpy(h.outputs.data(), outputs, n_outputs * sizeof(OutputRec));
    if (h.is_standard()) h.build_root_table();
    const daac_status st = h.validate();
    if (st != DAAC_OK) return st;
    *out = p.release();
    return DAAC_OK;
}

daac_status daac_bytewise_build(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n, uint8_t match_kind,
                                uint32_t num_free_blocks, daac_pma **out) {
    if (!out || (n && (!blob || !offsets))) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    try {
        std::unique_ptr<daac_pma> p(new daac_pma);
        const daac_status st = build_bytewise(blob, offsets, values, n, match_kind, num_free_blocks, p->host);
        if (st != DAAC_OK) return st;
        *out = p.release();
        return DAAC_OK;
    } catch (const std::bad_alloc &) {
        set_error("out of memory while building the automaton");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
}

daac_status daac_charwise_from_serialized(const uint8_t *blob, size_t len, daac_pma **out, size_t *consumed) {
    if (!blob || !out) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::unique_ptr<daac_pma> p(new daac_pma);
    p->charwise = true;
    const daac_status st = HostCharPma::deserialize(blob, len, p->chost, consumed);
    if (st != DAAC_OK) return st;
    *out = p.release();
    return DAAC_OK;
}

daac_status daac_charwise_build(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n, uint8_t match_kind,
                                uint32_t num_free_blocks, daac_pma **out) {
    if (!out || (n && (!blob || !offsets))) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    try {
        std::unique_ptr<daac_pma> p(new daac_pma);
        p->charwise = true;
        const daac_status st = build_charwise(blob, offsets, values, n, match_kind, num_free_blocks, p->chost);
        if (st != DAAC_OK) return st;
        *out = p.release();
        return DAAC_OK;
    } catch (const std::bad_alloc &) {
        set_error("out of memory while building the automaton");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
}

daac_status daac_pma_serialize(const daac_pma *pma, uint8_t **buf, size_t *len) {
    if (!pma || !buf || !len) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::vector<uint8_t> v;
    if (pma->charwise) pma->chost.serialize(v); else pma->host.serialize(v);
    uint8_t *b = static_cast<uint8_t *>(std::malloc(v.size() ? v.size() : 1));
    if (!b) { set_error("out of memory"); return DAAC_ERR_AUTOMATON_SCALE; }
    std::memcpy(b, v.data(), v.size());
    *buf = b;
    *len = v.size();
    return DAAC_OK;
}

uint32_t daac_abi_version(void) { return DAAC_ABI_VERSION; }

// the engine plan of a handle: what scan_count_impl / scan_range_device / make_plan decide for engine AUTO, said up front
static void fill_plan(const daac_pma *pma, const DeviceTables *t, daac_info &f) {
    auto set = [&](int req, int engine, int kernel, int why) {
        f.plan_engine[req] = static_cast<uint8_t>(engine); f.plan_kernel[req] = static_cast<uint8_t>(kernel); f.plan_reason[req] = static_cast<uint8_t>(why);
    };
    for (int r = 0; r < DAAC_REQ_N; ++r) set(r, DAAC_ENGINE_AUTO, DAAC_KERNEL_NONE, t ? DAAC_WHY_FASTEST : DAAC_WHY_NOT_UPLOADED);
    if (!t) return;
    if (pma->charwise) {
        const bool standard = pma->chost.match_kind == DAAC_STANDARD;
        if (standard) {
            set(DAAC_REQ_OVERLAPPING_COUNT, DAAC_ENGINE_DARRAY, DAAC_KERNEL_MICRO, DAAC_WHY_CHARWISE);
            set(DAAC_REQ_OVERLAPPING_CHECKSUM, DAAC_ENGINE_DARRAY, DAAC_KERNEL_MICRO, DAAC_WHY_CHARWISE);
            set(DAAC_REQ_OVERLAPPING_TUPLES, DAAC_ENGINE_DARRAY, DAAC_KERNEL_SEGMENT, DAAC_WHY_CHARWISE);
            set(DAAC_REQ_NO_SUFFIX, DAAC_ENGINE_DARRAY, DAAC_KERNEL_SEGMENT, DAAC_WHY_CHARWISE);
            set(DAAC_REQ_FIND, DAAC_ENGINE_DARRAY, DAAC_KERNEL_CHAIN, DAAC_WHY_CHAIN);
        } else {
            set(DAAC_REQ_LEFTMOST_FIND, DAAC_ENGINE_DARRAY, DAAC_KERNEL_CHAIN, DAAC_WHY_CHAIN);
        }
        return;
    }
    const HostPma &h = pma->host;
    // find3 / left3 serve the restart iterators' count while the handle's last such request did not meet text made of dictionary words
    const bool select_text_ok = t->find3_gave_up.load() < 2 && t->find3_rec_per_kib.load() <= 27;
    if (!h.is_standard()) {
        if (t->left3_ok && OPT(left3) != 0 && !pma->root_has_output() && (select_text_ok || OPT(left3) >= 2))
            set(DAAC_REQ_LEFTMOST_FIND, DAAC_ENGINE_GRAM, DAAC_KERNEL_SELECT, DAAC_WHY_FASTEST);
        else set(DAAC_REQ_LEFTMOST_FIND, DAAC_ENGINE_DARRAY, pma->root_has_output() ? DAAC_KERNEL_SEGMENT : DAAC_KERNEL_CHAIN, DAAC_WHY_CHAIN);
        return;
    }
    // why the byte-class tables were declined, as far as it is known
    int why_no_gram = DAAC_WHY_TRIE_SHAPE;
    if (pma->root_has_output()) why_no_gram = DAAC_WHY_EMPTY_PATTERN;
    else if (t->n_distinct_bytes > 61) why_no_gram = DAAC_WHY_ALPHABET;
    else if (t->n_distinct_bytes != 0) why_no_gram = DAAC_WHY_LDS;
    const int seg_engine = t->tier_ok ? DAAC_ENGINE_TIERED : DAAC_ENGINE_DARRAY;
    const int micro = OPT(overlap_micro) >= (t->tier_ok ? 2 : 1) ? DAAC_KERNEL_MICRO : DAAC_KERNEL_SEGMENT;
    const int micro_engine = micro == DAAC_KERNEL_MICRO ? DAAC_ENGINE_DARRAY : seg_engine;
    const int64_t gv = OPT(gram_version);
    if (t->gram2_ok && gv != 1) set(DAAC_REQ_OVERLAPPING_COUNT, DAAC_ENGINE_GRAM, DAAC_KERNEL_GRAM_COUNT, DAAC_WHY_FASTEST);
    else if (t->gram_ok) set(DAAC_REQ_OVERLAPPING_COUNT, DAAC_ENGINE_GRAM, DAAC_KERNEL_GRAM_EXACT, DAAC_WHY_FASTEST);
    else if (t->gramw_ok) set(DAAC_REQ_OVERLAPPING_COUNT, DAAC_ENGINE_GRAM, DAAC_KERNEL_GRAM_WIDE, DAAC_WHY_FASTEST);
    else if (t->pfx_ok) set(DAAC_REQ_OVERLAPPING_COUNT, DAAC_ENGINE_PFX, DAAC_KERNEL_PFX, why_no_gram);
    else set(DAAC_REQ_OVERLAPPING_COUNT, micro_engine, micro, why_no_gram);
    if (t->gram_ok || (t->gram2_ok && t->gram2.exact_ok)) set(DAAC_REQ_OVERLAPPING_CHECKSUM, DAAC_ENGINE_GRAM, DAAC_KERNEL_GRAM_EXACT, DAAC_WHY_FASTEST);
    else if (t->gramw_ok && t->gramw.exact_ok) set(DAAC_REQ_OVERLAPPING_CHECKSUM, DAAC_ENGINE_GRAM, DAAC_KERNEL_GRAM_WIDE, DAAC_WHY_FASTEST);
    else if (t->pfx_ok) set(DAAC_REQ_OVERLAPPING_CHECKSUM, DAAC_ENGINE_PFX, DAAC_KERNEL_PFX, (t->gram2_ok || t->gramw_ok) ? DAAC_WHY_LDS : why_no_gram);  // (as scan_count_impl: PFX wherever no GRAM table set serves the request)
    else set(DAAC_REQ_OVERLAPPING_CHECKSUM, micro_engine, micro, (t->gram2_ok || t->gramw_ok) ? DAAC_WHY_LDS : why_no_gram);
    if (t->emit3_ok && t->emit3_gave_up.load() < 2 && OPT(emit) != 0)
        set(DAAC_REQ_OVERLAPPING_TUPLES, DAAC_ENGINE_GRAM, DAAC_KERNEL_GRAM_EMIT, DAAC_WHY_FASTEST);
    else if (t->pfx_emit_ok && t->emit3_gave_up.load() < 2 && OPT(emit) != 0)
        set(DAAC_REQ_OVERLAPPING_TUPLES, DAAC_ENGINE_PFX, DAAC_KERNEL_PFX, why_no_gram);
    else set(DAAC_REQ_OVERLAPPING_TUPLES, seg_engine, DAAC_KERNEL_SEGMENT, t->gram2_ok ? DAAC_WHY_DUPLICATES : why_no_gram);
    set(DAAC_REQ_NO_SUFFIX, seg_engine, DAAC_KERNEL_SEGMENT, DAAC_WHY_FASTEST);
    if (t->find3_ok && OPT(find3) != 0 && !pma->root_has_output() && (select_text_ok || OPT(find3) >= 2))
        set(DAAC_REQ_FIND, DAAC_ENGINE_GRAM, DAAC_KERNEL_SELECT, DAAC_WHY_FASTEST);
    else set(DAAC_REQ_FIND, DAAC_ENGINE_DARRAY, pma->root_has_output() ? DAAC_KERNEL_SEGMENT : DAAC_KERNEL_CHAIN, DAAC_WHY_CHAIN);  // (the restart iterators run on the double array)
}

daac_status daac_pma_info(const daac_pma *pma, daac_info *info) {
    PmaScope scope_(pma);
    if (!pma || !info) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    const uint32_t cap = info->struct_size;
    if (cap < 8 || cap > (1u << 16)) {
        set_error("daac_info.struct_size must hold sizeof(daac_info) of the caller (ABI version " + std::to_string(DAAC_ABI_VERSION) + ")");
        return DAAC_ERR_INVALID_ARGUMENT;
    }
    daac_info full;
    daac_info *f = &full;
    std::memset(f, 0, sizeof(*f));
    f->struct_size = static_cast<uint32_t>(sizeof(daac_info));
    if (pma->charwise) {
        const HostCharPma &c = pma->chost;
        f->match_kind = c.match_kind;
        f->num_states = c.num_states;
        f->states_len = c.states.size();
        f->outputs_len = c.outputs.size();
        f->heap_bytes = c.heap_bytes();
        f->max_pattern_len = c.max_pattern_len();
        f->charwise = 1;
        f->alphabet_size = c.alphabet_size;
    } else {
        const HostPma &h = pma->host;
        f->match_kind = h.match_kind;
        f->num_states = h.num_states;
        f->states_len = h.states_len();
        f->outputs_len = h.outputs.size();
        f->heap_bytes = h.heap_bytes();
        f->max_pattern_len = h.max_pattern_len();
    }
    {
        std::lock_guard<std::mutex> g(const_cast<daac_pma *>(pma)->mu);
        const DeviceTables *t = pma->dev.empty() ? nullptr : pma->dev.begin()->second.get();
        if (t && !pma->charwise) {
            f->tiered_available = t->tier_ok;
            if (t->tier_ok) {
                f->num_classes = t->tier.C;
                f->tier_dense_states = t->tier.NA;
                f->tier_lds_states = t->tier.NB;
                f->tier_lds_bytes = t->tier.lds_bytes;
            }
            f->gram_available = t->gram_ok || t->gram2_ok;
            if (t->gram_ok) {
                f->gram_k = t->gram.K;
                f->gram_lds_bytes = t->gram.lds_bytes;
            }
            if (t->gramw_ok) { f->gram_available = 1; f->gram_k = 2; f->gram_lds_bytes = t->gramw.lds_count; f->num_classes = t->gramw.C; f->gram_wide = 1; }
            f->gram2_available = t->gram2_ok;
            if (t->gram2_ok) {
                f->gram2_k = t->gram2.K;
                f->gram2_exact = t->gram2.exact_ok;
                f->gram2_lds_count = t->gram2.lds_count;
                f->gram2_lds_exact = t->gram2.lds_exact;
                if (!t->gram_ok) { f->gram_k = t->gram2.K; f->gram_lds_bytes = t->gram2.lds_count; }
            }
            f->pfx_available = t->pfx_ok;
            if (t->pfx_ok) { f->pfx_key_bytes = t->pfx.G; f->pfx_lds_bytes = t->pfx.lds_bytes; }
        }
        fill_plan(pma, t, *f);
    }
    std::memcpy(info, f, std::min<size_t>(cap, sizeof(daac_info)));
    info->struct_size = static_cast<uint32_t>(std::min<size_t>(cap, sizeof(daac_info)));
    return DAAC_OK;
}

size_t daac_pma_explain(const daac_pma *pma, char *buf, size_t cap) {
    if (!pma) return 0;
    daac_info f;
    f.struct_size = static_cast<uint32_t>(sizeof(f));
    if (daac_pma_info(pma, &f) != DAAC_OK) return 0;
    static const char *req[] = {"find_overlapping_iter(h).count()", "find_overlapping count + checksum", "find_overlapping tuples", "find_iter",
                                "leftmost_find_iter", "find_overlapping_no_suffix_iter"};
    static const char *eng[] = {"auto", "tiered", "darray", "gram", "pfx"};
    static const char *ker[] = {"- (the crate panics: wrong MatchKind)", "gram4 count kernel (one LDS lookup per byte)", "gram count + checksum kernel",
                                "gram wide-alphabet kernel (31-62 byte classes)", "gram tuple emitter", "pfx (hashed prefix filter + start-anchored walks, any alphabet)",
                                "segment scanners (one lane per segment)", "micro-step walker over the double array", "chain walkers (speculate / reconcile / emit)",
                                "selection over the tuple emitter's detection (find3 / left3: no state chain)"};
    static const char *why[] = {"", "not uploaded yet", "more distinct pattern bytes than the byte-class tables take", "tables do not fit the LDS",
                                "\"\" is a pattern", "duplicate patterns the tables cannot encode", "the iterator is a chain through its own matches",
                                "charwise automaton", "trie shape / table limits"};
    std::string s;
    for (int r = 0; r < DAAC_REQ_N; ++r) {
        s += req[r];
        s += ": ";
        if (f.plan_reason[r] == DAAC_WHY_NOT_UPLOADED) { s += "not uploaded yet (daac_pma_upload decides the plan)\n"; continue; }
        if (f.plan_kernel[r] == DAAC_KERNEL_NONE) { s += ker[0]; s += "\n"; continue; }
        s += "engine "; s += eng[f.plan_engine[r] < sizeof(eng) / sizeof(*eng) ? f.plan_engine[r] : 0];
        s += ", "; s += ker[f.plan_kernel[r] < sizeof(ker) / sizeof(*ker) ? f.plan_kernel[r] : 0];
        if (f.plan_reason[r] != DAAC_WHY_FASTEST) { s += "  [not the fastest family: "; s += why[f.plan_reason[r] < sizeof(why) / sizeof(*why) ? f.plan_reason[r] : 0]; s += "]"; }
        s += "\n";
    }
    if (buf && cap) {
        const size_t n = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size() + 1;
}

// Gives back what the handle keeps between calls beside its tables: the emitter's / selection kernels' workspace on every device it was
// uploaded to (option workspace_keep bounds it while it is kept).  A workspace a scan is using right now stays.
daac_status daac_pma_trim(daac_pma *pma) {
    if (!pma) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::lock_guard<std::mutex> g(pma->mu);
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (auto &kv : pma->dev) {
        DeviceTables *t = kv.second.get();
        if (!t || t->ws_busy.exchange(true)) continue;
        if (t->ws_p && hipSetDevice(kv.first) == hipSuccess) { (void)hipFree(t->ws_p); t->ws_p = nullptr; t->ws_bytes = 0; }
        t->ws_want.store(0);
        t->ws_busy.store(false);
    }
    if (have_prev) (void)hipSetDevice(prev);
    return DAAC_OK;
}

void daac_pma_free(daac_pma *pma) { delete pma; }

daac_status daac_pma_upload(daac_pma *pma, int device) {
    PmaScope scope_(pma);
    if (!pma) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::lock_guard<std::mutex> g(pma->mu);
    DeviceTables *t = nullptr;
    return upload_locked(pma, device, &t);
}

static daac_status scan_count_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin, int hay_is_device,
                                   void *stream_, uint64_t *count, uint64_t *checksum, uint64_t *result_dev, bool want_checksum) {
    PmaScope scope_(pma);
    if (!pma || (len && !hay) || (!result_dev && (!count || (want_checksum && !checksum))) || begin > len) {
        set_error("bad argument");
        return DAAC_ERR_INVALID_ARGUMENT;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceTables *t = nullptr;
    dbg_mark("count: entry");
    daac_status st = check_mode_kind(pma, mode);
    if (st != DAAC_OK) return st;
    if ((st = get_tables(pma, &t)) != DAAC_OK) return st;
    // which GRAM table set serves this request: the second one where it applies (count only: always; with checksum: when
    // CID/H fit next to M), else the first
    const int64_t gv = OPT(gram_version);
    // (measured on cfg3: with the checksum both table sets spend three LDS lookups per position and the first is a little
    // faster; `.count()` alone needs one lookup per position on the second and runs 20-25 % faster there)
    const bool g1_can = t->gram_ok && gv != 2;
    const bool g2_can = t->gram2_ok && (!want_checksum || t->gram2.exact_ok) && gv != 1 && !(gv == 0 && want_checksum && g1_can) && !(gv == 3 && want_checksum && g1_can);
    const bool gw_can = t->gramw_ok && (!want_checksum || t->gramw.exact_ok);  // wide alphabets: built only where the others are not
    const bool use_gram = !pma->charwise && mode == DAAC_FIND_OVERLAPPING && pma->host.is_standard() && len - begin < (1ull << 35) &&
                          (engine == DAAC_ENGINE_GRAM || (engine == DAAC_ENGINE_AUTO && (g2_can || g1_can || gw_can)));
    // PFX: `.count()` for automata over any byte alphabet — what AUTO takes where the GRAM tables do not apply
    bool use_pfx = !pma->charwise && mode == DAAC_FIND_OVERLAPPING && pma->host.is_standard() && t->pfx_ok &&
                   len - begin < (1ull << 35) && (engine == DAAC_ENGINE_PFX || (engine == DAAC_ENGINE_AUTO && !use_gram));
    if (use_pfx && engine == DAAC_ENGINE_AUTO && OPT(pfx_probe) != 0) {
        // PFX is a filter: where the text's G-grams are mostly trie prefixes the micro-step walker over the double array is faster.  A
        // synchronous scan of a device haystack of 32 MiB or more samples the text (one small kernel + a read-back) and leaves its
        // verdict in the handle; every other call goes by the last verdict (none yet: PFX).
        int dense = t->pfx_dense.load();
        if (hay_is_device && !result_dev && len - begin >= (32ull << 20) && t->pfx_probe_word) {
            unsigned int *pin = pinned_words();
            unsigned int got = 0;
            unsigned int *tmp = const_cast<unsigned int *>(t->pfx_probe_word);   // (concurrent scans of one handle may read each other's sample: any of them is a sample)
            HIP_TRY(launch_pfx_probe(t->pfx, hay + begin, len - begin, tmp, stream));
            HIP_TRY(hipMemcpyAsync(pin ? pin : &got, tmp, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (pin) got = *pin;
            dense = got > static_cast<unsigned int>(std::max<int64_t>(0, OPT(pfx_probe))) ? 1 : 0;
            t->pfx_dense.store(dense);
        }
        if (dense > 0) use_pfx = false;   // (no verdict yet: PFX, as the handle's plan says)
    }
    if (engine == DAAC_ENGINE_PFX && !use_pfx) {
        set_error("PFX engine not available for this automaton / request (bytewise Standard automata without \"\", count (+ checksum) of find_overlapping)");
        return DAAC_ERR_UNSUPPORTED;
    }
    if (engine == DAAC_ENGINE_GRAM && (!use_gram || !(g2_can || g1_can || gw_can))) {
        set_error("GRAM engine not available for this automaton / mode");
        return DAAC_ERR_UNSUPPORTED;
    }
    const bool use_g2 = use_gram && g2_can;
    // `.count()` alone: gram4_kernels.hip on the renumbered tables (gram4.hpp), derived from the second table set
    Gram4Lds g4l{};
    uint32_t g4_ppl = 16;
    // (gram_version = 2 asks for gram2_kernels.hip, which counts with its checksum tables: a dictionary without room for those counts here)
    bool use_g4 = use_g2 && !want_checksum && t->gram4_ok && (gv == 4 || gv == 0 || (gv == 2 && !t->gram2.exact_ok));
    if (use_g4) {
        const bool want_rfull = OPT(gram2_rfull) != 0, want_arith = OPT(gram4_arith) != 0;
        const uint32_t waves = static_cast<uint32_t>(OPT(threads)) > 512 ? 16u : 8u;
        const int64_t ppl_opt = OPT(gram_ppl);
        // preference: the per-word directory first (two LDS reads per hit instead of five), then 32 positions per lane
        struct Shape { uint32_t ppl; bool rfull; } shapes[4] = {{32u, true}, {16u, true}, {32u, false}, {16u, false}};
        bool planned = false;
        for (const Shape &sh : shapes) {
            if ((ppl_opt == 16 || ppl_opt == 32) && sh.ppl != static_cast<uint32_t>(ppl_opt)) continue;
            if (sh.rfull && !want_rfull) continue;
            if (gram4_plan(t->gram4, sh.ppl, waves, sh.rfull, want_arith, 160u * 1024u, g4l)) { g4_ppl = sh.ppl; planned = true; break; }
        }
        if (!planned && waves == 16) {   // eight waves leave the tables more room
            for (const Shape &sh : shapes) {
                if (sh.rfull && !want_rfull) continue;
                if (gram4_plan(t->gram4, sh.ppl, 8, sh.rfull, want_arith, 160u * 1024u, g4l)) { g4_ppl = sh.ppl; planned = true; break; }
            }
        }
        use_g4 = planned;
    }
    if (gv == 4 && use_g2 && !want_checksum && !use_g4) {
        set_error("gram_version = 4: the gram4 tables are not there for this automaton (or do not fit the LDS with the launch shape asked for)");
        return DAAC_ERR_UNSUPPORTED;
    }
    if (use_g2 && !use_g4 && !t->gram2.exact_ok) {   // (`.count()` alone lives on gram4_kernels.hip; what is left of gram2_kernels.hip computes the checksum too)
        set_error("GRAM second table set: `.count()` runs on the gram4 kernel (gram_version 0 or 4); the count + checksum kernel needs tables this dictionary has no room for");
        return DAAC_ERR_UNSUPPORTED;
    }
    const bool use_gw = use_gram && !g2_can && !g1_can && gw_can;
    Plan pl;
    bool heads = false;
    if ((st = make_plan(pma, t, mode, (use_gram || use_pfx) ? DAAC_ENGINE_AUTO : engine, begin, len, pl, heads)) != DAAC_OK) return st;
    if (pl.a.nseg == 0 && begin == 0) pl.a.nseg = 1;  // ROOT's list at end = 0
    void *staged = nullptr;
    const uint8_t *dev_hay = hay;
    if (!hay_is_device && len) {
        const uint64_t from = begin > pl.a.halo ? begin - pl.a.halo : 0;
        if ((st = stage_window(hay, from, len, stream, &staged, &dev_hay)) != DAAC_OK) return st;
    }
    std::unique_ptr<void, void (*)(void *)> g1(staged, [](void *p) { if (p) (void)hipFree(p); });
    pl.a.hay = dev_hay;
    unsigned long long *d_res = reinterpret_cast<unsigned long long *>(result_dev);
    DevBuf own;
    if (!d_res) { HIP_TRY(own.alloc(3 * sizeof(unsigned long long), stream)); d_res = static_cast<unsigned long long *>(own.p); }
    pl.a.result = d_res;
    // find_iter over a whole haystack of a dictionary the emitter serves: selection over per-position flags instead of a walk (find3_kernels.hip)
    bool find3_served = false;
    unsigned long long find3_sums[3] = {0, 0, 0};
    dbg_mark("count: plan made");
    if (!pma->charwise && engine == DAAC_ENGINE_AUTO && !pma->root_has_output() && len != begin &&
        ((mode == DAAC_FIND && pma->host.is_standard()) || (mode == DAAC_LEFTMOST_FIND && !pma->host.is_standard()))) {
        if ((st = find_count3(pma, t, dev_hay, begin, len, stream, d_res, want_checksum, mode == DAAC_LEFTMOST_FIND, find3_sums, &find3_served)) != DAAC_OK) return st;
    }
    dbg_mark("count: find3 back");
    ChainBuffers chain_buffers;
    if (!find3_served && pl.a.nseg != 0 && (st = chain_resolve(pma, t, pl, stream, chain_buffers)) != DAAC_OK) return st;
    if (!find3_served) HIP_TRY(hipMemsetAsync(d_res, 0, 3 * sizeof(unsigned long long), stream));
    void *flagbuf = nullptr;
    if (pl.leftmost && pma->root_has_output()) {  // the one scan that can hit the non-terminating corner
        HIP_TRY(hipMalloc(&flagbuf, sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(flagbuf, 0, sizeof(unsigned long long), stream));
        pl.a.flags = static_cast<unsigned long long *>(flagbuf);
    }
    std::unique_ptr<void, void (*)(void *)> g3(flagbuf, [](void *p) { if (p) (void)hipFree(p); });
    if (!find3_served) g_last_engine = use_pfx ? DAAC_ENGINE_PFX : use_gram ? DAAC_ENGINE_GRAM : (pl.tier ? DAAC_ENGINE_TIERED : DAAC_ENGINE_DARRAY);
    if (find3_served) {
    } else if ((use_gram || use_pfx) && len != begin) {
        // A shard [begin, len): the occurrences with their end in (begin, len] = those of [from, len) scanned as a haystack of its own,
        // from = begin - halo, minus those of [from, begin) scanned as a haystack of its own (what lies wholly inside the halo) — two launches
        // of the same kernel, the second over at most max_pattern_len - 1 bytes into a scratch result, and one fix-up kernel that subtracts
        // and re-bases the ends (they were counted from `from`).  All on the stream, nothing read back.  (Until round 5 the shard itself was
        // scanned and the occurrences across `begin` came from a materialising scan of a sliver, synchronised and summed on the host:
        // ~1 ms per call, a quarter of a 4 GiB shard's scan — what every rank of a multi-GPU scan paid.)
        const uint64_t from = begin - std::min<uint64_t>(begin, pl.a.halo);
        const uint8_t *sub = dev_hay + from;
        GramArgs ga{};
        ga.lead = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(sub) & 15u);
        ga.hay_al = sub - ga.lead;
        ga.vlen = ga.lead + static_cast<uint64_t>(len - from);
        // a power of two >= 2 KiB: regions then never straddle a multiple of 4 GiB (the kernel keeps 32-bit positions per epoch)
        uint64_t region = 2048;
        // (second table set: 256 KiB regions once there are several per wave — a region's start costs a handful of dependent
        // loads and the refill of the prefetch pipeline: 64 KiB regions measured 2-6 % slower on 4 GiB)
        const int64_t region_opt = OPT(gram_region) > 0 ? OPT(gram_region)
                                   : (use_g2 || use_pfx) ? ((len - from) >= (1ull << 31) ? 262144 : 65536) : 16384;
        while (region * 2 <= static_cast<uint64_t>(std::max<int64_t>(2048, region_opt)) && region < (1ull << 30)) region *= 2;
        ga.ppl = use_pfx ? 16 : use_g4 ? g4_ppl : (!use_g2 && !use_gw && !t->gram.has_short && OPT(gram_ppl) != 16) ? 32 : 16;
        ga.region_bytes = region;
        ga.nregions = (ga.vlen + region - 1) / region;
        ga.result = d_res;
        uint32_t threads = static_cast<uint32_t>(OPT(threads));
        threads = std::min(1024u, std::max(64u, threads & ~63u));
        if (use_gw) threads = 1024;  // the wide kernel has one launch shape
        if (use_g4) threads = g4l.threads;
        if (use_pfx) threads = t->pfx.threads;
        const uint32_t wpb = threads / 64;
        uint32_t bpc = static_cast<uint32_t>(OPT(blocks_per_cu));
        const uint32_t gram_lds = use_pfx ? t->pfx.lds_bytes : use_g4 ? g4l.lds_bytes : use_gw ? (want_checksum ? t->gramw.lds_exact : t->gramw.lds_count)
                                         : use_g2 ? gram2_lds_bytes(t->gram2, want_checksum) : t->gram.lds_bytes;
        if (bpc == 0) bpc = std::max(1u, std::min(2048u / threads, (160u * 1024u) / gram_lds));
        const uint32_t blocks = static_cast<uint32_t>(
            std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(t->num_cu) * bpc, (ga.nregions + wpb - 1) / wpb)));
        // room for what one step can queue at worst (64 * ppl + 128 walkers) on top of a useful fill level
        ga.wq_slab = static_cast<uint32_t>(std::max<int64_t>(64 * ga.ppl + 128 + 64, OPT(gram_slab)));  // (a step can queue 64 * ppl walkers)
        // more than ~1 % of the (K+1)-grams are trie prefixes: some lane of the wave hits on nearly every position
        {
            const uint64_t n_deep = use_gw ? t->gramw.n_deep : use_g2 ? t->gram2.n_deep : t->gram.n_deep, C = use_gw ? t->gramw.C : use_g2 ? t->gram2.C : t->gram.C,
                           K = use_gw ? 2 : use_g2 ? t->gram2.K : t->gram.K;
            ga.dense = OPT(gram_dense) >= 0 ? OPT(gram_dense) != 0 : n_deep * 100 > C * C * C * (K == 3 ? C : 1);
        }
        // gram4: tail records from the hit record on pay on text made of dictionary words (+20 %) and cost 3-4 % elsewhere; unless
        // the option decides, every workgroup samples the haystack at its start and runs the variant the text calls for
        const int64_t tail_opt = OPT(gram3_tail);
        void *wq = nullptr;
        HIP_TRY(dev_malloc(&wq, static_cast<size_t>(blocks) * wpb * ga.wq_slab * ((use_g4 || use_pfx) ? sizeof(uint4) : sizeof(uint2)), stream));
        ga.wq = static_cast<uint2 *>(wq);
        ga.sel_want = tail_opt < 0 ? ((len - begin) >= (1ull << 20) ? 2u : 0u) : tail_opt > 0 ? 1u : 0u;
        auto launch_count = [&](const GramArgs &g, uint32_t nblocks) -> hipError_t {
            return use_pfx ? launch_pfx_scan(t->pfx, g, want_checksum, nblocks, stream)
                   : use_g4 ? launch_gram4_scan(t->gram4, g, g4l, nblocks, stream)
                   : use_gw ? launch_gram2w_scan(t->gramw, g, want_checksum, nblocks, stream)
                   : use_g2 ? launch_gram2_scan(t->gram2, g, want_checksum, nblocks, threads, stream)
                            : launch_gram_scan(t->gram, g, nblocks, threads, stream);
        };
        hipError_t le = launch_count(ga, blocks);
        DevBuf halo_res;
        if (le == hipSuccess && from != begin) {   // what lies wholly inside the halo, counted the same way (ends from `from` as well)
            le = halo_res.alloc(3 * sizeof(unsigned long long), stream);
            if (le == hipSuccess) le = hipMemsetAsync(halo_res.p, 0, 3 * sizeof(unsigned long long), stream);
            GramArgs gh = ga;
            gh.vlen = ga.lead + (begin - from);
            gh.nregions = (gh.vlen + gh.region_bytes - 1) / gh.region_bytes;   // (one: a halo is shorter than any region)
            gh.result = static_cast<unsigned long long *>(halo_res.p);
            gh.sel_want = 0;
            if (le == hipSuccess) le = launch_count(gh, 1);   // (the first workgroup's slab of the walker queue: the scan before has drained it)
            if (le == hipSuccess) {
                hipLaunchKernelGGL(shard_subtract_kernel, dim3(1), dim3(1), 0, stream, d_res, static_cast<const unsigned long long *>(halo_res.p),
                                   static_cast<unsigned long long>(from & 0xffffffffull));
                le = hipGetLastError();
            }
        } else if (le == hipSuccess && from != 0) {
            hipLaunchKernelGGL(shard_fixup_kernel, dim3(1), dim3(1), 0, stream, d_res, static_cast<unsigned long long>(from & 0xffffffffull), 0ull, 0ull, 0ull);
            le = hipGetLastError();
        }
        dev_free(wq, stream);
        HIP_TRY(le);
    } else if (pl.a.nseg != 0) {
        HIP_TRY(launch(t, pl, 0, heads, stream));
    }
    unsigned long long flagv = 0;
    if (flagbuf) HIP_TRY(hipMemcpyAsync(&flagv, flagbuf, sizeof(flagv), hipMemcpyDeviceToHost, stream));
    if (result_dev && !count) {
        if (staged || flagbuf) HIP_TRY(hipStreamSynchronize(stream));
        return (flagv & 1ull) ? diverged() : DAAC_OK;
    }
    unsigned long long r[3];
    if (find3_served) {   // (its windows were read back one by one)
        std::memcpy(r, find3_sums, sizeof(r));
    } else {
        HIP_TRY(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    dbg_mark("count: result read");
    if (flagv & 1ull) return diverged();
    if (count) *count = r[0];
    if (checksum) *checksum = ((r[1] & 0xffffffffull) << 32) | (r[2] & 0xffffffffull);
    return DAAC_OK;
}

daac_status daac_scan_count_range(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin, int hay_is_device,
                                  void *stream, uint64_t *count, uint64_t *checksum, uint64_t *result_dev) {
    return scan_count_impl(pma, mode, engine, hay, len, begin, hay_is_device, stream, count, checksum, result_dev, true);
}

daac_status daac_scan_count(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream,
                            uint64_t *count, uint64_t *checksum, uint64_t *result_dev) {
    return scan_count_impl(pma, mode, engine, hay, len, 0, hay_is_device, stream, count, checksum, result_dev, true);
}

daac_status daac_scan_count_only_range(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin, int hay_is_device,
                                       void *stream, uint64_t *count, uint64_t *result_dev) {
    return scan_count_impl(pma, mode, engine, hay, len, begin, hay_is_device, stream, count, nullptr, result_dev, false);
}

// One haystack sharded across the devices of a node (BASELINE configs[3]): the product's own form of what bench.py does with
// one process per GPU.  One host thread per shard: makes the shard's device current, uploads the tables there if they are not yet, runs
// daac_scan_count[_only]_range over [halo | shard] with begin = halo (matches with their end inside the shard, wherever they start), and
// the host adds the counts and the two checksum sums — `base` re-bases a shard's ends (S2 += low32(base) * S1).  No collective: RCCL is
// for callers that run one process per GPU (daachorse_amd/dist.py) and reduce {count, S1, S2} themselves.
daac_status daac_scan_count_multi(daac_pma *pma, int mode, int engine, const daac_shard *shards, size_t n, int hay_is_device, uint64_t *count,
                                  uint64_t *checksum) {
    PmaScope scope_(pma);
    if (!pma || !count || (n && !shards)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    daac_status st = check_mode_kind(pma, mode);
    if (st != DAAC_OK) return st;
    if (mode != DAAC_FIND_OVERLAPPING && mode != DAAC_FIND_OVERLAPPING_NO_SUFFIX) {
        set_error("daac_scan_count_multi: find_iter / leftmost_find_iter are chains through their own matches; a shard does not know where the chain enters it");
        return DAAC_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    const size_t halo_need = pma->halo();
    for (size_t k = 0; k < n; ++k) {
        if (shards[k].device < 0 || shards[k].device >= ndev || ((shards[k].len + shards[k].halo) && !shards[k].hay)) {
            set_error("daac_scan_count_multi: bad shard (device ordinal / null haystack)");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
        if (shards[k].halo < halo_need && shards[k].halo < shards[k].base) {   // fewer bytes in front than a match may reach back, and not the haystack's start
            set_error("daac_scan_count_multi: a shard needs max_pattern_len - 1 bytes of the haystack in front of it (charwise: max_pattern_len, at least 3)");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
    }
    int dev0 = 0;
    HIP_TRY(hipGetDevice(&dev0));
    struct Out { daac_status st = DAAC_OK; uint64_t count = 0, checksum = 0; int engine = DAAC_ENGINE_AUTO; std::string err; };
    std::vector<Out> outs(n);
    auto work = [&](size_t k) {
        Out &o = outs[k];
        const daac_shard &sh = shards[k];
        if (hipSetDevice(sh.device) != hipSuccess) { o.st = DAAC_ERR_DEVICE; o.err = "hipSetDevice failed"; (void)hipGetLastError(); return; }
        o.st = checksum ? scan_count_impl(pma, mode, engine, sh.hay, sh.halo + sh.len, sh.halo, hay_is_device, nullptr, &o.count, &o.checksum, nullptr, true)
                        : scan_count_impl(pma, mode, engine, sh.hay, sh.halo + sh.len, sh.halo, hay_is_device, nullptr, &o.count, nullptr, nullptr, false);
        o.engine = g_last_engine;
        if (o.st != DAAC_OK) o.err = daac_last_error();
    };
    std::vector<std::thread> threads;
    for (size_t k = 1; k < n; ++k) threads.emplace_back(work, k);
    if (n) work(0);
    for (std::thread &t : threads) t.join();
    (void)hipSetDevice(dev0);
    uint64_t total = 0;
    uint32_t s1 = 0, s2 = 0;
    for (size_t k = 0; k < n; ++k) {
        if (outs[k].st != DAAC_OK) { set_error("shard " + std::to_string(k) + " (device " + std::to_string(shards[k].device) + "): " + outs[k].err); return outs[k].st; }
        total += outs[k].count;
        const uint32_t k1 = static_cast<uint32_t>(outs[k].checksum >> 32), k2 = static_cast<uint32_t>(outs[k].checksum);
        const uint32_t shift = static_cast<uint32_t>(shards[k].base - shards[k].halo);   // ends were counted from the shard's first resident byte
        s1 += k1;
        s2 += k2 + shift * k1;
    }
    if (n) g_last_engine = outs[0].engine;
    *count = total;
    if (checksum) *checksum = (static_cast<uint64_t>(s1) << 32) | s2;
    return DAAC_OK;
}

daac_status daac_scan(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream_,
                      daac_matches **out) {
    PmaScope scope_(pma);
    if (!pma || !out || (len && !hay)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceTables *t = nullptr;
    daac_status st = check_mode_kind(pma, mode);
    if (st != DAAC_OK) return st;
    if ((st = get_tables(pma, &t)) != DAAC_OK) return st;
    void *staged = nullptr;
    const uint8_t *dev_hay = hay;
    if (!hay_is_device && len) {
        if ((st = stage_window(hay, 0, len, stream, &staged, &dev_hay)) != DAAC_OK) return st;
    }
    std::unique_ptr<void, void (*)(void *)> g1(staged, [](void *p) { if (p) (void)hipFree(p); });
    std::unique_ptr<daac_matches> m(new daac_matches);
    if ((st = scan_range_materialize(pma, t, mode, engine, dev_hay, 0, len, len, stream, m->v, nullptr)) != DAAC_OK) return st;
    *out = m.release();
    return DAAC_OK;
}

static daac_status scan_device_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream_,
                                    void **dev_out, uint64_t *count, bool f16);

daac_status daac_scan_device(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream_,
                             daac_match **dev_out, uint64_t *count) {
    PmaScope scope_(pma);
    return scan_device_impl(pma, mode, engine, hay, len, hay_is_device, stream_, reinterpret_cast<void **>(dev_out), count, false);
}
daac_status daac_scan_device16(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream_,
                               daac_match16 **dev_out, uint64_t *count) {
    PmaScope scope_(pma);
    return scan_device_impl(pma, mode, engine, hay, len, hay_is_device, stream_, reinterpret_cast<void **>(dev_out), count, true);
}

static daac_status scan_device_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream_,
                                    void **dev_out, uint64_t *count, bool f16) {
    if (!pma || !dev_out || !count || (len && !hay)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceTables *t = nullptr;
    dbg_mark("scan_device: entry");
    daac_status st = check_mode_kind(pma, mode);
    if (st != DAAC_OK) return st;
    if ((st = get_tables(pma, &t)) != DAAC_OK) return st;
    void *staged = nullptr;
    const uint8_t *dev_hay = hay;
    if (!hay_is_device && len) {
        if ((st = stage_window(hay, 0, len, stream, &staged, &dev_hay)) != DAAC_OK) return st;
    }
    std::unique_ptr<void, void (*)(void *)> g1(staged, [](void *p) { if (p) (void)hipFree(p); });
    DevMatches dm;
    dm.f16 = f16;
    if ((st = scan_range_device(pma, t, mode, engine, dev_hay, 0, len, len, stream, dm, nullptr)) != DAAC_OK) return st;
    if (f16 && !dm.f16_done && dm.n != 0) {  // an engine that writes daac_match: repacked on the device
        void *d16 = nullptr;
        HIP_TRY(dev_malloc(&d16, dm.n * 16, stream));
        hipLaunchKernelGGL(repack16_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>(65535, (dm.n + 255) / 256))), dim3(256), 0, stream,
                           dm.p, static_cast<uint4 *>(d16), static_cast<unsigned long long>(dm.n));
        HIP_TRY(hipGetLastError());
        dev_free(dm.release_keep_n(), stream);
        dm.p = static_cast<daac_match *>(d16);
    }
    HIP_TRY(hipStreamSynchronize(stream));
    *count = dm.n;
    *dev_out = dm.release();
    return DAAC_OK;
}

void daac_device_free(void *p) {
    // hipFree is legal for stream-ordered allocations of any device and synchronises: the list may come from another device's
    // pool than the current one, and its consumer ran on a stream this library never saw
    if (p) (void)hipFree(p);
}

daac_status daac_device_to_host(void *dst, const void *dev_src, size_t bytes) {
    if (bytes && (!dst || !dev_src)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (bytes) HIP_TRY(hipMemcpy(dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return DAAC_OK;
}

size_t daac_matches_count(const daac_matches *m) { return m ? m->v.size() : 0; }
const daac_match *daac_matches_data(const daac_matches *m) { return m && m->v.size() != 0 ? m->v.p : nullptr; }
void daac_matches_free(daac_matches *m) { delete m; }

}  // extern "C"

// ------------------------------------------------------------------------------ lazy iterator
// Iterator::next() over a haystack scanned window by window.  A worker thread runs the windows ahead of the consumer, three stages in
// flight on three streams: the H2D copy of window k + 1 (host haystacks), the scan of window k, the D2H copy of window k - 1's tuples
// — 16-byte tuples {end, length, value} (the crate's own Match fields), expanded to daac_match by daac_iter_next, handed out as they
// are by daac_iter_next_batch.  Staging buffers, result lists (stream-ordered pool) and the page-locked host buffers are reused from
// window to window; round 3 allocated, copied, scanned and copied back one window at a time (3.5 GB/s on cfg3).
namespace {

// page-locked blocks kept for the next iterator (pinning a GB costs ~50 ms); never freed at exit: the HIP runtime may be gone by then
struct PinnedPool {
    std::mutex mu;
    struct Block { void *p; size_t bytes; };
    std::vector<Block> free_blocks;
    void *take(size_t want, size_t *got) {
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].bytes >= want && (best == free_blocks.size() || free_blocks[i].bytes < free_blocks[best].bytes)) best = i;
            if (best != free_blocks.size() && free_blocks[best].bytes <= 4 * want + (64u << 20)) {
                void *p = free_blocks[best].p;
                *got = free_blocks[best].bytes;
                free_blocks.erase(free_blocks.begin() + static_cast<long>(best));
                return p;
            }
        }
        void *q = nullptr;
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *got = want;
        return q;
    }
    void give(void *p, size_t bytes) {
        if (!p) return;
        void *drop = nullptr;
        {
            std::lock_guard<std::mutex> g(mu);
            free_blocks.push_back(Block{p, bytes});
            if (free_blocks.size() > 4) {   // the smallest goes
                size_t k = 0;
                for (size_t i = 1; i < free_blocks.size(); ++i) if (free_blocks[i].bytes < free_blocks[k].bytes) k = i;
                drop = free_blocks[k].p;
                free_blocks.erase(free_blocks.begin() + static_cast<long>(k));
            }
        }
        if (drop) (void)hipHostFree(drop);
    }
};
PinnedPool &pinned_pool() { static PinnedPool *p = new PinnedPool; return *p; }

// What an iterator needs on the device — four streams, its events, two staging buffers — kept per device for the next iterator: creating
// and destroying them cost ~4 ms per iterator, a sixth of a sparse 1 GiB scan (profiles/r04_iterator.txt).  Never freed at exit.
struct IterDeviceKit {
    int device = -1;
    hipStream_t s_scan = nullptr, s_h2d = nullptr, s_d2h = nullptr, s_d2h2 = nullptr;
    hipEvent_t staged_ev[2] = {nullptr, nullptr}, half_ev = nullptr;
    void *stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
};
struct IterKitPool {
    std::mutex mu;
    std::vector<IterDeviceKit *> free_kits;
    IterDeviceKit *take(int device) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_kits.size(); ++i)
                if (free_kits[i]->device == device) { IterDeviceKit *k = free_kits[i]; free_kits.erase(free_kits.begin() + static_cast<long>(i)); return k; }
        }
        std::unique_ptr<IterDeviceKit> k(new IterDeviceKit);
        k->device = device;
        bool ok = hipStreamCreateWithFlags(&k->s_scan, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&k->s_h2d, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&k->s_d2h, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&k->s_d2h2, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&k->staged_ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&k->staged_ev[1], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&k->half_ev, hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); return nullptr; }   // (a half-made kit leaks a stream or two: this is an out-of-resources path)
        return k.release();
    }
    void give(IterDeviceKit *k) {
        if (!k) return;
        std::lock_guard<std::mutex> g(mu);
        if (free_kits.size() < 8) { free_kits.push_back(k); return; }
        // (more than eight idle kits: this one's buffers go back; the streams are few and stay)
        for (void *&b : k->stage) { if (b) (void)hipFree(b); b = nullptr; }
        k->stage_bytes = 0;
        free_kits.push_back(k);
    }
};
IterKitPool &iter_kits() { static IterKitPool *p = new IterKitPool; return *p; }

constexpr int kIterSlots = 3;
struct IterWindow {
    daac_status st = DAAC_OK;
    std::string err;
    uint64_t n = 0;                 // tuples of the window
    uint64_t base = 0;              // compact: ends count from here (the window's first byte)
    daac_match16 *host = nullptr;   // page-locked (pageable if pinning failed); compact: daac_match8 tuples
    size_t host_bytes = 0;
    bool host_pinned = false;
    hipEvent_t copied = nullptr;    // the tuples are in `host`
    int engine = DAAC_ENGINE_AUTO;
};

}  // namespace

struct daac_iter {
    daac_pma *pma = nullptr;
    int mode = 0, engine = 0, device = 0;
    const uint8_t *hay = nullptr;
    uint64_t len = 0;
    bool hay_is_device = false;
    bool restart = false;          // find_iter / leftmost_find_iter: windows end at sync points
    bool compact = false;          // daac_iter_open_compact: 8-byte tuples over PCIe {value, end relative to the window | length << end_bits}
    uint32_t end_bits = 32;        // ... 32 - bits of the longest pattern's length; a window spans less than 2^end_bits bytes
    void *owned_dev = nullptr;     // host haystack staged once (restart modes read past a window's nominal end)
    hipStream_t user_stream = nullptr;
    // the worker and its three streams
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    IterWindow win[kIterSlots];
    uint64_t produced = 0, consumed = 0;   // windows handed to / taken back from the consumer
    bool stop = false, finished = false;
    IterDeviceKit *kit = nullptr;
    hipStream_t s_scan = nullptr, s_h2d = nullptr, s_d2h = nullptr, s_d2h2 = nullptr;
    // the consumer's view of the window it is reading
    const daac_match16 *cur = nullptr;   // (compact: daac_match8 tuples behind this pointer)
    uint64_t cur_base = 0;
    size_t cur_n = 0, pos = 0;
    bool holding = false;
    bool started = false;          // the worker runs from the first next() on: an iterator that is only opened (the Rust cursor's
                                   // `.count()` fast path opens one and counts beside it) scans and copies nothing

    void run();
};

void daac_iter::run() {
    PmaScope scope_(pma);
    (void)hipSetDevice(device);
    auto fail_with = [&](IterWindow &w, daac_status st) { w.st = st; w.err = last_error_cstr(); w.n = 0; };
    DeviceTables *t = nullptr;
    uint64_t window = std::max<uint64_t>(4096, static_cast<uint64_t>(OPT(iter_window)));
    if (compact) window = std::min<uint64_t>(window, (1ull << end_bits) - pma->halo() - 4096);   // (a restart window runs on to a sync point: less than a pattern further)
    // windows grow from 16 MiB to the full size: the consumer has its first matches after a small window's scan and copy, not a big one's
    auto window_of = [&](uint64_t k) -> uint64_t { return std::min<uint64_t>(window, (16ull << 20) << std::min<uint64_t>(k, 16)); };
    const uint64_t halo = pma->halo();
    const bool per_window_copy = !hay_is_device;   // (restart modes were staged whole at open)
    void **stage = kit->stage;
    hipEvent_t *staged_ev = kit->staged_ev;
    uint64_t begin = 0;
    daac_status st0 = get_tables(pma, &t);
    if (st0 == DAAC_OK && per_window_copy) {
        const size_t want = static_cast<size_t>(std::min<uint64_t>(len, window) + halo + 64);
        if (kit->stage_bytes < want) {
            for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipFree(stage[i]); stage[i] = nullptr; }
            kit->stage_bytes = 0;
            if (hipMalloc(&stage[0], want) != hipSuccess || hipMalloc(&stage[1], want) != hipSuccess) st0 = hip_fail(hipGetLastError(), "iterator staging buffers");
            else kit->stage_bytes = want;
        }
    }
    // window k of a host haystack: bytes [from, end) -> stage[k & 1], asynchronously on the copy stream
    auto issue_stage = [&](uint64_t k, uint64_t wb) -> hipError_t {
        const uint64_t we = std::min<uint64_t>(len, wb + window_of(k));
        const uint64_t from = wb > halo ? wb - halo : 0;
        const uint64_t skew = from & 15;  // keep the haystack's 16-byte phase for the vector loop
        if (we > from) {
            const hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(stage[k & 1]) + skew, hay + from, we - from, hipMemcpyHostToDevice, s_h2d);
            if (e != hipSuccess) return e;
        }
        return hipEventRecord(staged_ev[k & 1], s_h2d);
    };
    if (st0 == DAAC_OK && per_window_copy && len != 0) {
        const hipError_t e = issue_stage(0, 0);
        if (e != hipSuccess) st0 = hip_fail(e, "iterator: host-to-device copy");
    }
    for (uint64_t k = 0;; ++k) {
        {   // a free slot (the consumer is at most kIterSlots - 1 windows behind)
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return stop || produced - consumed < static_cast<uint64_t>(kIterSlots); });
            if (stop) break;
        }
        IterWindow &w = win[k % kIterSlots];
        w.st = DAAC_OK; w.err.clear(); w.n = 0;
        bool last = false;
        if (st0 != DAAC_OK) {
            fail_with(w, st0);
            last = true;
        } else {
            const uint64_t end = std::min<uint64_t>(len, begin + window_of(k));
            const uint8_t *dev_hay = hay;
            if (per_window_copy) {
                const uint64_t from = begin > halo ? begin - halo : 0;
                dev_hay = static_cast<const uint8_t *>(stage[k & 1]) + (from & 15) - from;
                (void)hipStreamWaitEvent(s_scan, staged_ev[k & 1], 0);
            }
            uint64_t next_begin = end;
            DevMatches dm;
            dm.f16 = true;
            daac_status st = DAAC_OK;
            // the next window's bytes go to the other staging buffer while this window is scanned (its last reader, the scan of
            // window k - 1, is through; only the overlapping modes come here, and their windows begin where the last one ended)
            if (per_window_copy && end < len) {
                const hipError_t e = issue_stage(k + 1, end);
                if (e != hipSuccess) st = hip_fail(e, "iterator: host-to-device copy");
            }
            if (st == DAAC_OK) st = scan_range_device(pma, t, mode, engine, dev_hay, begin, end, len, s_scan, dm, &next_begin);
            w.engine = g_last_engine;
            if (st == DAAC_OK && !dm.f16_done && dm.n != 0) {   // an engine that writes daac_match: repacked on the device
                void *d16 = nullptr;
                if (dev_malloc(&d16, dm.n * 16, s_scan) != hipSuccess) { st = hip_fail(hipGetLastError(), "iterator: repack buffer"); }
                else {
                    hipLaunchKernelGGL(repack16_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>(65535, (dm.n + 255) / 256))), dim3(256), 0, s_scan, dm.p,
                                       static_cast<uint4 *>(d16), static_cast<unsigned long long>(dm.n));
                    dev_free(dm.release_keep_n(), s_scan);
                    dm.p = static_cast<daac_match *>(d16);
                }
            }
            if (st == DAAC_OK && compact && dm.n != 0) {   // ends relative to the window's first byte, the length above them: 8 bytes per tuple over the link
                void *d8 = nullptr;
                if (next_begin - begin >= (1ull << end_bits)) { set_error("iterator: a window ran past what its compact form can say"); st = DAAC_ERR_UNSUPPORTED; }
                else if (dev_malloc(&d8, dm.n * 8, s_scan) != hipSuccess) { st = hip_fail(hipGetLastError(), "iterator: repack buffer"); }
                else {
                    hipLaunchKernelGGL(repack8_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>(65535, (dm.n + 255) / 256))), dim3(256), 0, s_scan,
                                       reinterpret_cast<const uint4 *>(dm.p), static_cast<uint2 *>(d8), static_cast<unsigned long long>(dm.n),
                                       static_cast<unsigned long long>(begin), end_bits);
                    dev_free(dm.release_keep_n(), s_scan);
                    dm.p = static_cast<daac_match *>(d8);
                }
            }
            w.base = begin;
            // the list (repacked or not) is complete before another stream copies it, and before the staging buffer is written again
            if (st == DAAC_OK && hipStreamSynchronize(s_scan) != hipSuccess) st = hip_fail(hipGetLastError(), "iterator: scan");
            if (st == DAAC_OK && dm.n != 0) {
                const size_t need = dm.n * (compact ? 8 : sizeof(daac_match16));
                if (w.host_bytes < need) {
                    if (w.host) { if (w.host_pinned) pinned_pool().give(w.host, w.host_bytes); else std::free(w.host); }
                    w.host = nullptr; w.host_bytes = 0;
                    // sized for a FULL window of this density at once (the first windows are small: a buffer that grew with them would be
                    // pinned four times over, at ~50 ms per GB)
                    const uint64_t cur_w = std::max<uint64_t>(1, end - begin);
                    const size_t full = static_cast<size_t>(static_cast<double>(need) * (static_cast<double>(std::max<uint64_t>(window, cur_w)) / static_cast<double>(cur_w)) * 1.15) + 4096;
                    size_t got = 0;
                    void *q = pinned_pool().take(std::max(full, need + need / 4), &got);
                    w.host_pinned = q != nullptr;
                    if (!q) { q = std::malloc(need); got = need; }
                    if (!q) { set_error("out of host memory for the match list"); st = DAAC_ERR_AUTOMATON_SCALE; }
                    w.host = static_cast<daac_match16 *>(q);
                    w.host_bytes = q ? got : 0;
                }
                if (st == DAAC_OK) {
                    // (the list is complete: s_scan was waited for above) the copy runs beside the next window's scan
                    // two halves on two streams: one copy engine alone stayed at 40 GB/s of the link's ~55 (profiles/r04_iterator.txt)
                    const size_t half = need >= (8u << 20) ? (need / 2) & ~size_t(4095) : need;
                    hipError_t e = hipMemcpyAsync(w.host, dm.p, half, hipMemcpyDeviceToHost, s_d2h);
                    if (e == hipSuccess && half != need) {
                        e = hipMemcpyAsync(reinterpret_cast<char *>(w.host) + half, reinterpret_cast<const char *>(dm.p) + half, need - half, hipMemcpyDeviceToHost, s_d2h2);
                        if (e == hipSuccess) e = hipEventRecord(kit->half_ev, s_d2h2);
                        if (e == hipSuccess) e = hipStreamWaitEvent(s_d2h, kit->half_ev, 0);
                    }
                    if (e == hipSuccess) e = hipEventRecord(w.copied, s_d2h);
                    if (e != hipSuccess) st = hip_fail(e, "iterator: device-to-host copy");
                    dm.s = s_d2h;   // released behind the copy
                    w.n = dm.n;
                }
            }
            if (st != DAAC_OK) { fail_with(w, st); last = true; }
            begin = next_begin;
            if (next_begin >= len) last = true;
        }
        {
            std::lock_guard<std::mutex> g(mu);
            ++produced;
            if (last) finished = true;
        }
        cv.notify_all();
        if (last) break;
    }
    (void)hipStreamSynchronize(s_d2h);
    (void)hipStreamSynchronize(s_d2h2);
    (void)hipStreamSynchronize(s_h2d);
    (void)hipStreamSynchronize(s_scan);
    {
        std::lock_guard<std::mutex> g(mu);
        finished = true;
    }
    cv.notify_all();
}

extern "C" {

static daac_status iter_open_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream,
                                  bool compact, daac_iter **out);
void daac_iter_close(daac_iter *it);
daac_status daac_iter_open(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream,
                           daac_iter **out) {
    return iter_open_impl(pma, mode, engine, hay, len, hay_is_device, stream, false, out);
}
daac_status daac_iter_open_compact(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream,
                                   daac_iter **out) {
    return iter_open_impl(pma, mode, engine, hay, len, hay_is_device, stream, true, out);
}
static daac_status iter_open_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, int hay_is_device, void *stream,
                                  bool compact, daac_iter **out) {
    PmaScope scope_(pma);
    if (!pma || !out || (len && !hay)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    DeviceTables *t = nullptr;
    daac_status st = check_mode_kind(pma, mode);
    if (st != DAAC_OK) return st;
    if ((st = get_tables(pma, &t)) != DAAC_OK) return st;
    Plan pl;
    bool heads;
    if ((st = make_plan(pma, t, mode, (engine == DAAC_ENGINE_GRAM || engine == DAAC_ENGINE_PFX) ? DAAC_ENGINE_AUTO : engine, 0, len, pl, heads)) != DAAC_OK) return st;  // kind / mode checks up front
    std::unique_ptr<daac_iter> it(new daac_iter);
    it->pma = pma; it->mode = mode; it->engine = engine; it->hay = hay; it->len = len;
    it->compact = compact;
    if (compact) {   // length bits above the end: the longest pattern decides the split, and with it the largest window
        uint32_t lb = 1;
        while ((1ull << lb) <= pma->max_pattern_len()) ++lb;
        it->end_bits = 32 - lb;
        if ((1ull << it->end_bits) < 4ull * pma->halo() + (1ull << 20)) {
            set_error("patterns of " + std::to_string(pma->max_pattern_len()) + " bytes leave no room for a window in the compact tuple; use daac_iter_open");
            return DAAC_ERR_UNSUPPORTED;
        }
    }
    it->hay_is_device = hay_is_device != 0;
    it->user_stream = static_cast<hipStream_t>(stream);
    it->restart = pl.restart;
    HIP_TRY(hipGetDevice(&it->device));
    if (it->restart && !it->hay_is_device && len) {
        const uint8_t *virt = nullptr;
        if ((st = stage_window(hay, 0, len, it->user_stream, &it->owned_dev, &virt)) != DAAC_OK) return st;
        it->hay = virt;
        it->hay_is_device = true;
    }
    // whatever the caller queued on its stream (a device haystack being written, the staging copy above) comes first
    if (hipStreamSynchronize(it->user_stream) != hipSuccess) {
        const daac_status e = hip_fail(hipGetLastError(), "iterator: the caller's stream");
        if (it->owned_dev) (void)hipFree(it->owned_dev);
        return e;
    }
    it->kit = iter_kits().take(it->device);
    if (!it->kit) { if (it->owned_dev) (void)hipFree(it->owned_dev); set_error("iterator: no streams / events to be had"); return DAAC_ERR_DEVICE; }
    it->s_scan = it->kit->s_scan; it->s_h2d = it->kit->s_h2d; it->s_d2h = it->kit->s_d2h; it->s_d2h2 = it->kit->s_d2h2;
    for (IterWindow &w : it->win) {
        if (hipEventCreateWithFlags(&w.copied, hipEventDisableTiming) != hipSuccess) {
            const daac_status e = hip_fail(hipGetLastError(), "iterator: events");
            daac_iter_close(it.release());   // gives back the kit, the staged haystack and the events made so far
            return e;
        }
    }
    *out = it.release();
    return DAAC_OK;
}

// The next window's tuples (blocks until the worker has them); 1 = it->cur / cur_n are set, 0 = exhausted, < 0 = -daac_status
static int iter_advance(daac_iter *it) {
    if (!it->started) {   // the first next(): the worker starts scanning now
        it->started = true;
        it->worker = std::thread([it] { it->run(); });
    }
    for (;;) {
        if (it->holding) {   // give the slot back
            { std::lock_guard<std::mutex> g(it->mu); ++it->consumed; }
            it->cv.notify_all();
            it->holding = false;
            it->cur = nullptr; it->cur_n = 0; it->pos = 0;
        }
        {
            std::unique_lock<std::mutex> g(it->mu);
            it->cv.wait(g, [&] { return it->produced > it->consumed || it->finished; });
            if (it->produced == it->consumed) return 0;
        }
        IterWindow &w = it->win[it->consumed % kIterSlots];
        it->holding = true;
        if (w.st != DAAC_OK) { set_error(w.err); return -static_cast<int>(w.st); }
        g_last_engine = w.engine;
        if (w.n == 0) continue;
        if (hipEventSynchronize(w.copied) != hipSuccess) return -static_cast<int>(hip_fail(hipGetLastError(), "iterator: waiting for the tuples"));
        it->cur = w.host; it->cur_n = static_cast<size_t>(w.n); it->pos = 0; it->cur_base = w.base;
        return 1;
    }
}

int daac_iter_next(daac_iter *it, daac_match *m) {
    if (!it || !m) return -DAAC_ERR_INVALID_ARGUMENT;
    if (it->pos >= it->cur_n) {
        const int r = iter_advance(it);
        if (r <= 0) return r;
    }
    if (it->compact) {
        const daac_match8 &t = reinterpret_cast<const daac_match8 *>(it->cur)[it->pos++];
        m->end = it->cur_base + (t.end_len & ((1u << it->end_bits) - 1u)); m->start = m->end - (t.end_len >> it->end_bits); m->value = t.value; m->_pad = 0;
        return 1;
    }
    const daac_match16 &t = it->cur[it->pos++];
    m->start = t.end - t.length; m->end = t.end; m->value = t.value; m->_pad = 0;
    return 1;
}

int daac_iter_next_batch8(daac_iter *it, const daac_match8 **batch, size_t *n, uint64_t *end_base, uint32_t *end_bits) {
    if (!it || !batch || !n || !end_base || !end_bits) return -DAAC_ERR_INVALID_ARGUMENT;
    if (!it->compact) { set_error("daac_iter_next_batch8 serves iterators opened with daac_iter_open_compact"); return -DAAC_ERR_UNSUPPORTED; }
    *end_bits = it->end_bits;
    if (it->pos >= it->cur_n) {
        const int r = iter_advance(it);
        if (r <= 0) { *batch = nullptr; *n = 0; return r; }
    }
    *batch = reinterpret_cast<const daac_match8 *>(it->cur) + it->pos;
    *n = it->cur_n - it->pos;
    *end_base = it->cur_base;
    it->pos = it->cur_n;
    return 1;
}

int daac_iter_next_batch(daac_iter *it, const daac_match16 **batch, size_t *n) {
    if (!it || !batch || !n) return -DAAC_ERR_INVALID_ARGUMENT;
    if (it->compact) { set_error("daac_iter_next_batch serves iterators opened with daac_iter_open (this one is compact: daac_iter_next_batch8)"); return -DAAC_ERR_UNSUPPORTED; }
    if (it->pos >= it->cur_n) {
        const int r = iter_advance(it);
        if (r <= 0) { *batch = nullptr; *n = 0; return r; }
    }
    *batch = it->cur + it->pos;
    *n = it->cur_n - it->pos;
    it->pos = it->cur_n;
    return 1;
}

void daac_iter_close(daac_iter *it) {
    if (!it) return;
    {
        std::lock_guard<std::mutex> g(it->mu);
        it->stop = true;
    }
    it->cv.notify_all();
    if (it->worker.joinable()) it->worker.join();
    int prev = 0;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != it->device && hipSetDevice(it->device) == hipSuccess;
    for (IterWindow &w : it->win) {
        if (w.host) { if (w.host_pinned) pinned_pool().give(w.host, w.host_bytes); else std::free(w.host); }
        if (w.copied) (void)hipEventDestroy(w.copied);
    }
    if (it->kit) iter_kits().give(it->kit);   // (the worker, if it ever ran, has drained its streams)
    if (it->owned_dev) (void)hipFree(it->owned_dev);
    if (switched) (void)hipSetDevice(prev);
    delete it;
}

}  // extern "C"

// ------------------------------------------------------------------------------ chunk-fed steppers
struct daac_stream {
    daac_pma *pma;
    int mode, engine;
    hipStream_t stream;
    uint64_t consumed = 0;   // bytes fed so far
    uint64_t resume = 0;     // FIND: where the chain restarts (>= kept_from)
    uint64_t kept_from = 0;  // stream offset of byte 0 of `kept`
    void *kept = nullptr;    // device copy of stream bytes [kept_from, consumed)
    bool started = false;
    ~daac_stream() { if (kept) (void)hipFree(kept); }
};

// first character boundary at or after `pos` (charwise streams); the bytes are on the device
static daac_status boundary_at_or_after(const uint8_t *dev_virt, uint64_t pos, uint64_t end, hipStream_t stream, uint64_t *out) {
    uint8_t b[4] = {0, 0, 0, 0};
    const uint64_t n = std::min<uint64_t>(4, end > pos ? end - pos : 0);
    if (n) {
        HIP_TRY(hipMemcpyAsync(b, dev_virt + pos, n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    uint64_t k = 0;
    while (k < n && (b[k] & 0xc0u) == 0x80u) ++k;
    *out = pos + k;
    return DAAC_OK;
}

// end of the last complete character of [.., end) (charwise streams hold an incomplete tail back)
static daac_status last_complete_char_end(const uint8_t *dev_virt, uint64_t from, uint64_t end, hipStream_t stream, uint64_t *out) {
    const uint64_t n = std::min<uint64_t>(4, end - from);
    uint8_t b[4] = {0, 0, 0, 0};
    if (n) {
        HIP_TRY(hipMemcpyAsync(b, dev_virt + end - n, n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    *out = end;
    for (uint64_t back = 1; back <= n; ++back) {  // the last lead byte within 4 bytes of the end
        const uint8_t c = b[n - back];
        if ((c & 0xc0u) == 0x80u) continue;
        const uint64_t need = c < 0x80u ? 1 : c < 0xe0u ? 2 : c < 0xf0u ? 3 : 4;
        if (need > back) *out = end - back;       // that character is not complete yet
        break;
    }
    return DAAC_OK;
}

extern "C" {

daac_status daac_stream_open(daac_pma *pma, int mode, int engine, void *stream, daac_stream **out) {
    PmaScope scope_(pma);
    if (!pma || !out) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (mode == DAAC_LEFTMOST_FIND) {
        set_error("the reference has no stepper for leftmost automata (a leftmost match needs the text after it)");
        return DAAC_ERR_UNSUPPORTED;
    }
    daac_status st = check_mode_kind(pma, mode);
    if (st != DAAC_OK) return st;
    if (pma->charwise ? (engine != DAAC_ENGINE_AUTO && engine != DAAC_ENGINE_DARRAY)
                      : (engine != DAAC_ENGINE_AUTO && engine != DAAC_ENGINE_TIERED && engine != DAAC_ENGINE_DARRAY)) {
        set_error("engine cannot serve a stepper");
        return DAAC_ERR_UNSUPPORTED;
    }
    daac_stream *s = new daac_stream;
    s->pma = pma; s->mode = mode; s->engine = engine;
    s->stream = static_cast<hipStream_t>(stream);
    *out = s;
    return DAAC_OK;
}

daac_status daac_stream_feed(daac_stream *s, const uint8_t *chunk, size_t len, int chunk_is_device, daac_matches **out) {
    PmaScope scope_(s ? s->pma : nullptr);
    if (!s || !out || (len && !chunk)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::unique_ptr<daac_matches> m(new daac_matches);
    if (len == 0 && s->started) { *out = m.release(); return DAAC_OK; }
    DeviceTables *t = nullptr;
    daac_status st = get_tables(s->pma, &t);
    if (st != DAAC_OK) return st;
    const uint64_t halo = s->pma->halo();
    const bool find = s->mode == DAAC_FIND;
    const uint64_t total = s->consumed + len;
    // what of the old bytes the next scan can still look at: FIND restarts at `resume`, the overlapping scans
    // warm up over the halo
    const uint64_t keep = find ? s->resume : (s->consumed > halo ? s->consumed - halo : 0);
    void *fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, total - keep + 32));
    std::unique_ptr<void, void (*)(void *)> guard(fresh, [](void *p) { (void)hipFree(p); });
    uint8_t *nb = static_cast<uint8_t *>(fresh);
    if (s->consumed > keep)
        HIP_TRY(hipMemcpyAsync(nb, static_cast<const uint8_t *>(s->kept) + (keep - s->kept_from), s->consumed - keep, hipMemcpyDeviceToDevice, s->stream));
    if (len) HIP_TRY(hipMemcpyAsync(nb + (s->consumed - keep), chunk, len, chunk_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->stream));
    const uint8_t *virt = nb - keep;  // address stream byte 0 would have
    if (!find) {
        // FindOverlappingStepper / no-suffix: everything that ends inside this chunk (and ROOT's list at 0 on the first)
        if ((st = scan_range_materialize(s->pma, t, s->mode, s->engine, virt, s->consumed, total, total, s->stream, m->v, nullptr)) != DAAC_OK) return st;
    } else {
        // FindStepper: the chain goes on from `resume` as if the text ended here; what it has not decided yet is
        // re-read with the next chunk (a charwise stream also holds an incomplete last character back)
        uint64_t end = total;
        if (s->pma->charwise && (st = last_complete_char_end(virt, s->resume, total, s->stream, &end)) != DAAC_OK) return st;
        if (end < s->resume) end = s->resume;
        if ((st = scan_range_materialize(s->pma, t, s->mode, s->engine, virt, s->resume, end, end, s->stream, m->v, nullptr)) != DAAC_OK) return st;
        uint64_t r = m->v.size() ? m->v.p[m->v.size() - 1].end : s->resume;
        if (s->pma->root_has_output()) {
            r = end;  // "" among the patterns: one report per position, nothing is ever pending
        } else if (end > halo && r < end - halo) {
            // nothing matched for more than a halo: a fresh start `halo` bytes back reaches the same state
            r = end - halo;
            if (s->pma->charwise && (st = boundary_at_or_after(virt, r, end, s->stream, &r)) != DAAC_OK) return st;
        }
        s->resume = r;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->started) {
        // what ends at position 0 ("" among the patterns) was reported by the first call, even an empty one
        size_t skip = 0;
        while (skip < m->v.n && m->v.p[skip].end == 0) ++skip;
        if (skip) { std::memmove(m->v.p, m->v.p + skip, (m->v.n - skip) * sizeof(daac_match)); m->v.n -= skip; }
    }
    if (s->kept) (void)hipFree(s->kept);
    s->kept = guard.release();
    s->kept_from = keep;
    s->consumed = total;
    s->started = true;
    *out = m.release();
    return DAAC_OK;
}

void daac_stream_close(daac_stream *s) { delete s; }

}  // extern "C"

extern "C" {

// user-facing option name -> {field of Options (the name OPT() looks up), the process-wide atomic}; `value` is clamped where the option has a range
static bool option_slot(const std::string &n, int64_t &value, const char *&field, std::atomic<int64_t> *&slot) {
#define SLOT(NAME) if (n == #NAME) { field = #NAME; slot = &g_opt.NAME; return true; }
    SLOT(seg_bytes) SLOT(lds_budget) SLOT(dense_depth) SLOT(rows_share_pct) SLOT(blocks_per_cu) SLOT(threads) SLOT(iter_window) SLOT(max_result_bytes)
    SLOT(gram_lds_budget) SLOT(gram_region) SLOT(gram_slab) SLOT(gram_dense) SLOT(gram_ppl) SLOT(gram_version) SLOT(gram2_dpp) SLOT(gram2_rfull)
    SLOT(gram4_arith) SLOT(gram3_tail) SLOT(pfx) SLOT(pfx_probe) SLOT(find3) SLOT(left3) SLOT(select_emit) SLOT(emit) SLOT(emit_stagger) SLOT(emit_v3_lds)
    SLOT(emit_rec_per_kib) SLOT(gram_rank_in_lds) SLOT(restart_chain) SLOT(restart_bpc) SLOT(chain_rounds) SLOT(overlap_micro) SLOT(pool) SLOT(pool_keep)
    SLOT(char_map_lds) SLOT(char_row_lds)
#undef SLOT
    if (n == "gram_tail") { field = "gram3_tail"; slot = &g_opt.gram3_tail; return true; }
    if (n == "find3_window") { value = std::min<int64_t>(1ll << 30, std::max<int64_t>(8192, value)); field = "find3_window"; slot = &g_opt.find3_window; return true; }
    if (n == "workspace_keep") { value = std::max<int64_t>(0, value); field = "workspace_keep"; slot = &g_opt.workspace_keep; return true; }
    // options of engines that left the library (the TIERED chain walkers, the round-3 COUNT + WRITE emitter): accepted, nothing left to steer
    if (n == "restart_tier" || n == "emit_tiles" || n == "emit_rec_cap" || n == "emit_version") { field = nullptr; slot = nullptr; return true; }
    return false;
}

daac_status daac_set_option(const char *name, int64_t value) {
    if (!name) { set_error("null option name"); return DAAC_ERR_INVALID_ARGUMENT; }
    const char *field = nullptr;
    std::atomic<int64_t> *slot = nullptr;
    if (!option_slot(name, value, field, slot)) { set_error(std::string("unknown option: ") + name); return DAAC_ERR_INVALID_ARGUMENT; }
    if (slot) slot->store(value);
    return DAAC_OK;
}

// The same option for ONE handle: overrides the process-wide value for every scan, iterator and stream of `pma` (and, for the options read
// when the tables are laid out — gram_lds_budget, pfx, left3, lds_budget, char_map_lds ... — for its next daac_pma_upload).  `unset` != 0 takes
// the override away.  `pool` / `pool_keep` (the device's allocator) have no per-handle meaning: status 1.
daac_status daac_pma_set_option(daac_pma *pma, const char *name, int64_t value, int unset) {
    if (!pma || !name) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    const char *field = nullptr;
    std::atomic<int64_t> *slot = nullptr;
    if (!option_slot(name, value, field, slot)) { set_error(std::string("unknown option: ") + name); return DAAC_ERR_INVALID_ARGUMENT; }
    if (slot == &g_opt.pool || slot == &g_opt.pool_keep) { set_error("pool / pool_keep are properties of the device's allocator, not of a handle"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (!field) return DAAC_OK;
    std::lock_guard<std::mutex> g(pma->opt_mu);
    if (unset) pma->opt_ov.erase(field); else pma->opt_ov[field] = value;
    pma->opt_n.store(static_cast<int>(pma->opt_ov.size()));
    return DAAC_OK;
}

}  // extern "C"
