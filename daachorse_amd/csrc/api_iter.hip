// C ABI (include/daachorse_amd.h), part 4: the lazy iterator (daac_iter_*) and the chunk-fed steppers (daac_stream_*).
#include "api_internal.hpp"

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
                    (void)launch_repack16(dm.p, d16, dm.n, s_scan);
                    dev_free(dm.release_keep_n(), s_scan);
                    dm.p = static_cast<daac_match *>(d16);
                }
            }
            if (st == DAAC_OK && compact && dm.n != 0) {   // ends relative to the window's first byte, the length above them: 8 bytes per tuple over the link
                void *d8 = nullptr;
                if (next_begin - begin >= (1ull << end_bits)) { set_error("iterator: a window ran past what its compact form can say"); st = DAAC_ERR_UNSUPPORTED; }
                else if (dev_malloc(&d8, dm.n * 8, s_scan) != hipSuccess) { st = hip_fail(hipGetLastError(), "iterator: repack buffer"); }
                else {
                    (void)launch_repack8(dm.p, d8, dm.n, begin, end_bits, s_scan);
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
// A feed scans [kept bytes | chunk].  The two device buffers that hold it (ping-pong: the kept tail of one is copied to the front of the other)
// and the page-locked block the compact tuples come back in belong to the stream object and grow geometrically: a feed allocates nothing
// once they are large enough.  (Until round 6 every feed did a hipMalloc of kept + chunk bytes, a synchronise and a hipFree of the last
// feed's buffer: 0.65 ms per 64 MiB chunk of sparse text, of which the scan was a quarter.)
struct daac_stream {
    daac_pma *pma;
    int mode, engine;
    hipStream_t stream;
    uint64_t consumed = 0;   // bytes fed so far
    uint64_t resume = 0;     // FIND: where the chain restarts (>= kept_from)
    uint64_t kept_from = 0;  // stream offset of byte 0 of the current buffer's text
    void *buf[2] = {nullptr, nullptr};   // device: text of stream bytes [kept_from, consumed) starts at buf[cur] + skew
    size_t cap[2] = {0, 0};
    int cur = 0;
    int device = -1;
    uint32_t end_bits = 0;   // compact tuples: 32 - bits of the longest pattern's length (0: not worked out yet)
    void *pin8 = nullptr;    // page-locked: the last compact feed's tuples
    size_t pin8_bytes = 0;
    bool pin8_pinned = false;
    hipStream_t s2 = nullptr;   // compact feeds: the second half of the tuples' copy back runs beside the first (one copy engine alone stays
    hipEvent_t ev_ready = nullptr, ev_half = nullptr;   // at ~40 GB/s of the link's ~55: profiles/r04_iterator.txt)
    bool started = false;
    ~daac_stream() {
        if (s2) (void)hipStreamDestroy(s2);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_half) (void)hipEventDestroy(ev_half);
        for (void *b : buf) if (b) (void)hipFree(b);
        if (pin8) { if (pin8_pinned) (void)hipHostFree(pin8); else std::free(pin8); }
    }
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

// One feed.  Exactly one of `out` (24-byte tuples in a host list of its own) and `batch` (8-byte tuples in the stream's page-locked block)
// is asked for.
static daac_status stream_feed_impl(daac_stream *s, const uint8_t *chunk, size_t len, int chunk_is_device, daac_matches **out,
                                    const daac_match8 **batch, size_t *n_out, uint64_t *end_base, uint32_t *end_bits_out) {
    PmaScope scope_(s ? s->pma : nullptr);
    const bool compact = batch != nullptr;
    if (!s || (!out && !compact) || (compact && (!n_out || !end_base || !end_bits_out)) || (len && !chunk)) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    std::unique_ptr<daac_matches> m(compact ? nullptr : new daac_matches);
    if (compact) {
        if (s->end_bits == 0) {   // length bits above the end, as daac_iter_open_compact
            uint32_t lb = 1;
            while ((1ull << lb) <= s->pma->max_pattern_len()) ++lb;
            s->end_bits = 32 - lb;
        }
        if ((1ull << s->end_bits) < 4ull * s->pma->halo() + (1ull << 20)) {
            set_error("patterns of " + std::to_string(s->pma->max_pattern_len()) + " bytes leave no room for a chunk in the compact tuple; use daac_stream_feed");
            return DAAC_ERR_UNSUPPORTED;
        }
        *batch = nullptr; *n_out = 0; *end_base = s->consumed; *end_bits_out = s->end_bits;
    }
    if (len == 0 && s->started) { if (out) *out = m.release(); return DAAC_OK; }
    DeviceTables *t = nullptr;
    daac_status st = get_tables(s->pma, &t);
    if (st != DAAC_OK) return st;
    const uint64_t halo = s->pma->halo();
    const bool find = s->mode == DAAC_FIND;
    const uint64_t total = s->consumed + len;
    // what of the old bytes the next scan can still look at: FIND restarts at `resume`, the overlapping scans warm up over the halo
    const uint64_t keep = find ? s->resume : (s->consumed > halo ? s->consumed - halo : 0);
    const uint64_t scan_from = find ? s->resume : s->consumed;
    if (compact && total - std::min(keep, scan_from) >= (1ull << s->end_bits)) {
        set_error("daac_stream_feed_compact: a chunk (plus what is kept of the earlier ones) must stay below 2^end_bits bytes — feed smaller chunks, or use daac_stream_feed");
        return DAAC_ERR_UNSUPPORTED;
    }
    // the other buffer takes [kept bytes | chunk], keeping the stream's 16-byte phase (the kernels read whole aligned granules)
    const int nx = s->cur ^ 1;
    const uint64_t skew = keep & 15;
    const size_t need = static_cast<size_t>(total - keep + skew + 64);
    if (s->cap[nx] < need) {
        if (s->buf[nx]) (void)hipFree(s->buf[nx]);
        s->buf[nx] = nullptr; s->cap[nx] = 0;
        const size_t want = std::max(need + need / 4, s->cap[s->cur]);
        HIP_TRY(hipMalloc(&s->buf[nx], want));
        s->cap[nx] = want;
    }
    uint8_t *nb = static_cast<uint8_t *>(s->buf[nx]) + skew;
    if (s->consumed > keep) {
        const uint8_t *old = static_cast<const uint8_t *>(s->buf[s->cur]) + (s->kept_from & 15) + (keep - s->kept_from);
        HIP_TRY(hipMemcpyAsync(nb, old, s->consumed - keep, hipMemcpyDeviceToDevice, s->stream));
    }
    if (len) HIP_TRY(hipMemcpyAsync(nb + (s->consumed - keep), chunk, len, chunk_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->stream));
    const uint8_t *virt = nb - keep;  // address stream byte 0 would have
    uint64_t end = total;
    if (find) {
        // FindStepper: the chain goes on from `resume` as if the text ended here; what it has not decided yet is
        // re-read with the next chunk (a charwise stream also holds an incomplete last character back)
        if (s->pma->charwise && (st = last_complete_char_end(virt, s->resume, total, s->stream, &end)) != DAAC_OK) return st;
        if (end < s->resume) end = s->resume;
    }
    uint64_t last_end = scan_from;   // FIND: end of the last match reported (the chain's position)
    bool any = false;
    if (!compact) {
        // FindOverlappingStepper / no-suffix: everything that ends inside this chunk (and ROOT's list at 0 on the first)
        if ((st = scan_range_materialize(s->pma, t, s->mode, s->engine, virt, scan_from, end, end, s->stream, m->v, nullptr)) != DAAC_OK) return st;
        if (m->v.size()) { any = true; last_end = m->v.p[m->v.size() - 1].end; }
    } else {
        DevMatches dm;
        dm.f16 = true;
        if ((st = scan_range_device(s->pma, t, s->mode, s->engine, virt, scan_from, end, end, s->stream, dm, nullptr)) != DAAC_OK) return st;
        if (dm.n != 0 && !dm.f16_done) {   // an engine that writes daac_match: repacked on the device
            void *d16 = nullptr;
            HIP_TRY(dev_malloc(&d16, dm.n * 16, s->stream));
            (void)launch_repack16(dm.p, d16, dm.n, s->stream);
            dev_free(dm.release_keep_n(), s->stream);
            dm.p = static_cast<daac_match *>(d16);
        }
        if (dm.n != 0) {
            const size_t bytes = dm.n * 8;
            if (s->pin8_bytes < bytes) {
                if (s->pin8) { if (s->pin8_pinned) (void)hipHostFree(s->pin8); else std::free(s->pin8); }
                s->pin8 = nullptr; s->pin8_bytes = 0;
                const size_t want = bytes + bytes / 2 + 4096;
                void *q = nullptr;
                if (hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess) s->pin8_pinned = true;
                else { (void)hipGetLastError(); q = std::malloc(want); s->pin8_pinned = false; }
                if (!q) { set_error("out of host memory for the match list"); return DAAC_ERR_AUTOMATON_SCALE; }
                s->pin8 = q; s->pin8_bytes = want;
            }
            void *d8 = nullptr;
            HIP_TRY(dev_malloc(&d8, bytes, s->stream));
            hipError_t e = launch_repack8(dm.p, d8, dm.n, scan_from, s->end_bits, s->stream);
            // two halves on two streams once the list is MBs long and the tuples are going to page-locked memory
            const size_t half = (bytes >= (8u << 20) && s->pin8_pinned) ? (bytes / 2) & ~size_t(4095) : bytes;
            if (e == hipSuccess && half != bytes && !s->s2) {
                if (hipStreamCreateWithFlags(&s->s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&s->ev_half, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    if (s->s2) { (void)hipStreamDestroy(s->s2); s->s2 = nullptr; }
                }
            }
            const bool split = half != bytes && s->s2 && s->ev_ready && s->ev_half;
            if (e == hipSuccess && split) {
                e = hipEventRecord(s->ev_ready, s->stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(s->s2, s->ev_ready, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(static_cast<char *>(s->pin8) + half, static_cast<const char *>(d8) + half, bytes - half, hipMemcpyDeviceToHost, s->s2);
                if (e == hipSuccess) e = hipEventRecord(s->ev_half, s->s2);
                if (e == hipSuccess) e = hipMemcpyAsync(s->pin8, d8, half, hipMemcpyDeviceToHost, s->stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(s->stream, s->ev_half, 0);
            } else if (e == hipSuccess) {
                e = hipMemcpyAsync(s->pin8, d8, bytes, hipMemcpyDeviceToHost, s->stream);
            }
            dev_free(d8, s->stream);
            HIP_TRY(e);
            HIP_TRY(hipStreamSynchronize(s->stream));
            const daac_match8 *p8 = static_cast<const daac_match8 *>(s->pin8);
            any = true;
            last_end = scan_from + (p8[dm.n - 1].end_len & ((1u << s->end_bits) - 1u));
            size_t skip = 0;
            if (s->started)   // what ends at position 0 ("" among the patterns) was reported by the first call, even an empty one
                while (skip < dm.n && scan_from + (p8[skip].end_len & ((1u << s->end_bits) - 1u)) == 0) ++skip;
            *batch = p8 + skip;
            *n_out = static_cast<size_t>(dm.n) - skip;
            *end_base = scan_from;
        }
    }
    if (find) {
        uint64_t r = any ? last_end : s->resume;
        if (s->pma->root_has_output()) {
            r = end;  // "" among the patterns: one report per position, nothing is ever pending
        } else if (end > halo && r < end - halo) {
            // nothing matched for more than a halo: a fresh start `halo` bytes back reaches the same state
            r = end - halo;
            if (s->pma->charwise && (st = boundary_at_or_after(virt, r, end, s->stream, &r)) != DAAC_OK) return st;
        }
        s->resume = r;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));   // (the chunk has been read: the caller may reuse it)
    if (!compact && s->started) {
        // what ends at position 0 ("" among the patterns) was reported by the first call, even an empty one
        size_t skip = 0;
        while (skip < m->v.n && m->v.p[skip].end == 0) ++skip;
        if (skip) { std::memmove(m->v.p, m->v.p + skip, (m->v.n - skip) * sizeof(daac_match)); m->v.n -= skip; }
    }
    s->cur = nx;
    s->kept_from = keep;
    s->consumed = total;
    s->started = true;
    if (out) *out = m.release();
    return DAAC_OK;
}

daac_status daac_stream_feed(daac_stream *s, const uint8_t *chunk, size_t len, int chunk_is_device, daac_matches **out) {
    if (!out) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    return stream_feed_impl(s, chunk, len, chunk_is_device, out, nullptr, nullptr, nullptr, nullptr);
}

// The same feed with the chunk's matches as 8-byte tuples (daac_match8, as daac_iter_next_batch8 hands them out) in a page-locked block of
// the stream object, valid until the next feed or close: a third of the bytes of daac_match over PCIe, and no list to allocate and free.
daac_status daac_stream_feed_compact(daac_stream *s, const uint8_t *chunk, size_t len, int chunk_is_device, const daac_match8 **batch,
                                     size_t *n, uint64_t *end_base, uint32_t *end_bits) {
    if (!batch) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    return stream_feed_impl(s, chunk, len, chunk_is_device, nullptr, batch, n, end_base, end_bits);
}

void daac_stream_close(daac_stream *s) { delete s; }

}  // extern "C"
