// C ABI (include/daachorse_amd.h), part 3: the drivers of the one-detection tuple emitter (emit3_kernels.hip, PFX's) and of the restart
// iterators' selection kernels (find3_kernels.hip / left3_kernels.hip): windows, scratch, give-ups and retries.
#include "api_internal.hpp"

namespace daac {
namespace api {

// FindOverlappingIterator of a bytewise Standard automaton through the one-detection tuple emitter (emit3_kernels.hip):
// DETECT (annotated class stream, tile counts, deep-match records) -> scans of the tile counts -> BIN (records by tile) -> EXPAND.
// *served = false when the automaton / request does not qualify or the haystack is of the adversarial kind the kernels give up on
// (then nothing is returned and the other engines take over).
// `dest`: the tuples go to this place (room for dest_cap of them) instead of a buffer of the call's own: out.n says how many, out.p stays null.
// `raw`: the PFX engine's tuples (any byte alphabet): pfx_emit_kernel logs every match of two or more bytes as a record, EXPAND runs over the
// haystack itself (one-byte patterns by table) — same glue, same record list, no annotated stream.
daac_status emit_overlapping3(daac_pma *pma, DeviceTables *t, const uint8_t *dev_hay, uint64_t begin, uint64_t end, hipStream_t stream,
                              DevMatches &out, bool *served, bool raw, void *dest, uint64_t dest_cap) {
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
        a.v3_in_lds = (!raw && e.v3c != nullptr &&
                       emit3_expand_lds_bytes(e, xwaves, out.f16, true) <= (160u * 1024u) / (out.f16 ? 2u : 3u)) ? 1u : 0u;
        a.off_wave = e.v1_bytes + e.v2_bytes + (a.v3_in_lds ? e.v3c_bytes : 0u);
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
}  // namespace api
}  // namespace daac
