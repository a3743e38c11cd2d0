// C ABI (include/daachorse_amd.h), part 2: the scan drivers — plans, the chain walkers' passes, ranged scans that leave their tuple list on the
// device or hand it to the host, count (+ checksum) incl. shards and several devices.  No CPU scan fallback lives here.
#include "api_internal.hpp"

static daac_status scan_count_impl(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin, int hay_is_device,
                                   void *stream_, uint64_t *count, uint64_t *checksum, uint64_t *result_dev, bool want_checksum);

namespace {

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

}  // namespace

namespace daac {
namespace api {

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

hipError_t launch(const DeviceTables *t, const Plan &pl, int kmode, bool heads, hipStream_t s, unsigned long long *next_begin) {
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

// a few page-locked words per host thread for flags read back between passes
unsigned int *pinned_words() {
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

hipError_t launch_repack16(const daac_match *in, void *out, unsigned long long n, hipStream_t stream) {
    hipLaunchKernelGGL(repack16_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>(65535, (n + 255) / 256))), dim3(256), 0, stream, in, static_cast<uint4 *>(out), n);
    return hipGetLastError();
}
hipError_t launch_repack8(const void *in, void *out, unsigned long long n, unsigned long long base, uint32_t end_bits, hipStream_t stream) {
    hipLaunchKernelGGL(repack8_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>(65535, (n + 255) / 256))), dim3(256), 0, stream, static_cast<const uint4 *>(in),
                       static_cast<uint2 *>(out), n, base, end_bits);
    return hipGetLastError();
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


// Which engine counts an overlapping scan of `span` bytes (scan_count_impl runs what this says; fill_plan reports it).  GRAM: the second
// table set where it applies (count only: always; with the checksum: when CID / H fit next to M), else the first, else the wide one; PFX:
// what AUTO takes where no GRAM table set applies (before the text has been probed: PFX is a filter, and dense text goes to the walkers).
// (measured on cfg3: with the checksum both table sets spend three LDS lookups per position and the first is a little faster; `.count()`
// alone needs one lookup per position on the second and runs 20-25 % faster there)
CountRoute count_route(const daac_pma *pma, const DeviceTables *t, int mode, int engine, bool want_checksum, uint64_t span) {
    CountRoute r{};
    const int64_t gv = OPT(gram_version);
    r.g1_can = t->gram_ok && gv != 2;
    r.g2_can = t->gram2_ok && (!want_checksum || t->gram2.exact_ok) && gv != 1 && !(gv == 0 && want_checksum && r.g1_can);
    r.gw_can = t->gramw_ok && (!want_checksum || t->gramw.exact_ok);  // wide alphabets: built only where the others are not
    const bool applies = !pma->charwise && mode == DAAC_FIND_OVERLAPPING && pma->host.is_standard() && span < (1ull << 35);
    r.gram = applies && (engine == DAAC_ENGINE_GRAM || (engine == DAAC_ENGINE_AUTO && (r.g2_can || r.g1_can || r.gw_can)));
    r.pfx = applies && t->pfx_ok && (engine == DAAC_ENGINE_PFX || (engine == DAAC_ENGINE_AUTO && !r.gram));
    return r;
}

}  // namespace api
}  // namespace daac

extern "C" {

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
    // which engine and which GRAM table set serve this request (count_route: the decision daac_pma_info's plan reports as well)
    const int64_t gv = OPT(gram_version);
    const CountRoute route = count_route(pma, t, mode, engine, want_checksum, len - begin);
    const bool g1_can = route.g1_can, g2_can = route.g2_can, gw_can = route.gw_can, use_gram = route.gram;
    bool use_pfx = route.pfx;
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
        const bool want_rfull = OPT(gram2_rfull) != 0, want_arith = OPT(gram4_arith) != 0, want_filter = OPT(gram4_filter) != 0;
        const uint32_t waves = static_cast<uint32_t>(OPT(threads)) > 512 ? 16u : 8u;
        const int64_t ppl_opt = OPT(gram_ppl);
        // preference (profiles/r05_gram4_decomposition.txt: p32 rfull 1 420, p32 coarse 1 378, p16 rfull 1 326 GB/s): 32 positions per lane
        // first, then the per-word directory (two LDS reads per hit instead of five); eight waves leave the tables more room where sixteen
        // do not fit.  gram4_plan hands back exactly the shape asked for or nothing.
        struct Shape { uint32_t ppl; bool rfull; } shapes[4] = {{32u, true}, {32u, false}, {16u, true}, {16u, false}};
        bool planned = false;
        for (uint32_t w = waves; w >= 8u && !planned; w >>= 1) {
            for (const Shape &sh : shapes) {
                if ((ppl_opt == 16 || ppl_opt == 32) && sh.ppl != static_cast<uint32_t>(ppl_opt)) continue;
                if (sh.rfull && !want_rfull) continue;
                if (gram4_plan(t->gram4, sh.ppl, w, sh.rfull, want_arith, want_filter, 160u * 1024u, g4l)) { g4_ppl = sh.ppl; planned = true; break; }
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
    if (!find3_served) {   // what daac_last_kernel() says: the kernel family and, for the `.count()` kernel, the launch shape the options gave it
        if (use_g4) {
            const int64_t tail_opt = OPT(gram_tail);
            g_last_kernel = "gram4 ppl=" + std::to_string(g4_ppl) + " dir=" + std::to_string(g4l.dir) + " waves=" + std::to_string(g4l.threads / 64) + " arith=" + std::to_string(g4l.arith) +
                            " filter=" + std::to_string(g4l.filter) + " tail=" + (tail_opt < 0 ? std::string("auto") : std::to_string(tail_opt > 0 ? 1 : 0));
        } else {
            g_last_kernel = use_pfx ? "pfx" : use_gw ? "gram2w" : use_g2 ? "gram2" : use_gram ? "gram" : pl.charwise ? "charwise" : pl.tier ? "tiered" : "darray";
        }
    } else {
        g_last_kernel = "find3";
    }
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
        const int64_t tail_opt = OPT(gram_tail);
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

}  // extern "C"

// ---- the shard workers of daac_scan_count_multi (api_internal.hpp: ShardWorker)
ShardWorker::ShardWorker(int dev, const ::daac_pma *p) : device(dev), pma(p) {
    th = std::thread([this]() {
        PmaScope scope_(pma);
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) {
            (void)hipGetLastError();
            stream = nullptr;
            std::lock_guard<std::mutex> g(mu);
            failed = true;
        }
        for (;;) {
            std::function<void(ShardWorker &)> job;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [this]() { return stop || !jobs.empty(); });
                if (jobs.empty()) break;   // (stop: what was posted before it is still run)
                job = std::move(jobs.front());
                jobs.pop_front();
            }
            job(*this);
        }
        if (d_res) (void)hipFree(d_res);
        if (h_res) (void)hipHostFree(h_res);
        if (stream) (void)hipStreamDestroy(stream);
    });
}
ShardWorker::~ShardWorker() {
    {
        std::lock_guard<std::mutex> g(mu);
        stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
}
void ShardWorker::post(std::function<void(ShardWorker &)> job) {
    {
        std::lock_guard<std::mutex> g(mu);
        jobs.push_back(std::move(job));
    }
    cv.notify_one();
}
bool ShardWorker::reserve(size_t shards) {
    if (shards <= res_cap) return true;
    if (d_res) (void)hipFree(d_res);
    if (h_res) (void)hipHostFree(h_res);
    d_res = h_res = nullptr;
    res_cap = 0;
    const size_t cap = std::max<size_t>(shards, 16), bytes = cap * 3 * sizeof(unsigned long long);
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (d) (void)hipFree(d);
        return false;
    }
    d_res = static_cast<unsigned long long *>(d);
    h_res = static_cast<unsigned long long *>(h);
    res_cap = cap;
    return true;
}

extern "C" {

// One haystack sharded across the devices of a node (BASELINE configs[3]): the product's own form of what bench.py does with
// one process per GPU.  One persistent host thread per DEVICE the shards name (ShardWorker: created at the handle's first call that names
// the device, with a stream of its own and the handle's options in scope): it uploads the tables there if they are not yet, queues
// daac_scan_count[_only]_range over [halo | shard] with begin = halo for every shard of its device back to back — results stay in device
// memory, nothing is waited for between shards —, synchronises once and hands {count, S1, S2} per shard back; the caller adds the counts
// and the two checksum sums — `base` re-bases a shard's ends (S2 += low32(base - halo) * S1).  No collective: RCCL is for callers that
// run one process per GPU (daachorse_amd/dist.py) and reduce {count, S1, S2} themselves.
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
        if (shards[k].device < 0 || shards[k].device >= ndev || shards[k].device >= kMaxDevices || ((shards[k].len + shards[k].halo) && !shards[k].hay)) {
            set_error("daac_scan_count_multi: bad shard (device ordinal / null haystack)");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
        if (shards[k].halo > shards[k].base) {   // (the ends are re-based by base - halo)
            set_error("daac_scan_count_multi: a shard's halo reaches in front of the haystack (halo > base)");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
        if (shards[k].halo < halo_need && shards[k].halo < shards[k].base) {   // fewer bytes in front than a match may reach back, and not the haystack's start
            set_error("daac_scan_count_multi: a shard needs max_pattern_len - 1 bytes of the haystack in front of it (charwise: max_pattern_len, at least 3)");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
    }
    struct Out { daac_status st = DAAC_OK; uint64_t count = 0; uint32_t s1 = 0, s2 = 0; int engine = DAAC_ENGINE_AUTO; std::string err, kernel; };
    std::vector<Out> outs(n);
    // the shards of each device, in the caller's order
    std::map<int, std::vector<size_t>> by_dev;
    for (size_t k = 0; k < n; ++k) by_dev[shards[k].device].push_back(k);
    struct Latch { std::mutex mu; std::condition_variable cv; size_t left = 0; } latch;
    latch.left = by_dev.size();
    const bool want_checksum = checksum != nullptr;
    std::vector<ShardWorker *> posted;
    std::string spawn_err;
    for (auto &kv : by_dev) {
        ShardWorker *w = nullptr;
        try {
            std::lock_guard<std::mutex> g(pma->workers_mu);
            std::unique_ptr<ShardWorker> &slot = pma->workers[kv.first];
            if (!slot) slot.reset(new ShardWorker(kv.first, pma));
            w = slot.get();
        } catch (const std::exception &e) {   // (std::system_error from std::thread, bad_alloc: nothing may leave an extern "C" function)
            spawn_err = e.what();
        }
        if (!w) {
            for (size_t k : kv.second) { outs[k].st = DAAC_ERR_DEVICE; outs[k].err = "no worker thread for the device: " + spawn_err; }
            std::lock_guard<std::mutex> g(latch.mu);
            --latch.left;
            continue;
        }
        const std::vector<size_t> *mine = &kv.second;
        try {
        w->post([&, mine](ShardWorker &me) {
            const size_t m = mine->size();
            bool ok = !me.failed && me.stream != nullptr && me.reserve(m);
            if (!ok) {
                for (size_t k : *mine) { outs[k].st = DAAC_ERR_DEVICE; outs[k].err = "the shard's device could not be made current (hipSetDevice / stream / result buffers)"; }
            } else {
                size_t queued = 0;
                for (size_t i = 0; i < m; ++i) {
                    const size_t k = (*mine)[i];
                    const daac_shard &sh = shards[k];
                    Out &o = outs[k];
                    uint64_t *rd = reinterpret_cast<uint64_t *>(me.d_res + 3 * i);
                    o.st = scan_count_impl(pma, mode, engine, sh.hay, sh.halo + sh.len, sh.halo, hay_is_device, me.stream, nullptr, nullptr, rd, want_checksum);
                    o.engine = g_last_engine;
                    o.kernel = g_last_kernel;
                    if (o.st != DAAC_OK) o.err = daac_last_error(); else ++queued;
                }
                hipError_t e = hipSuccess;
                if (queued) e = hipMemcpyAsync(me.h_res, me.d_res, m * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, me.stream);
                if (e == hipSuccess) e = hipStreamSynchronize(me.stream);
                for (size_t i = 0; i < m; ++i) {
                    Out &o = outs[(*mine)[i]];
                    if (o.st != DAAC_OK) continue;
                    if (e != hipSuccess) { o.st = DAAC_ERR_DEVICE; o.err = std::string("shard scan: ") + hipGetErrorString(e); continue; }
                    o.count = me.h_res[3 * i];
                    o.s1 = static_cast<uint32_t>(me.h_res[3 * i + 1]);
                    o.s2 = static_cast<uint32_t>(me.h_res[3 * i + 2]);
                }
                if (e != hipSuccess) (void)hipGetLastError();
            }
            std::lock_guard<std::mutex> g(latch.mu);
            if (--latch.left == 0) latch.cv.notify_all();
        });
        } catch (const std::exception &e) {   // (the job could not be queued: nothing of it runs)
            for (size_t k : kv.second) { outs[k].st = DAAC_ERR_DEVICE; outs[k].err = std::string("the shard's job could not be queued: ") + e.what(); }
            std::lock_guard<std::mutex> g(latch.mu);
            --latch.left;
        }
    }
    {
        std::unique_lock<std::mutex> g(latch.mu);
        latch.cv.wait(g, [&]() { return latch.left == 0; });
    }
    uint64_t total = 0;
    uint32_t s1 = 0, s2 = 0;
    for (size_t k = 0; k < n; ++k) {
        if (outs[k].st != DAAC_OK) { set_error("shard " + std::to_string(k) + " (device " + std::to_string(shards[k].device) + "): " + outs[k].err); return outs[k].st; }
        total += outs[k].count;
        const uint32_t shift = static_cast<uint32_t>(shards[k].base - shards[k].halo);   // ends were counted from the shard's first resident byte
        s1 += outs[k].s1;
        s2 += outs[k].s2 + shift * outs[k].s1;
    }
    if (n) { g_last_engine = outs[0].engine; g_last_kernel = outs[0].kernel; }
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
        HIP_TRY(launch_repack16(dm.p, d16, dm.n, stream));
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
