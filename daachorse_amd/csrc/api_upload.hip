// C ABI (include/daachorse_amd.h), part 1: handles — construction, serialisation, the upload of the re-packed automaton and of every
// engine's tables to a device, daac_pma_info / _explain / _trim.  No CPU scan fallback lives here.
#include "api_internal.hpp"

namespace {

// The patterns a bytewise automaton of either kind was built from, read back from its trie (goto edges of the double array, bytewise.rs:1070-1077)
// with their values: a state's own pattern is the output as long as the state is deep.  (LeftmostFirst: the builder never inserted what lies
// below an earlier-registered pattern, nfa_builder.rs:60-66 — what is read back is what can be reported.)  false: not a tree / "" / too large.
bool recover_patterns(const HostPma &p, std::vector<uint8_t> &blob, std::vector<uint64_t> &offs, std::vector<uint32_t> &vals) {
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
std::vector<U32x4> zip_first_child(const std::vector<U32x2> &hit, const std::vector<uint32_t> &first) {
    std::vector<U32x4> out(hit.size());
    for (size_t i = 0; i < hit.size(); ++i) out[i] = U32x4{hit[i].x, hit[i].y, first[i], 0u};
    return out;
}

}  // namespace

namespace daac {
namespace api {

// --------------------------------------------------------------------------------------- upload
daac_status upload_locked(daac_pma *pma, int device, DeviceTables **out) {
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
                // the filter in front of rank + gather (gram4_filter.hpp): as large as the preferred launch shape leaves room for, less a margin
                {
                    const uint32_t room = gram4_filter_room(q.m_bytes, q.s_bytes, g4.arith, 160u * 1024u);
                    if (room > 1024u && build_gram4_filter(g4, room - 512u)) {
                        if ((st = t->put(g4.bloom, q.bloom)) != DAAC_OK) return st;
                        q.bloom_words = static_cast<uint32_t>(g4.bloom.size());
                    }
                }
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

daac_status get_tables(daac_pma *pma, DeviceTables **out) {
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> g(pma->mu);
    return upload_locked(pma, device, out);
}

}  // namespace api
}  // namespace daac

// ============================================================================== exported C ABI
extern "C" {

const char *daac_last_error(void) { return last_error_cstr(); }
int daac_last_engine(void) { return g_last_engine; }
const char *daac_last_kernel(void) { return g_last_kernel.c_str(); }
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
    // count / count + checksum: what scan_count_impl will run (the same decision function)
    for (const bool cs : {false, true}) {
        const int req = cs ? DAAC_REQ_OVERLAPPING_CHECKSUM : DAAC_REQ_OVERLAPPING_COUNT;
        const CountRoute cr = count_route(pma, t, DAAC_FIND_OVERLAPPING, DAAC_ENGINE_AUTO, cs, 0);
        const int why = (cs && (t->gram2_ok || t->gramw_ok)) ? DAAC_WHY_LDS : why_no_gram;   // (tables there, their checksum half without room)
        if (cr.gram) set(req, DAAC_ENGINE_GRAM, cr.g2_can ? (cs ? DAAC_KERNEL_GRAM_EXACT : DAAC_KERNEL_GRAM_COUNT) : cr.g1_can ? DAAC_KERNEL_GRAM_EXACT : DAAC_KERNEL_GRAM_WIDE, DAAC_WHY_FASTEST);
        else if (cr.pfx) set(req, DAAC_ENGINE_PFX, DAAC_KERNEL_PFX, why);
        else set(req, micro_engine, micro, why);
    }
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

}  // extern "C"
