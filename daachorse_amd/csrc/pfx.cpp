// PFX engine (host side) — see pfx.hpp.
#include "pfx.hpp"

#include <algorithm>
#include <numeric>

namespace daac {

namespace {
constexpr uint32_t kNone = 0xffffffffu;
struct Key { uint32_t k0, k1, state; };
}  // namespace

bool build_pfx_tables(const HostPma &p, uint32_t lds_budget, PfxTables &out) {
    out = PfxTables{};
    if (!p.is_standard()) return false;
    const uint32_t n = static_cast<uint32_t>(p.states.size());
    if (n == 0 || output_pos_of(p.states[kRoot].opos_ch) != 0) return false;  // "" is a pattern: left to the AC engines

    // ---- breadth-first walk over the double array: depth and first bytes of every state ----
    std::vector<uint32_t> depth(n, kNone), order{kRoot}, k0(n, 0), k1(n, 0), parent(n, kNone);
    std::vector<uint8_t> label(n, 0);
    depth[kRoot] = 0;
    bool seen_byte[256] = {false};
    for (size_t qi = 0; qi < order.size(); ++qi) {
        const uint32_t s = order[qi], base = p.states[s].base;
        if (base == 0) continue;
        for (uint32_t c = 0; c < 256; ++c) {
            const uint32_t t = base ^ c;
            if (t >= n || t == kRoot || check_of(p.states[t].opos_ch) != c) continue;
            if (depth[t] != kNone) return false;  // not a tree
            depth[t] = depth[s] + 1;
            parent[t] = s;
            label[t] = static_cast<uint8_t>(c);
            seen_byte[c] = true;
            k0[t] = k0[s]; k1[t] = k1[s];
            if (depth[s] < 4) k0[t] |= c << (8 * depth[s]); else if (depth[s] < 8) k1[t] |= c << (8 * (depth[s] - 4));
            order.push_back(t);
        }
    }
    for (bool b : seen_byte) out.n_distinct_bytes += b ? 1u : 0u;
    // ---- patterns that END in a state (its own, not those of its suffixes): list entries as long as the state is deep ----
    std::vector<uint32_t> own(n, 0), own_hs(n, 0), own_val(n, 0);
    uint32_t min_len2 = kNone;
    bool has_len1 = false, no_dups = true;
    for (const uint32_t s : order) {
        uint32_t op = output_pos_of(p.states[s].opos_ch);
        while (op != 0 && p.outputs[op - 1].length == depth[s]) {
            if (own[s] == 0) own_val[s] = p.outputs[op - 1].value;
            ++own[s];
            own_hs[s] += match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            op = p.outputs[op - 1].parent;
        }
        if (own[s] > 1) no_dups = false;
        if (own[s] >= (1u << 24)) return false;
        if (own[s] != 0) {
            if (depth[s] == 1) has_len1 = true; else min_len2 = std::min(min_len2, depth[s]);
        }
    }
    if (min_len2 == kNone) return false;  // one-byte patterns only: nothing for a prefix filter to do
    const uint32_t G = std::min<uint32_t>(min_len2, 6);
    out.G = G;
    out.has_len1 = has_len1;

    std::vector<Key> keys;
    for (const uint32_t s : order)
        if (depth[s] == G) keys.push_back(Key{k0[s], k1[s], s});
    if (keys.empty()) return false;
    const uint32_t nk = static_cast<uint32_t>(keys.size());
    out.n_keys = nk;

    // ---- single paths: children come after their parents in `order`, so one pass from the back knows every subtree ----
    std::vector<uint8_t> path_len(n, 0xff), nchild(n, 0);  // edges of the single path below a state; 0xff: it branches / is too long / has duplicates
    std::vector<uint32_t> only_child(n, kNone);
    for (size_t qi = order.size(); qi-- > 1;) { const uint32_t t = order[qi]; if (nchild[parent[t]] < 255) ++nchild[parent[t]]; only_child[parent[t]] = t; }
    for (size_t qi = order.size(); qi-- > 0;) {
        const uint32_t s = order[qi];
        if (own[s] > 1) continue;
        if (nchild[s] == 0) { path_len[s] = 0; continue; }
        if (nchild[s] != 1) continue;
        const uint32_t c = only_child[s];
        if (path_len[c] == 0xff || path_len[c] >= 8) continue;
        path_len[s] = static_cast<uint8_t>(path_len[c] + 1);
    }

    // ---- sizes: DISP with eight keys per bucket, SLOTS at a load of 0.6, BLOOM takes the LDS that is left (at most 64 bits per key) ----
    out.buckets = std::max<uint32_t>(16, (nk + 7) / 8);
    out.n_slots = std::max<uint32_t>(64, static_cast<uint32_t>((static_cast<uint64_t>(nk) * 5 + 2) / 3));
    const uint32_t disp_bytes = (out.buckets * 2 + 15) & ~15u;
    if (disp_bytes + 512 + 4096 > lds_budget) return false;
    uint32_t bloom_words = ((lds_budget - disp_bytes - 512) / 4) & ~3u;
    bloom_words = std::min<uint32_t>(bloom_words, std::max<uint32_t>(256, ((nk * 2 + 3) & ~3u)));
    out.bloom_words = bloom_words;
    out.lds_tables = bloom_words * 4 + disp_bytes + 512;

    // ---- BLOOM: word by range reduction of m, two bits from its middle ----
    out.bloom.assign(bloom_words, 0);
    for (const Key &k : keys) {
        const uint32_t m = k.k0 * kPfxMulBloom0 + k.k1 * kPfxMulBloom1;
        const uint32_t m2 = m * kPfxMulBits;
        out.bloom[static_cast<uint32_t>((static_cast<uint64_t>(m) * bloom_words) >> 32)] |= (1u << ((m2 >> kPfxBit1) & 31u)) | (1u << ((m2 >> kPfxBit2) & 31u));
    }
    // ---- CNT1 ----
    out.cnt1.assign(256, 0);
    out.hs1.assign(256, 0);
    out.v1.assign(256, 0);
    out.has1.assign(256, 0);
    for (const uint32_t s : order)
        if (depth[s] == 1) {
            if (own[s] > 0xffff) return false;
            out.cnt1[k0[s] & 0xffu] = static_cast<uint16_t>(own[s]);
            out.hs1[k0[s] & 0xffu] = own_hs[s];
            out.v1[k0[s] & 0xffu] = own_val[s];
            out.has1[k0[s] & 0xffu] = own[s] ? 0x20 : 0;
        }

    // ---- hash and displace ----
    const uint32_t M = out.n_slots;
    bool placed = false;
    for (uint32_t seed = 0; seed < 16 && !placed; ++seed) {
        std::vector<std::vector<uint32_t>> bucket(out.buckets);
        for (uint32_t i = 0; i < nk; ++i) {
            const uint32_t mb = keys[i].k0 * kPfxMulBucket0 + (keys[i].k1 ^ seed) * kPfxMulBucket1;
            bucket[static_cast<uint32_t>((static_cast<uint64_t>(mb) * out.buckets) >> 32)].push_back(i);
        }
        std::vector<uint32_t> by_size(out.buckets);
        std::iota(by_size.begin(), by_size.end(), 0u);
        std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) { return bucket[a].size() > bucket[b].size(); });
        std::vector<uint8_t> used(M, 0);
        std::vector<uint16_t> disp(out.buckets, 0);
        std::vector<uint32_t> slot_of(nk, 0);
        bool ok = true;
        for (const uint32_t b : by_size) {
            const auto &ks = bucket[b];
            if (ks.empty()) break;
            std::vector<uint32_t> ms(ks.size());
            for (size_t j = 0; j < ks.size(); ++j) ms[j] = keys[ks[j]].k0 * kPfxMulSlot0 + (keys[ks[j]].k1 ^ seed) * kPfxMulSlot1;
            auto at = [&](size_t j, uint32_t d) { return pfx_slot(ms[j], d, M); };
            bool found = false;
            for (uint32_t d = 0; d < 65536 && !found; ++d) {
                bool fits = true;
                for (size_t j = 0; j < ks.size() && fits; ++j) {
                    const uint32_t sl = at(j, d);
                    if (used[sl]) fits = false;
                    for (size_t j2 = 0; j2 < j && fits; ++j2) fits = at(j2, d) != sl;
                }
                if (!fits) continue;
                for (size_t j = 0; j < ks.size(); ++j) { const uint32_t sl = at(j, d); used[sl] = 1; slot_of[ks[j]] = sl; }
                disp[b] = static_cast<uint16_t>(d);
                found = true;
            }
            if (!found) { ok = false; break; }
        }
        if (!ok) continue;
        out.seed = seed;
        out.disp = std::move(disp);
        out.slots.assign(M, U32x4{0u, kPfxEmpty, 0u, 0u});
        out.slots_x.assign(M, U32x4{0u, kPfxEmpty, 0u, 0u});
        out.slots_e.assign(M, U32x4{0u, kPfxEmpty, 0u, 0u});
        for (uint32_t i = 0; i < nk; ++i) {
            const uint32_t s = keys[i].state;
            if (own[s] >= (1u << 14)) return false;
            out.slots_x[slot_of[i]] = U32x4{keys[i].k0, keys[i].k1 | (own[s] << 16), p.states[s].base, own_hs[s]};
            out.slots_e[slot_of[i]] = U32x4{keys[i].k0, keys[i].k1 | (own[s] << 16), p.states[s].base, own_val[s]};
            if (path_len[s] != 0xff && path_len[s] != 0) {  // one path below the key: the record carries it
                uint64_t bytes = 0;
                uint32_t ends = own[s] ? 1u : 0u, cur = s;
                for (uint32_t e = 0; e < path_len[s]; ++e) {
                    cur = only_child[cur];
                    bytes |= static_cast<uint64_t>(label[cur]) << (8 * e);
                    if (own[cur]) ends |= 2u << e;
                }
                out.slots[slot_of[i]] = U32x4{keys[i].k0, keys[i].k1 | kPfxTail | (static_cast<uint32_t>(path_len[s]) << 16) | (ends << 20),
                                              static_cast<uint32_t>(bytes), static_cast<uint32_t>(bytes >> 32)};
                ++out.n_tails;
            } else {
                if (own[s] >= (1u << 14)) return false;
                uint32_t filter = 0;
                for (uint32_t c0 = 0; c0 < 256 && p.states[s].base != 0; ++c0) {
                    const uint32_t t = p.states[s].base ^ c0;
                    if (t >= n || depth[t] != G + 1 || parent[t] != s) continue;
                    if (own[t] != 0) filter = 0xffffffffu;  // a pattern ends one byte below the key: whatever follows, the branch counts
                    for (uint32_t c1 = 0; c1 < 256 && p.states[t].base != 0; ++c1) {
                        const uint32_t u = p.states[t].base ^ c1;
                        if (u < n && depth[u] == G + 2 && parent[u] == t) filter |= 1u << pfx_pair_bit(c0 | (c1 << 8));
                    }
                }
                out.slots[slot_of[i]] = U32x4{keys[i].k0, keys[i].k1 | (own[s] << 16), p.states[s].base, filter};
            }
        }
        placed = true;
    }
    if (!placed) return false;

    // ---- WREC ----
    out.wrec.resize(n);
    out.wrec_x.resize(n);
    for (uint32_t s = 0; s < n; ++s) {
        // vacant slots keep a CHECK no transition can produce (builder.rs:391-400): they are copied as they are, with nothing ending there
        const uint32_t o = depth[s] != kNone ? own[s] : 0u;
        out.wrec[s] = U32x2{p.states[s].base, static_cast<uint32_t>(check_of(p.states[s].opos_ch)) | (o << 8)};
        out.wrec_x[s] = U32x4{p.states[s].base, static_cast<uint32_t>(check_of(p.states[s].opos_ch)) | (o << 8), depth[s] != kNone ? own_hs[s] : 0u,
                              depth[s] != kNone ? own_val[s] : 0u};
    }
    out.emit_ok = no_dups;
    if (!no_dups) { out.slots_e.clear(); out.slots_e.shrink_to_fit(); }
    out.available = true;
    return true;
}

}  // namespace daac
