// GRAM engine, second table set (host side) — see gram2.hpp.
#include "gram2.hpp"

#include <algorithm>
#include <map>

namespace daac {

namespace {
uint64_t ipow(uint64_t b, uint32_t e) {
    uint64_t r = 1;
    while (e--) r *= b;
    return r;
}
constexpr uint32_t kNone = 0xffffffffu;
}  // namespace

bool build_gram2_tables(const HostPma &p, uint32_t lds_budget, Gram2Tables &out) {
    out = Gram2Tables{};
    if (!p.is_standard()) return false;
    const uint32_t n = static_cast<uint32_t>(p.states.size());
    if (n == 0 || n >= (1u << 27)) return false;
    // "" as a pattern makes every position (and end = 0) a match: left to the AC engines
    if (output_pos_of(p.states[kRoot].opos_ch) != 0) return false;

    // ---- breadth-first renumbering over the double array: children of a state are contiguous, byte-ascending ----
    std::vector<uint32_t> new_of_old(n, kNone), old_of_new{kRoot}, depth{0}, first_child, nchild;
    std::vector<uint8_t> label{0};
    new_of_old[kRoot] = 0;
    bool used[256] = {false};
    for (uint32_t s = 0; s < old_of_new.size(); ++s) {
        const uint32_t base = p.states[old_of_new[s]].base;
        first_child.push_back(static_cast<uint32_t>(old_of_new.size()));
        uint32_t cnt = 0;
        if (base != 0) {
            for (uint32_t c = 0; c < 256; ++c) {
                const uint32_t t = base ^ c;
                if (t >= n || t == kRoot || check_of(p.states[t].opos_ch) != c) continue;
                if (new_of_old[t] != kNone) return false;  // not a tree
                new_of_old[t] = static_cast<uint32_t>(old_of_new.size());
                old_of_new.push_back(t);
                depth.push_back(depth[s] + 1);
                label.push_back(static_cast<uint8_t>(c));
                used[c] = true;
                ++cnt;
            }
        }
        nchild.push_back(cnt);
    }
    const uint32_t N = static_cast<uint32_t>(old_of_new.size());

    // ---- byte classes (ascending byte order, so that class order = child order) ----
    out.cls.assign(256, 0);
    uint32_t C = 1;
    uint8_t rep[32] = {0};
    int unused = -1;
    for (uint32_t c = 0; c < 256; ++c) {
        if (used[c]) {
            if (C >= 30) return false;  // bits 1..29 of an M word are continuation bits
            rep[C] = static_cast<uint8_t>(c);
            out.cls[c] = static_cast<uint8_t>(C++);
        } else if (unused < 0) {
            unused = static_cast<int>(c);
        }
    }
    if (unused < 0 || C < 2) return false;
    rep[0] = static_cast<uint8_t>(unused);

    // ---- per state: child bitmap, own patterns (list entries as long as the state is deep), class string ----
    std::vector<uint32_t> cmap(N, 0), own_cnt(N, 0), own_hs(N, 0);
    std::vector<uint64_t> gram(N, 0);
    for (uint32_t s = 0; s < N; ++s) {
        for (uint32_t j = 0; j < nchild[s]; ++j) {
            const uint32_t ch = first_child[s] + j, k = out.cls[label[ch]];
            cmap[s] |= 1u << k;
            gram[ch] = depth[ch] <= 6 ? gram[s] * C + k : 0;
        }
        uint32_t op = output_pos_of(p.states[old_of_new[s]].opos_ch);
        while (op != 0 && p.outputs[op - 1].length == depth[s]) {
            own_cnt[s]++;
            own_hs[s] += match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            op = p.outputs[op - 1].parent;
        }
    }

    // output-list aggregates per output record (the `parent` chain always points to a smaller index)
    std::vector<OutSum> osum(p.outputs.size());
    for (size_t i = 0; i < p.outputs.size(); ++i) {
        const OutputRec &o = p.outputs[i];
        OutSum s{1u, match_hash32(o.value, o.length)};
        if (o.parent != 0) { s.cnt += osum[o.parent - 1].cnt; s.hsum += osum[o.parent - 1].hsum; }
        osum[i] = s;
    }

    auto pad16 = [](uint64_t x) { return static_cast<uint32_t>((x + 15) & ~15ull); };
    for (uint32_t K : {3u, 2u}) {
        const uint64_t ngram = ipow(C, K);
        if (ngram * 4 >= (1ull << 17)) continue;  // the queue entry keeps the byte offset of the M word in 17 bits
        // depth-(K+1) states: regular own patterns, lexicographic numbering (rank == offset within the level)
        bool ok = true;
        uint32_t n_deep = 0, level_start = N;
        uint64_t prev = 0;
        for (uint32_t s = 0; s < N && ok; ++s) {
            if (depth[s] != K + 1) continue;
            if (own_cnt[s] != (own_hs[s] != 0 ? 1u : 0u)) ok = false;
            if (n_deep == 0) level_start = s; else if (gram[s] <= prev) ok = false;
            prev = gram[s];
            ++n_deep;
        }
        if (!ok) continue;
        const uint32_t nm = static_cast<uint32_t>((ngram + 3) & ~3ull);
        std::vector<uint32_t> m(nm, 0);
        std::vector<uint16_t> cid4(nm, 0);
        std::vector<uint32_t> hs{0};
        std::map<uint32_t, uint32_t> id_of;  // hsum -> id
        bool exact_ok = true, has_short = false;
        for (uint32_t g = 0; g < ngram && ok; ++g) {
            uint32_t st = kRoot;
            for (uint32_t i = 0; i < K; ++i) st = p.next_state(st, rep[(g / static_cast<uint32_t>(ipow(C, K - 1 - i))) % C]);
            const uint32_t op = output_pos_of(p.states[st].opos_ch);
            if (op == 0) continue;
            const OutSum o = osum[op - 1];
            if (o.cnt > 3) { ok = false; break; }  // two count bits per word
            has_short = true;
            m[g] |= o.cnt << 30;
            if (o.hsum != 0) {  // a zero sum needs no table entry (id 0 adds nothing)
                auto it = id_of.find(o.hsum);
                if (it == id_of.end()) {
                    it = id_of.emplace(o.hsum, static_cast<uint32_t>(hs.size())).first;
                    hs.push_back(o.hsum);
                }
                if (it->second >= 16384) exact_ok = false; else cid4[g] = static_cast<uint16_t>(4 * it->second);
            }
        }
        if (!ok) continue;
        for (uint32_t s = level_start; s < level_start + n_deep; ++s) {
            const uint32_t g = static_cast<uint32_t>(gram[s] / C), d = static_cast<uint32_t>(gram[s] % C);
            m[g] |= 1u << d;  // d >= 1: class 0 labels no edge
        }
        std::vector<uint32_t> sdir(nm / 4, 0);
        uint32_t run = 0;
        for (uint32_t g = 0; g < nm; ++g) {
            if ((g & 3) == 0) sdir[g >> 2] = run;
            run += static_cast<uint32_t>(__builtin_popcount(m[g] & kGram2MaskBits));
        }
        if (run != n_deep) continue;  // cannot happen for a consistent trie
        const bool s16 = n_deep < 65536;
        const uint64_t lds_count = kGram2OffMHost + pad16(static_cast<uint64_t>(nm) * 4) + pad16(static_cast<uint64_t>(nm / 4) * (s16 ? 2 : 4));
        const uint64_t lds_exact = lds_count + pad16(static_cast<uint64_t>(nm) * 2) + pad16(hs.size() * 4);
        if (lds_count > lds_budget) continue;
        out.K = K;
        out.m = std::move(m);
        out.sdir = std::move(sdir);
        out.cid4 = std::move(cid4);
        out.hsum = std::move(hs);
        out.s16 = s16;
        out.has_short = has_short;
        out.level_start = level_start;
        out.lds_count = static_cast<uint32_t>(lds_count);
        out.lds_exact = static_cast<uint32_t>(lds_exact);
        out.exact_available = exact_ok && lds_exact <= lds_budget;
        out.dhit.clear();
        out.dhit_c.clear();
        out.cfirst.clear();
        for (uint32_t s = level_start; s < level_start + n_deep; ++s) {
            out.dhit.push_back(U32x2{cmap[s], own_hs[s]});
            out.dhit_c.push_back(U32x2{cmap[s] | (own_cnt[s] ? 1u : 0u), first_child[s]});  // (class 0 labels no edge: bit 0 is free)
            out.cfirst.push_back(first_child[s]);
        }
        break;
    }
    if (out.K == 0) return false;
    out.C = C;
    out.N = N;
    out.unused_byte = rep[0];
    out.drec.resize(N);
    for (uint32_t s = 0; s < N; ++s) out.drec[s] = U32x4{cmap[s], first_child[s], own_cnt[s], own_hs[s]};
    {   // tail records: children carry larger numbers than their parents, so one pass from the back knows every subtree
        out.drec_c = out.drec;
        std::vector<uint8_t> path_len(N, 0xff);  // edges of the single path below s; 0xff: the subtree branches (or is too long)
        for (uint32_t s = N; s-- > 0;) {
            const uint32_t kids = static_cast<uint32_t>(__builtin_popcount(cmap[s] & kGram2MaskBits));
            if (own_cnt[s] > 1) continue;
            if (kids == 0) { path_len[s] = 0; continue; }
            if (kids != 1) continue;
            const uint32_t c = first_child[s];
            if (path_len[c] == 0xff || path_len[c] >= 8) continue;
            path_len[s] = static_cast<uint8_t>(path_len[c] + 1);
        }
        out.drec_t = out.drec;
        out.dhit_t.clear();
        for (uint32_t s = out.level_start; s < out.level_start + out.dhit_c.size(); ++s) out.dhit_t.push_back(U32x4{out.dhit_c[s - out.level_start].x, out.dhit_c[s - out.level_start].y, 0u, 0u});
        for (uint32_t s = 0; s < N; ++s) {
            if (depth[s] < out.K + 1 || path_len[s] == 0xff || path_len[s] == 0) continue;
            uint64_t bytes = 0;
            uint32_t ends = own_cnt[s] ? 1u : 0u, cur = s;
            const uint32_t first_class = static_cast<uint32_t>(__builtin_ctz(cmap[s] & kGram2MaskBits));
            for (uint32_t i = 0; i < path_len[s]; ++i) {
                const uint32_t d = static_cast<uint32_t>(__builtin_ctz(cmap[cur] & kGram2MaskBits));
                bytes |= static_cast<uint64_t>(rep[d]) << (8 * i);
                cur = first_child[cur];
                if (own_cnt[cur]) ends |= 2u << i;
            }
            const U32x4 tail{0x80000000u | path_len[s] | (ends << 4) | (first_class << 13), 0u, static_cast<uint32_t>(bytes), static_cast<uint32_t>(bytes >> 32)};
            if (depth[s] == out.K + 1) out.dhit_t[s - out.level_start] = tail;
            else out.drec_t[s] = tail;
            // (`.count()` of gram2_kernels.hip: not at depth K + 2, where its walkers start — most of them end there, and that
            // record stays as cheap as it was)
            if (depth[s] >= out.K + 3) out.drec_c[s] = tail;
        }
        // the count + checksum walkers' records (round 6, as gram.cpp): from depth K + 3 on, a single path with exactly ONE pattern end is
        // {h of that pattern, 1 << 31 | edges | index of the ending node << 4, path bytes 0-3, path bytes 4-7} (first_child < 2^27: bit 31 of the
        // second word marks the form).  After drec_c / drec_t were taken from the plain records.
        for (uint32_t s = 0; s < N; ++s) {
            if (depth[s] < out.K + 3 || path_len[s] == 0xff || path_len[s] == 0) continue;
            uint64_t bytes = 0;
            uint32_t cur = s, n_end = own_cnt[s] ? 1u : 0u, at = 0, h = own_cnt[s] ? own_hs[s] : 0u;
            for (uint32_t i = 0; i < path_len[s]; ++i) {
                const uint32_t d = static_cast<uint32_t>(__builtin_ctz(cmap[cur] & kGram2MaskBits));
                bytes |= static_cast<uint64_t>(rep[d]) << (8 * i);
                cur = first_child[cur];
                if (own_cnt[cur]) { ++n_end; at = i + 1; h = own_hs[cur]; }
            }
            if (n_end != 1) continue;
            out.drec[s] = U32x4{h, 0x80000000u | path_len[s] | (at << 4), static_cast<uint32_t>(bytes), static_cast<uint32_t>(bytes >> 32)};
        }
    }
    out.available = true;

    // ---- tuple emission tables ----
    {
        const uint32_t K = out.K;
        uint32_t max_len = 0;
        bool ok = C <= 29;
        for (const OutputRec &o : p.outputs) max_len = std::max(max_len, o.length);
        for (uint32_t s = 0; s < N && ok; ++s) ok = own_cnt[s] <= (depth[s] <= K ? 1u : 256u);  // short duplicates: two values per flag bit
        if (max_len >= (1u << 24)) ok = false;
        out.max_len = max_len;
        if (ok) {
            const uint32_t ngram = static_cast<uint32_t>(ipow(C, K));
            out.me.assign(out.m.size(), 0);
            out.v1.assign(C, 0);
            out.v2.assign(static_cast<size_t>(C) * C, 0);
            if (K == 3) out.v3.assign(static_cast<size_t>(C) * C * C, 0);
            for (uint32_t g = 0; g < ngram; ++g) {
                uint32_t st = kRoot;
                for (uint32_t i = 0; i < K; ++i) st = p.next_state(st, rep[(g / static_cast<uint32_t>(ipow(C, K - 1 - i))) % C]);
                uint32_t word = out.m[g] & 0x1ffffffeu;  // continuation bits 1..28
                for (uint32_t op = output_pos_of(p.states[st].opos_ch); op != 0; op = p.outputs[op - 1].parent) {
                    const uint32_t len = p.outputs[op - 1].length, val = p.outputs[op - 1].value;
                    word |= 1u << (28 + len);  // len in 1..K
                    const uint32_t sub = g % static_cast<uint32_t>(ipow(C, len));  // the last `len` classes of the context
                    (len == 1 ? out.v1 : len == 2 ? out.v2 : out.v3)[sub] = val;
                }
                out.me[g] = word;
            }
            out.erec.resize(N);
            out.dupo.assign(N, 0);
            out.dupv.clear();
            for (uint32_t s = 0; s < N; ++s) {
                uint32_t val = 0;
                uint32_t op = output_pos_of(p.states[old_of_new[s]].opos_ch);
                if (own_cnt[s]) val = p.outputs[op - 1].value;
                if (own_cnt[s] > 1) {  // the further copies, in the order the iterator reports them (nfa_builder.rs:203-222)
                    out.dupo[s] = static_cast<uint32_t>(out.dupv.size());
                    for (uint32_t k = 1; k < own_cnt[s]; ++k) {
                        op = p.outputs[op - 1].parent;
                        out.dupv.push_back(p.outputs[op - 1].value);
                    }
                }
                out.erec[s] = U32x4{cmap[s] | (own_cnt[s] ? 1u : 0u), first_child[s], val, depth[s] | ((own_cnt[s] > 1 ? own_cnt[s] - 1 : 0u) << 24)};
            }
            out.ehit.clear();
            out.ecopies.clear();
            for (uint32_t s = out.level_start; s < out.level_start + out.dhit.size(); ++s) {
                out.ehit.push_back(U32x2{out.erec[s].x, out.erec[s].z});
                out.ecopies.push_back(out.erec[s].w >> 24);
            }
            if (out.dupv.empty()) out.dupv.push_back(0);
            out.emit_available = true;
        }
    }
    return true;
}

}  // namespace daac
