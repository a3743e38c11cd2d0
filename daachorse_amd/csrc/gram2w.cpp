// GRAM engine for wide alphabets (host side) — see gram2w.hpp.
#include "gram2w.hpp"

#include <map>

namespace daac {

bool build_gram2w_tables(const HostPma &p, uint32_t lds_budget, Gram2WTables &out) {
    out = Gram2WTables{};
    constexpr uint32_t kNone = 0xffffffffu, K = 2;
    if (!p.is_standard()) return false;
    const uint32_t n = static_cast<uint32_t>(p.states.size());
    if (n == 0 || n >= (1u << 26)) return false;  // a walker entry keeps the state in 26 bits
    if (output_pos_of(p.states[kRoot].opos_ch) != 0) return false;

    // breadth-first renumbering: children contiguous, byte-ascending
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

    out.cls.assign(256, 0);
    uint32_t C = 1;
    uint8_t rep[64] = {0};
    int unused = -1;
    for (uint32_t c = 0; c < 256; ++c) {
        if (used[c]) {
            if (C >= 62) return false;  // bits 1..61 are continuation bits
            rep[C] = static_cast<uint8_t>(c);
            out.cls[c] = static_cast<uint8_t>(C++);
        } else if (unused < 0) {
            unused = static_cast<int>(c);
        }
    }
    if (unused < 0 || C < 2) return false;
    rep[0] = static_cast<uint8_t>(unused);

    std::vector<uint64_t> cmap(N, 0), gram(N, 0);
    std::vector<uint32_t> own_cnt(N, 0), own_hs(N, 0);
    for (uint32_t s = 0; s < N; ++s) {
        for (uint32_t j = 0; j < nchild[s]; ++j) {
            const uint32_t ch = first_child[s] + j, k = out.cls[label[ch]];
            cmap[s] |= 1ull << k;
            gram[ch] = depth[ch] <= 4 ? gram[s] * C + k : 0;
        }
        uint32_t op = output_pos_of(p.states[old_of_new[s]].opos_ch);
        while (op != 0 && p.outputs[op - 1].length == depth[s]) {
            own_cnt[s]++;
            own_hs[s] += match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            op = p.outputs[op - 1].parent;
        }
        if (own_cnt[s] != (own_hs[s] != 0 ? 1u : 0u)) return false;  // the records carry no count
    }
    std::vector<OutSum> osum(p.outputs.size());
    for (size_t i = 0; i < p.outputs.size(); ++i) {
        const OutputRec &o = p.outputs[i];
        OutSum s{1u, match_hash32(o.value, o.length)};
        if (o.parent != 0) { s.cnt += osum[o.parent - 1].cnt; s.hsum += osum[o.parent - 1].hsum; }
        osum[i] = s;
    }

    // depth-3 states: lexicographic numbering (rank == offset within the level)
    uint32_t n_deep = 0, level_start = N;
    uint64_t prev = 0;
    for (uint32_t s = 0; s < N; ++s) {
        if (depth[s] != K + 1) continue;
        if (n_deep == 0) level_start = s; else if (gram[s] <= prev) return false;
        prev = gram[s];
        ++n_deep;
    }
    const uint32_t ngram = C * C, nm = (ngram + 3) & ~3u;
    out.m.assign(nm, 0);
    out.cid4.assign(nm, 0);
    out.hsum.assign(1, 0);
    std::map<uint32_t, uint32_t> id_of;
    bool exact_ok = true;
    for (uint32_t g = 0; g < ngram; ++g) {
        uint32_t st = p.next_state(kRoot, rep[g / C]);
        st = p.next_state(st, rep[g % C]);
        const uint32_t op = output_pos_of(p.states[st].opos_ch);
        if (op == 0) continue;
        const OutSum o = osum[op - 1];
        if (o.cnt > 3) return false;
        out.m[g] |= static_cast<uint64_t>(o.cnt) << 62;
        if (o.hsum != 0) {
            auto it = id_of.find(o.hsum);
            if (it == id_of.end()) {
                it = id_of.emplace(o.hsum, static_cast<uint32_t>(out.hsum.size())).first;
                out.hsum.push_back(o.hsum);
            }
            if (it->second >= 16000) exact_ok = false; else out.cid4[g] = static_cast<uint16_t>(4 * it->second);
        }
    }
    for (uint32_t s = level_start; s < level_start + n_deep; ++s) out.m[gram[s] / C] |= 1ull << (gram[s] % C);
    out.sdir.assign(nm / 4, 0);
    uint32_t run = 0;
    for (uint32_t g = 0; g < nm; ++g) {
        if ((g & 3) == 0) out.sdir[g >> 2] = run;
        run += static_cast<uint32_t>(__builtin_popcountll(out.m[g] & kGram2WMaskBits));
    }
    if (run != n_deep) return false;
    auto pad16 = [](uint64_t x) { return static_cast<uint32_t>((x + 15) & ~15ull); };
    out.lds_count = 512 + pad16(static_cast<uint64_t>(nm) * 8) + pad16(static_cast<uint64_t>(nm / 4) * 4);
    out.lds_exact = out.lds_count + pad16(static_cast<uint64_t>(nm) * 2) + pad16(out.hsum.size() * 4);
    if (out.lds_count > lds_budget) return false;
    out.exact_available = exact_ok && out.lds_exact <= lds_budget;
    out.C = C;
    out.N = N;
    out.level_start = level_start;
    out.unused_byte = rep[0];
    out.drec.resize(N);
    for (uint32_t s = 0; s < N; ++s)
        out.drec[s] = U32x4{static_cast<uint32_t>(cmap[s]), static_cast<uint32_t>(cmap[s] >> 32), first_child[s], own_hs[s]};
    for (uint32_t s = level_start; s < level_start + n_deep; ++s)
        out.dhit.push_back(U32x4{static_cast<uint32_t>(cmap[s]), static_cast<uint32_t>(cmap[s] >> 32), own_hs[s], first_child[s]});
    out.available = true;
    return true;
}

}  // namespace daac
