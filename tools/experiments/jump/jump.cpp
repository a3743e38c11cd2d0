// JUMP engine (host side) — see jump.hpp.
#include "jump.hpp"

#include <algorithm>

namespace daac {

namespace {
constexpr uint32_t kNone = 0xffffffffu;
}

bool build_jump_tables(const HostPma &p, JumpTables &out) {
    out = JumpTables{};
    if (!p.is_standard()) return false;
    const uint32_t n = static_cast<uint32_t>(p.states.size());
    if (n == 0 || n >= (1u << 27)) return false;
    if (output_pos_of(p.states[kRoot].opos_ch) != 0) return false;  // "" as a pattern: FindIterator degenerates (iter.rs:60-85)
    uint32_t max_len = 0;
    for (const OutputRec &o : p.outputs) max_len = std::max(max_len, o.length);
    if (max_len == 0 || max_len > kJumpMaxLen) return false;

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
    int unused = -1;
    for (uint32_t c = 0; c < 256; ++c) {
        if (used[c]) {
            if (C >= 30) return false;  // bits 1..29 of an MS word are continuation bits
            out.cls[c] = static_cast<uint8_t>(C++);
        } else if (unused < 0) {
            unused = static_cast<int>(c);
        }
    }
    if (unused < 0 || C < 2) return false;
    constexpr uint32_t K = 3;

    // ---- per state: child bitmap, the state's own first-registered pattern ----
    std::vector<uint32_t> cmap(N, 0), own(N, 0), own_h(N, 0);
    for (uint32_t s = 0; s < N; ++s) {
        for (uint32_t j = 0; j < nchild[s]; ++j) cmap[s] |= 1u << out.cls[label[first_child[s] + j]];
        const uint32_t op = output_pos_of(p.states[old_of_new[s]].opos_ch);
        if (op != 0 && p.outputs[op - 1].length == depth[s]) {  // the head of a state's list is its own pattern if it has one
            own[s] = 1;
            own_h[s] = match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
        }
    }
    auto child_of = [&](uint32_t s, uint32_t k) -> uint32_t {  // class k >= 1
        if (!((cmap[s] >> k) & 1u)) return kNone;
        return first_child[s] + static_cast<uint32_t>(__builtin_popcount(cmap[s] & ((1u << k) - 2u)));
    };

    // ---- MS, H1..H3: walk every K-gram of classes down from ROOT ----
    const uint32_t ngram = C * C * C;
    const uint32_t nm = (ngram + 3u) & ~3u;
    out.ms.assign(nm, 0);
    out.h1.assign(C, 0);
    out.h2.assign(static_cast<size_t>(C) * C, 0);
    out.h3.assign(ngram, 0);
    std::vector<uint32_t> deep_of(ngram, kNone);  // the depth-K state of a gram whose continuation bits are set
    for (uint32_t g = 0; g < ngram; ++g) {
        const uint32_t k[3] = {g / (C * C), (g / C) % C, g % C};
        uint32_t s = 0, shortest = 0;
        for (uint32_t i = 0; i < K; ++i) {
            if (k[i] == 0) { s = kNone; break; }
            s = child_of(s, k[i]);
            if (s == kNone) break;
            if (own[s]) {
                if (i == 0) out.h1[k[0]] = own_h[s];
                else if (i == 1) out.h2[k[0] * C + k[1]] = own_h[s];
                else out.h3[g] = own_h[s];
                if (shortest == 0) shortest = i + 1;
            }
        }
        uint32_t word = shortest << 30;
        if (shortest == 0 && s != kNone) {
            word |= cmap[s] & 0x3ffffffeu;
            if (cmap[s] != 0) deep_of[g] = s;
        }
        out.ms[g] = word;
    }
    out.sdir.assign(nm / 4, 0);
    out.jhit.clear();
    uint32_t run = 0;
    for (uint32_t g = 0; g < nm; ++g) {
        if ((g & 3) == 0) out.sdir[g >> 2] = run;
        const uint32_t bits = out.ms[g] & 0x3ffffffeu;
        if (bits == 0) continue;
        const uint32_t s = deep_of[g];
        for (uint32_t d = 1; d < C; ++d) {
            if (!((bits >> d) & 1u)) continue;
            const uint32_t ch = child_of(s, d);
            out.jhit.push_back(U32x4{(cmap[ch] & 0x3ffffffeu) | own[ch], first_child[ch], own_h[ch], depth[ch]});
            ++run;
        }
    }
    if (out.jhit.empty()) out.jhit.push_back(U32x4{0, 0, 0, 0});
    out.jrec.resize(N);
    for (uint32_t s = 0; s < N; ++s) out.jrec[s] = U32x4{(cmap[s] & 0x3ffffffeu) | own[s], first_child[s], own_h[s], depth[s]};

    out.K = K;
    out.C = C;
    out.N = N;
    out.unused_byte = static_cast<uint8_t>(unused);
    out.max_len = max_len;
    out.lds_bytes = kJumpOffMS + nm * 4u + (nm / 4u) * 4u;
    out.available = true;
    return true;
}

}  // namespace daac
