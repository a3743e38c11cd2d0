// GPU re-pack of a daachorse automaton (host side) — see repack.hpp for the layouts.
#include "repack.hpp"

#include <algorithm>

namespace daac {

static inline uint64_t mix64(uint64_t z) {  // SplitMix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

uint32_t match_hash32(uint32_t value, uint32_t length) {
    return static_cast<uint32_t>(mix64((static_cast<uint64_t>(value) << 32) | length));
}

// Chain aggregates by dynamic programming over `parent` (always a smaller index:
// bytewise.rs:955-962 is enforced by validate()).
static void build_chain_sums(const HostPma &p, std::vector<OutSum> &osum) {
    osum.resize(p.outputs.size());
    for (size_t i = 0; i < p.outputs.size(); ++i) {
        const OutputRec &o = p.outputs[i];
        OutSum s{1u, match_hash32(o.value, o.length)};
        if (o.parent != 0) {
            s.cnt += osum[o.parent - 1].cnt;
            s.hsum += osum[o.parent - 1].hsum;
        }
        osum[i] = s;
    }
}

void build_darray_tables(const HostPma &p, DArrayTables &out) {
    const size_t n = p.states_len();
    out.hot.resize(n);
    out.fail.resize(n);
    for (size_t i = 0; i < n; ++i) {
        out.hot[i] = U32x2{p.base(static_cast<uint32_t>(i)), p.opos_ch(static_cast<uint32_t>(i))};
        out.fail[i] = p.fail(static_cast<uint32_t>(i));
    }
    // Dense root row: child index plus the child's own hot record, so a step out of ROOT costs
    // one LDS read and no HBM read (the reference's root_table, bytewise.rs:57-59, plus the record).
    out.root.assign(256, U32x4{0, 0, 0, 0});
    const uint32_t rbase = n ? p.base(kRoot) : 0;
    for (uint32_t c = 0; c < 256; ++c) {
        uint32_t child = 0;
        if (rbase != 0) {
            const uint32_t t = rbase ^ c;
            if (t < n && check_of(p.opos_ch(t)) == c) child = t;
        }
        out.root[c] = U32x4{child, n ? p.base(child) : 0, n ? p.opos_ch(child) : 0, (n && child) ? p.fail(child) : 0};
    }
    build_chain_sums(p, out.osum);

    // child filters: every base belongs to one state, so the parent of an element t is the owner of base t ^ check(t)
    // (vacant elements carry a CHECK no transition can land on, reference src/bytewise/builder.rs:391-400: no owner)
    out.fmap.assign(n, 0);
    {
        std::vector<uint32_t> owner(n, 0xffffffffu);
        for (size_t s = 0; s < n; ++s) {
            const uint32_t b = p.base(static_cast<uint32_t>(s));
            if (b != 0 && b < n) owner[b] = static_cast<uint32_t>(s);
        }
        for (size_t t = 0; t < n; ++t) {
            if (t == kRoot) continue;
            const uint32_t c = check_of(p.opos_ch(static_cast<uint32_t>(t))), b = static_cast<uint32_t>(t) ^ c;
            if (b < n && owner[b] != 0xffffffffu) out.fmap[owner[b]] |= 1u << (c & 31u);
        }
    }
    out.root_chain.assign(256, U32x4{0, 0, 0, 0});
    for (uint32_t c = 0; c < 256 && n; ++c) {
        const U32x4 &r = out.root[c];
        out.root_chain[c] = U32x4{r.x, r.y, (r.z & ~0xffu) | (r.w & 0xffu), out.fmap[r.x]};  // (r.w is 0 or 1 here)
    }

    // Leftmost automata cut their failure links at output states (reference src/nfa_builder.rs:146-201).
    // The restart scanners additionally need to know where NO occurrence can span a position; that is
    // "the classic automaton is at ROOT", so the classic links are recomputed here over the same trie
    // (breadth-first; fail(child of s on c) = delta(fail(s), c)).
    out.fail_plain.clear();
    if (!p.is_standard() && n != 0) {
        out.fail_plain.assign(n, kRoot);
        auto child_of = [&](uint32_t s, uint32_t c) -> uint32_t {
            const uint32_t b = p.base(s);
            if (b == 0) return 0xffffffffu;
            const uint32_t t = b ^ c;
            return (t < n && check_of(p.opos_ch(t)) == c && t != kRoot) ? t : 0xffffffffu;
        };
        std::vector<uint32_t> queue{kRoot};
        std::vector<uint8_t> seen(n, 0);
        seen[kRoot] = 1;
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const uint32_t s = queue[qi];
            for (uint32_t c = 0; c < 256; ++c) {
                const uint32_t t = child_of(s, c);
                if (t == 0xffffffffu || seen[t]) continue;
                seen[t] = 1;
                uint32_t f = kRoot;
                if (s != kRoot) {
                    uint32_t x = out.fail_plain[s];
                    for (;;) {
                        const uint32_t y = child_of(x, c);
                        if (y != 0xffffffffu) { f = y; break; }
                        if (x == kRoot) break;
                        x = out.fail_plain[x];
                    }
                }
                out.fail_plain[t] = f;
                queue.push_back(t);
            }
        }
    }
}

uint32_t TierTables::lds_bytes() const {
    const uint32_t rows = NA * C * (row32 ? 4u : 2u);
    const uint32_t a = (rows + 15u) & ~15u;
    const uint32_t b = ((NB - NA) * 4u + 15u) & ~15u;
    return a + 2 * b + ((NA * 8u + 15u) & ~15u) + 256u;
}

bool build_tier_tables(const HostPma &p, const RepackOptions &opt, TierTables &out) {
    out = TierTables{};
    if (!p.is_standard()) return false;  // leftmost kinds run on the DARRAY engine
    const uint32_t n = static_cast<uint32_t>(p.states.size());
    if (n == 0 || n >= (1u << 30)) return false;

    // ---- 1. breadth-first renumbering over the double array ---------------------------------
    constexpr uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> new_of_old(n, kNone), old_of_new, depth;
    std::vector<uint8_t> label_of_new;
    old_of_new.reserve(p.num_states + 1);
    old_of_new.push_back(kRoot);
    depth.push_back(0);
    label_of_new.push_back(0);
    new_of_old[kRoot] = 0;
    bool used[256] = {false};
    std::vector<uint32_t> first_child, nchild;
    for (uint32_t s = 0; s < old_of_new.size(); ++s) {
        const uint32_t old = old_of_new[s];
        const uint32_t base = p.states[old].base;
        first_child.push_back(static_cast<uint32_t>(old_of_new.size()));
        uint32_t cnt = 0;
        if (base != 0) {
            for (uint32_t c = 0; c < 256; ++c) {
                const uint32_t t = base ^ c;
                if (t >= n || check_of(p.states[t].opos_ch) != c) continue;
                if (new_of_old[t] != kNone) return false;  // not a tree: let DARRAY emulate it literally
                new_of_old[t] = static_cast<uint32_t>(old_of_new.size());
                old_of_new.push_back(t);
                depth.push_back(depth[s] + 1);
                label_of_new.push_back(static_cast<uint8_t>(c));
                used[c] = true;
                ++cnt;
            }
        }
        nchild.push_back(cnt);
    }
    const uint32_t N = static_cast<uint32_t>(old_of_new.size());

    // ---- 2. byte classes ---------------------------------------------------------------------
    out.cls.assign(256, 0);
    uint32_t C = 1;
    uint8_t rep[257] = {0};
    int unused_byte = -1;
    for (uint32_t c = 0; c < 256; ++c) {
        if (used[c]) {
            rep[C] = static_cast<uint8_t>(c);
            out.cls[c] = static_cast<uint8_t>(C++);
            if (C > 33) return false;
        } else if (unused_byte < 0) {
            unused_byte = static_cast<int>(c);
        }
    }
    if (C > 32 || unused_byte < 0) return false;
    rep[0] = static_cast<uint8_t>(unused_byte);

    // ---- 3. per-state records -----------------------------------------------------------------
    out.grec.resize(N);
    out.sopos.resize(N);
    std::vector<OutSum> osum;
    build_chain_sums(p, osum);
    out.ssum.resize(N);
    for (uint32_t s = 0; s < N; ++s) {
        const uint32_t old = old_of_new[s];
        uint32_t cmap = 0, omap = 0;
        for (uint32_t j = 0; j < nchild[s]; ++j) {
            const uint32_t ch = first_child[s] + j;
            const uint32_t k = out.cls[label_of_new[ch]];
            cmap |= 1u << k;
            if (output_pos_of(p.states[old_of_new[ch]].opos_ch) != 0) omap |= 1u << k;
        }
        const uint32_t f = new_of_old[p.states[old].fail];
        if (f == kNone) return false;
        out.grec[s] = U32x4{cmap, omap, first_child[s], f};
        const uint32_t op = output_pos_of(p.states[old].opos_ch);
        out.sopos[s] = op;
        out.ssum[s] = op ? osum[op - 1] : OutSum{0, 0};
    }
    out.root_flag = out.sopos[0] != 0;

    // ---- 4. tier boundaries --------------------------------------------------------------------
    std::vector<uint32_t> upto;  // upto[d] = number of states with depth <= d
    for (uint32_t s = 0; s < N; ++s) {
        if (depth[s] >= upto.size()) upto.resize(depth[s] + 1, 0);
        upto[depth[s]]++;
    }
    for (size_t d = 1; d < upto.size(); ++d) upto[d] += upto[d - 1];
    auto cnt_upto = [&](uint32_t d) { return d < upto.size() ? upto[d] : N; };
    auto rows_bytes = [&](uint32_t d, bool &r32) {
        r32 = cnt_upto(d + 1) >= 32768u;
        return static_cast<uint64_t>(cnt_upto(d)) * C * (r32 ? 4u : 2u) + static_cast<uint64_t>(cnt_upto(d)) * 8u;
    };
    const uint64_t fixed = 256 + 64;
    const uint64_t budget = opt.lds_budget > fixed ? opt.lds_budget - fixed : 0;
    uint32_t DA = 0;
    bool r32 = false;
    if (opt.dense_depth >= 0) {
        DA = static_cast<uint32_t>(opt.dense_depth);
        if (rows_bytes(DA, r32) > budget) return false;
    } else {
        const uint64_t rows_cap = budget * opt.rows_share_pct / 100;
        // the whole automaton dense if it fits; else the deepest level within the rows share
        const uint32_t maxd = static_cast<uint32_t>(upto.size() - 1);
        bool t32;
        if (rows_bytes(maxd, t32) <= budget) {
            DA = maxd;
        } else {
            while (DA + 1 <= maxd && rows_bytes(DA + 1, t32) <= rows_cap) ++DA;
        }
        if (rows_bytes(DA, r32) > budget) return false;  // even the root row does not fit
    }
    const uint32_t NA = cnt_upto(DA);
    const uint64_t left = budget - rows_bytes(DA, r32);
    const uint32_t NB = static_cast<uint32_t>(std::min<uint64_t>(N, NA + left / 8));

    out.C = C;
    out.N = N;
    out.NA = NA;
    out.NB = NB;
    out.dense_depth = DA;
    out.row32 = r32;

    // ---- 5. dense rows: the reference's own delta, evaluated ahead of time ----------------------
    if (r32) out.rows32.resize(static_cast<size_t>(NA) * C); else out.rows16.resize(static_cast<size_t>(NA) * C);
    for (uint32_t s = 0; s < NA; ++s) {
        const uint32_t old = old_of_new[s];
        for (uint32_t k = 0; k < C; ++k) {
            const uint32_t t_old = p.next_state(old, rep[k]);
            const uint32_t t = new_of_old[t_old];
            if (t == kNone) return false;
            const uint32_t flag = output_pos_of(p.states[t_old].opos_ch) != 0;
            if (r32) out.rows32[static_cast<size_t>(s) * C + k] = t | (flag << 31);
            else out.rows16[static_cast<size_t>(s) * C + k] = static_cast<uint16_t>(t | (flag << 15));
        }
    }
    out.bcmap.resize(NB - NA);
    out.bfail.resize(NB - NA);
    for (uint32_t s = NA; s < NB; ++s) {
        out.bcmap[s - NA] = out.grec[s].x;
        out.bfail[s - NA] = out.grec[s].w;
    }
    out.old_of_new = old_of_new;
    out.available = true;
    return true;
}

}  // namespace daac
