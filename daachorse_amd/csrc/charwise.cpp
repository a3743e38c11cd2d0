// Host-side container of the charwise automaton: parsing, validation, serialisation, and the
// device tables derived from it (see charwise.hpp).
#include "charwise.hpp"

#include <algorithm>
#include <string>

#include "repack.hpp"

namespace daac {

uint32_t HostCharPma::max_pattern_len() const {
    uint32_t m = 0;
    for (const OutputRec &o : outputs) m = std::max(m, o.length);
    return m;
}

// alphabet_size.next_power_of_two().max(2): the XOR of a BASE with any code stays inside its block
static uint64_t block_len_of(uint32_t alphabet_size) {
    uint64_t b = 2;
    while (b < alphabet_size) b <<= 1;
    return b;
}

daac_status HostCharPma::validate() const {
    auto bad = [](const char *why) {
        set_error(std::string("invalid automaton: ") + why);
        return DAAC_ERR_INVALID_AUTOMATON;
    };
    for (uint32_t code : table)
        if (code != kInvalidCode && code >= alphabet_size) return bad("mapped code >= alphabet_size");
    const size_t n = states.size();
    if (n == 0) return bad("empty states");
    if (n % block_len_of(alphabet_size) != 0) return bad("states.len() not a multiple of the block length");
    for (const CStateRec &s : states) {
        if (s.base != 0 && s.base >= n) return bad("base out of range");
        if (s.fail >= n) return bad("fail out of range");
        if (s.output_pos != 0 && static_cast<size_t>(s.output_pos - 1) >= outputs.size()) return bad("output_pos out of range");
    }
    // charwise.rs:1022-1092 follows FAIL until ROOT (leftmost kinds also stop at DEAD; vacant slots
    // carry FAIL = DEAD in every kind, charwise.rs:1103-1112)
    if (!fail_links_terminate(n, &states[0].fail, 4, true)) return bad("failure links do not end at the root");
    for (size_t i = 0; i < outputs.size(); ++i) {
        const uint32_t par = outputs[i].parent;
        if (par != 0 && static_cast<size_t>(par - 1) >= i) return bad("output parent not below its child");
    }
    return DAAC_OK;
}

namespace {
inline void put_u32(std::vector<uint8_t> &v, uint32_t x) {
    const uint8_t b[4] = {uint8_t(x), uint8_t(x >> 8), uint8_t(x >> 16), uint8_t(x >> 24)};
    v.insert(v.end(), b, b + 4);
}
struct Cursor {
    const uint8_t *p;
    size_t left;
    bool u32(uint32_t &x) {
        if (left < 4) return false;
        x = uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24;
        p += 4;
        left -= 4;
        return true;
    }
    // Vec<S> length with the allocation guard of serializer.rs:110-118 (in-memory element size)
    bool count(size_t mem_size, uint32_t &n) { return u32(n) && static_cast<uint64_t>(n) * mem_size <= left; }
};
}  // namespace

void HostCharPma::serialize(std::vector<uint8_t> &out) const {
    out.clear();
    out.reserve(4 + states.size() * 16 + 4 + table.size() * 4 + 4 + 4 + outputs.size() * 12 + 5);
    put_u32(out, static_cast<uint32_t>(states.size()));
    for (const CStateRec &s : states) { put_u32(out, s.base); put_u32(out, s.check); put_u32(out, s.fail); put_u32(out, s.output_pos); }
    put_u32(out, static_cast<uint32_t>(table.size()));  // CodeMapper: table, alphabet_size (mapper.rs:61-77)
    for (uint32_t c : table) put_u32(out, c);
    put_u32(out, alphabet_size);
    put_u32(out, static_cast<uint32_t>(outputs.size()));
    for (const OutputRec &o : outputs) { put_u32(out, o.value); put_u32(out, o.length); put_u32(out, o.parent); }
    out.push_back(match_kind);
    put_u32(out, num_states);
}

daac_status HostCharPma::deserialize(const uint8_t *src, size_t len, HostCharPma &out, size_t *consumed) {
    auto truncated = []() {
        set_error("invalid automaton: truncated or oversized serialized data");
        return DAAC_ERR_INVALID_AUTOMATON;
    };
    Cursor r{src, len};
    HostCharPma p;
    uint32_t n;
    if (!r.count(16, n)) return truncated();
    p.states.resize(n);
    for (CStateRec &s : p.states)
        if (!r.u32(s.base) || !r.u32(s.check) || !r.u32(s.fail) || !r.u32(s.output_pos)) return truncated();
    if (!r.count(4, n)) return truncated();
    p.table.resize(n);
    for (uint32_t &c : p.table)
        if (!r.u32(c)) return truncated();
    if (!r.u32(p.alphabet_size)) return truncated();
    if (!r.count(12, n)) return truncated();
    p.outputs.resize(n);
    for (OutputRec &o : p.outputs)
        if (!r.u32(o.value) || !r.u32(o.length) || !r.u32(o.parent)) return truncated();
    if (r.left < 1) return truncated();
    const uint8_t kind = *r.p++;
    --r.left;
    p.match_kind = kind == 1 ? DAAC_LEFTMOST_LONGEST : kind == 2 ? DAAC_LEFTMOST_FIRST : DAAC_STANDARD;  // lib.rs:349-358
    if (!r.u32(p.num_states)) return truncated();
    const daac_status st = p.validate();
    if (st != DAAC_OK) return st;
    if (consumed) *consumed = len - r.left;
    out = std::move(p);
    return DAAC_OK;
}

// ---------------------------------------------------------------------------------- device tables
void build_char_tables(const HostCharPma &p, CharTables &out) {
    out.states = p.states;
    out.table = p.table;
    out.root_flag = p.states[kRoot].output_pos != 0;

    // {count, h32 sum} of every output chain; parents sit below their children (validate())
    out.osum.resize(p.outputs.size());
    for (size_t i = 0; i < p.outputs.size(); ++i) {
        const OutputRec &o = p.outputs[i];
        OutSum s{1u, match_hash32(o.value, o.length)};
        if (o.parent != 0) { s.cnt += out.osum[o.parent - 1].cnt; s.hsum += out.osum[o.parent - 1].hsum; }
        out.osum[i] = s;
    }

    out.fail_plain.clear();
    if (p.is_standard()) {  // the automaton's own links are the classic ones
        // no state of a well-formed Standard automaton fails to DEAD; make sure a malformed one cannot spin there
        for (CStateRec &s : out.states)
            if (s.fail == kDead) s.fail = kRoot;
        return;
    }

    // Leftmost kinds cut their failure links at DEAD below every output state (nfa_builder.rs:
    // 144-180).  The restart scanners need the classic links of the same trie to find the
    // positions where no pattern occurrence is in flight.  CHECK holds the parent, so the trie is
    // recovered by grouping the slots by parent; links are then computed breadth-first.
    const size_t n = p.states.size();
    std::vector<uint32_t> child_cnt(n + 1, 0);
    auto is_child = [&](size_t i) {
        if (i == kRoot) return false;
        const uint32_t par = p.states[i].check;
        if (par >= n || par == i) return false;
        const uint32_t base = p.states[par].base;
        return base != 0 && (base ^ static_cast<uint32_t>(i)) < p.alphabet_size;
    };
    for (size_t i = 0; i < n; ++i)
        if (is_child(i)) ++child_cnt[p.states[i].check + 1];
    for (size_t i = 0; i < n; ++i) child_cnt[i + 1] += child_cnt[i];
    std::vector<uint32_t> kids(child_cnt[n]);
    {
        std::vector<uint32_t> fill(child_cnt.begin(), child_cnt.end() - 1);
        for (size_t i = 0; i < n; ++i)
            if (is_child(i)) kids[fill[p.states[i].check]++] = static_cast<uint32_t>(i);
    }
    auto goto_child = [&](uint32_t s, uint32_t code, uint32_t &child) {
        const uint32_t base = p.states[s].base;
        if (base == 0) return false;
        child = base ^ code;  // stays inside base's block, hence < n
        return p.states[child].check == s;
    };
    out.fail_plain.assign(n, kRoot);
    std::vector<uint8_t> seen(n, 0);
    std::vector<uint32_t> depth(n, 0);
    std::vector<uint32_t> queue{kRoot};
    seen[kRoot] = 1;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const uint32_t u = queue[qi];
        for (uint32_t k = child_cnt[u]; k < child_cnt[u + 1]; ++k) {
            const uint32_t v = kids[k];
            if (seen[v]) continue;  // only a corrupt blob has slots reachable twice
            seen[v] = 1;
            depth[v] = depth[u] + 1;
            const uint32_t code = p.states[u].base ^ v;
            uint32_t f = kRoot;
            if (u != kRoot) {
                uint32_t w = out.fail_plain[u], t;
                for (;;) {
                    // a link only ever points to a shallower state, so following links always ends
                    if (goto_child(w, code, t) && seen[t] && depth[t] <= depth[u]) { f = t; break; }
                    if (w == kRoot) break;
                    w = out.fail_plain[w];
                }
            }
            out.fail_plain[v] = f;
            queue.push_back(v);
        }
    }
}

}  // namespace daac
