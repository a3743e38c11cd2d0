// Host-side automaton container: parsing, validation, serialisation (see pma.hpp).
#include "pma.hpp"

#include <algorithm>
#include <cstring>

namespace daac {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
const char *last_error_cstr() { return g_last_error.c_str(); }

bool fail_links_terminate(size_t n, const uint32_t *fail_at, size_t stride_words, bool dead_stops) {
    std::vector<uint8_t> mark(n, 0);  // 0 unknown, 1 reaches a terminal, 2 on the current walk
    std::vector<uint32_t> walk;
    if (n > kRoot) mark[kRoot] = 1;
    if (dead_stops && n > kDead) mark[kDead] = 1;
    for (size_t s = 0; s < n; ++s) {
        uint32_t cur = static_cast<uint32_t>(s);
        walk.clear();
        while (mark[cur] == 0) {
            mark[cur] = 2;
            walk.push_back(cur);
            cur = fail_at[static_cast<size_t>(cur) * stride_words];
            if (cur >= n) return false;
        }
        if (mark[cur] == 2) return false;  // came back to the walk: a cycle
        for (uint32_t w : walk) mark[w] = 1;
    }
    return true;
}

size_t HostPma::heap_bytes() const {
    return states.size() * sizeof(StateRec) + root_table.size() * sizeof(uint32_t) +
           lstates.size() * sizeof(LStateRec) + fails.size() * sizeof(uint32_t) +
           outputs.size() * sizeof(OutputRec);
}

uint32_t HostPma::max_pattern_len() const {
    uint32_t m = 0;
    for (const OutputRec &o : outputs) m = std::max(m, o.length);
    return m;
}

uint32_t HostPma::next_state(uint32_t s, uint8_t c) const {
    for (;;) {
        if (s == kRoot) return root_table[c];
        const StateRec &st = states[s];
        if (st.base != 0) {
            const uint32_t child = st.base ^ c;
            if (check_of(states[child].opos_ch) == c) return child;
        }
        s = st.fail;
    }
}

uint32_t HostPma::next_state_leftmost(uint32_t s, uint8_t c) const {
    for (;;) {
        const LStateRec &st = lstates[s];
        if (st.base != 0) {
            const uint32_t child = st.base ^ c;
            if (check_of(lstates[child].opos_ch) == c) return child;
        }
        if (s == kRoot) return kRoot;
        const uint32_t f = fails[s];
        if (f == kDead) return kRoot;
        s = f;
    }
}

void HostPma::build_root_table() {
    root_table.assign(256, kRoot);
    if (states.empty() || states[kRoot].base == 0) return;
    const uint32_t base = states[kRoot].base;
    for (uint32_t c = 0; c < 256; ++c) {
        const uint32_t child = base ^ c;
        if (child < states.size() && check_of(states[child].opos_ch) == c) root_table[c] = child;
    }
}

daac_status HostPma::validate() const {
    auto bad = [](const char *why) {
        set_error(std::string("invalid automaton: ") + why);
        return DAAC_ERR_INVALID_AUTOMATON;
    };
    const size_t nout = outputs.size();
    if (match_kind != DAAC_STANDARD) {
        if (!states.empty()) return bad("leftmost kind with non-empty standard states");
        if (lstates.empty()) return bad("empty leftmost_states");
        if (lstates.size() % kBlockLen != 0) return bad("leftmost_states.len() not a multiple of 256");
        if (fails.size() != lstates.size()) return bad("fails.len() != leftmost_states.len()");
        const size_t n = lstates.size();
        for (const LStateRec &s : lstates) {
            if (s.base != 0 && s.base >= n) return bad("base out of range");
            const uint32_t op = output_pos_of(s.opos_ch);
            if (op != 0 && static_cast<size_t>(op - 1) >= nout) return bad("output_pos out of range");
        }
        for (uint32_t f : fails)
            if (f >= n) return bad("fail out of range");
        if (!fail_links_terminate(n, fails.data(), 1, true)) return bad("failure links do not end at the root");
    } else {
        if (!lstates.empty() || !fails.empty()) return bad("standard kind with leftmost arrays");
        if (states.empty()) return bad("empty states");
        if (states.size() % kBlockLen != 0) return bad("states.len() not a multiple of 256");
        if (root_table.size() != kBlockLen) return bad("root_table");
        const size_t n = states.size();
        for (const StateRec &s : states) {
            if (s.base != 0 && s.base >= n) return bad("base out of range");
            if (s.fail >= n) return bad("fail out of range");
            const uint32_t op = output_pos_of(s.opos_ch);
            if (op != 0 && static_cast<size_t>(op - 1) >= nout) return bad("output_pos out of range");
        }
        if (!fail_links_terminate(n, &states[0].fail, 3, false)) return bad("failure links do not end at the root");
    }
    for (size_t i = 0; i < nout; ++i) {
        const uint32_t par = outputs[i].parent;
        if (par != 0 && static_cast<size_t>(par - 1) >= i) return bad("output parent not below its child");
    }
    return DAAC_OK;
}

namespace {
struct Writer {
    std::vector<uint8_t> &v;
    void u32(uint32_t x) {
        const uint8_t b[4] = {uint8_t(x), uint8_t(x >> 8), uint8_t(x >> 16), uint8_t(x >> 24)};
        v.insert(v.end(), b, b + 4);
    }
};
struct Reader {
    const uint8_t *p;
    size_t left;
    bool u32(uint32_t &x) {
        if (left < 4) return false;
        x = uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24;
        p += 4;
        left -= 4;
        return true;
    }
    // Vec<S> header with the allocation guard of serializer.rs:110-118 (in-memory element size)
    bool vec_len(size_t mem_size, uint32_t &n) {
        if (!u32(n)) return false;
        return static_cast<uint64_t>(n) * mem_size <= left;
    }
};
}  // namespace

void HostPma::serialize(std::vector<uint8_t> &out) const {
    out.clear();
    out.reserve(4 + states.size() * 12 + 4 + lstates.size() * 8 + 4 + fails.size() * 4 + 4 + outputs.size() * 12 + 5);
    Writer w{out};
    w.u32(static_cast<uint32_t>(states.size()));
    for (const StateRec &s : states) { w.u32(s.base); w.u32(s.fail); w.u32(s.opos_ch); }
    w.u32(static_cast<uint32_t>(lstates.size()));
    for (const LStateRec &s : lstates) { w.u32(s.base); w.u32(s.opos_ch); }
    w.u32(static_cast<uint32_t>(fails.size()));
    for (uint32_t f : fails) w.u32(f);
    w.u32(static_cast<uint32_t>(outputs.size()));
    for (const OutputRec &o : outputs) { w.u32(o.value); w.u32(o.length); w.u32(o.parent); }
    out.push_back(match_kind);
    w.u32(num_states);
}

daac_status HostPma::deserialize(const uint8_t *src, size_t len, HostPma &out, size_t *consumed) {
    auto trunc = []() {
        set_error("invalid automaton: truncated or oversized blob");
        return DAAC_ERR_INVALID_AUTOMATON;
    };
    Reader r{src, len};
    uint32_t n = 0;
    HostPma p;
    if (!r.vec_len(sizeof(StateRec), n)) return trunc();
    p.states.resize(n);
    for (StateRec &s : p.states)
        if (!r.u32(s.base) || !r.u32(s.fail) || !r.u32(s.opos_ch)) return trunc();
    if (!r.vec_len(sizeof(LStateRec), n)) return trunc();
    p.lstates.resize(n);
    for (LStateRec &s : p.lstates)
        if (!r.u32(s.base) || !r.u32(s.opos_ch)) return trunc();
    if (!r.vec_len(sizeof(uint32_t), n)) return trunc();
    p.fails.resize(n);
    for (uint32_t &f : p.fails)
        if (!r.u32(f)) return trunc();
    if (!r.vec_len(sizeof(OutputRec), n)) return trunc();
    p.outputs.resize(n);
    for (OutputRec &o : p.outputs)
        if (!r.u32(o.value) || !r.u32(o.length) || !r.u32(o.parent)) return trunc();
    if (r.left < 1) return trunc();
    const uint8_t kind = *r.p++;
    r.left--;
    p.match_kind = kind == 1 ? DAAC_LEFTMOST_LONGEST : kind == 2 ? DAAC_LEFTMOST_FIRST : DAAC_STANDARD;  // lib.rs:366-374
    if (!r.u32(p.num_states)) return trunc();
    if (p.match_kind == DAAC_STANDARD) p.build_root_table();
    const daac_status st = p.validate();
    if (st != DAAC_OK) return st;
    if (consumed) *consumed = len - r.left;
    out = std::move(p);
    return DAAC_OK;
}

}  // namespace daac
