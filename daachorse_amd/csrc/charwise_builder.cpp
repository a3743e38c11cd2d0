// Host-side construction of the charwise double-array automaton (C++17), so that
// `CharwiseDoubleArrayAhoCorasick::new(patterns)` works without the Rust crate.  The arrays are
// byte-identical to the crate's — same trie numbering, code assignment and BASE search order — so
// serialize() blobs are interchangeable:
//   sparse trie over code points, failure links, output lists   reference src/nfa_builder.rs:78-222
//   code point -> code by descending frequency                  reference src/charwise/mapper.rs:16-34
//   double-array placement, CHECK = parent slot                  reference src/charwise/builder.rs:178-359
#include <algorithm>
#include <limits>
#include <utility>

#include "build_common.hpp"
#include "charwise.hpp"

namespace daac {
namespace {

using CharTrie = build::SparseTrie<uint32_t>;
using CharNode = build::TrieNode<uint32_t>;

// One scalar value of a UTF-8 string (patterns arrive as &str in the reference: well-formed).
// Malformed or truncated input is rejected instead of read past.
bool next_scalar(const uint8_t *p, size_t len, size_t &i, uint32_t &cp) {
    const uint32_t b0 = p[i];
    uint32_t need;
    if (b0 < 0x80) { cp = b0; need = 0; }
    else if (b0 < 0xc0) return false;
    else if (b0 < 0xe0) { cp = b0 & 0x1f; need = 1; }
    else if (b0 < 0xf0) { cp = b0 & 0x0f; need = 2; }
    else if (b0 < 0xf8) { cp = b0 & 0x07; need = 3; }
    else return false;
    if (len - i <= need) return false;
    for (uint32_t k = 1; k <= need; ++k) {
        const uint32_t b = p[i + k];
        if ((b & 0xc0) != 0x80) return false;
        cp = (cp << 6) | (b & 0x3f);
    }
    i += need + 1;
    return cp <= 0x10ffff;
}

class CharPlacer {
public:
    CharPlacer(HostCharPma &p, uint32_t block_len, uint32_t window) : p_(p), bl_(block_len), ring_(block_len, window) {}

    // charwise/builder.rs:241-304
    daac_status run(const std::vector<CharNode> &nodes) {
        p_.states.assign(bl_, blank());
        daac_status st = ring_.append_block();
        if (st != DAAC_OK) return st;
        ring_.take_index(kRoot);
        ring_.take_index(kDead);

        std::vector<uint32_t> slot_of(nodes.size(), kDead);
        slot_of[kRoot] = kRoot;
        std::vector<uint32_t> stack{kRoot};
        std::vector<std::pair<uint32_t, uint32_t>> coded;  // (code, child node), ascending code
        while (!stack.empty()) {
            const uint32_t id = stack.back();
            stack.pop_back();
            const CharNode &s = nodes[id];
            if (s.edges.empty()) continue;
            coded.clear();
            for (const auto &e : s.edges) coded.emplace_back(p_.code_of(e.first), e.second);
            std::sort(coded.begin(), coded.end());
            const uint32_t base = pick_base(coded);
            if (base >= p_.states.size() && (st = grow()) != DAAC_OK) return st;
            const uint32_t parent = slot_of[id];
            for (const auto &ce : coded) {
                const uint32_t slot = base ^ ce.first;
                ring_.take_index(slot);
                p_.states[slot].check = parent;  // CHECK names the parent: siblings of different parents may share a BASE
                slot_of[ce.second] = slot;
                stack.push_back(ce.second);
            }
            p_.states[parent].base = base;
        }
        for (size_t id = 0; id < nodes.size(); ++id) {  // :284-302
            if (id == kDead) continue;
            CStateRec &rec = p_.states[slot_of[id]];
            rec.output_pos = nodes[id].output_pos;
            rec.fail = nodes[id].fail == kDead ? kDead : slot_of[nodes[id].fail];
        }
        p_.states.shrink_to_fit();
        return DAAC_OK;
    }

private:
    static CStateRec blank() { return CStateRec{0, kDead, kDead, 0}; }  // State::default(), charwise.rs:1103-1112

    // builder.rs:320-333: the first vacant slot (ascending) that can hold the lowest code with all
    // sibling slots vacant; otherwise the same position relative to the block about to be added.
    uint32_t pick_base(const std::vector<std::pair<uint32_t, uint32_t>> &coded) const {
        const uint32_t c0 = coded[0].first;
        if (ring_.has_vacant()) {
            uint32_t slot = ring_.head();
            do {
                const uint32_t base = slot ^ c0;
                bool ok = base != 0;
                for (size_t k = 0; ok && k < coded.size(); ++k) ok = !ring_.index_taken(base ^ coded[k].first);
                if (ok) return base;
                slot = ring_.next_of(slot);
            } while (slot != ring_.head());
        }
        return static_cast<uint32_t>(p_.states.size()) ^ c0;
    }

    // builder.rs:346-358
    daac_status grow() {
        if (p_.states.size() > std::numeric_limits<uint32_t>::max() - bl_) {
            set_error("states.len() must be <= u32::MAX");
            return DAAC_ERR_AUTOMATON_SCALE;
        }
        const daac_status st = ring_.append_block();
        if (st != DAAC_OK) return st;
        p_.states.resize(p_.states.size() + bl_, blank());
        return DAAC_OK;
    }

    HostCharPma &p_;
    uint32_t bl_;
    build::VacantRing ring_;
};

}  // namespace

// charwise/builder.rs:178-239
daac_status build_charwise(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n, uint8_t match_kind,
                           uint32_t num_free_blocks, HostCharPma &out) {
    if (match_kind > 2) { set_error("match_kind must be 0, 1 or 2"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (num_free_blocks < 1) { set_error("num_free_blocks must be >= 1"); return DAAC_ERR_INVALID_ARGUMENT; }  // builder.rs:131
    if (!values && n > std::numeric_limits<uint32_t>::max()) {
        set_error("index cannot be converted to V");
        return DAAC_ERR_INVALID_CONVERSION;
    }
    CharTrie trie(match_kind);
    std::vector<uint32_t> freq;  // occurrences per code point over ALL input patterns, pruned or not (:217-229)
    std::vector<uint32_t> chars;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *pat = blob + offsets[i];
        const size_t plen = static_cast<size_t>(offsets[i + 1] - offsets[i]);
        chars.clear();
        for (size_t k = 0; k < plen;) {
            uint32_t cp;
            if (!next_scalar(pat, plen, k, cp)) {
                set_error("pattern " + std::to_string(i) + " is not valid UTF-8");
                return DAAC_ERR_INVALID_ARGUMENT;
            }
            chars.push_back(cp);
        }
        const daac_status st = trie.add(chars.data(), chars.size(), plen, values ? values[i] : static_cast<uint32_t>(i));
        if (st != DAAC_OK) return st;
        for (uint32_t cp : chars) {
            if (freq.size() <= cp) freq.resize(static_cast<size_t>(cp) + 1, 0);
            ++freq[cp];
        }
    }

    HostCharPma p;
    {  // CodeMapper::new, mapper.rs:16-34: frequent code points get small codes; ties by code point
        std::vector<std::pair<uint32_t, uint32_t>> order;  // (code point, frequency)
        for (size_t cp = 0; cp < freq.size(); ++cp)
            if (freq[cp] != 0) order.emplace_back(static_cast<uint32_t>(cp), freq[cp]);
        std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
        p.table.assign(freq.size(), kInvalidCode);
        for (size_t k = 0; k < order.size(); ++k) p.table[order[k].first] = static_cast<uint32_t>(k);
        p.alphabet_size = static_cast<uint32_t>(order.size());
    }
    trie.link_failures();
    trie.merge_outputs(p.outputs);

    uint32_t block_len = 2;  // alphabet_size.next_power_of_two().max(2), builder.rs:308
    while (block_len < p.alphabet_size) block_len <<= 1;
    if (static_cast<uint64_t>(block_len) * num_free_blocks > std::numeric_limits<uint32_t>::max()) {  // build_helper.rs:31-33
        set_error("block_len * num_free_blocks must be <= u32::MAX");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
    CharPlacer placer(p, block_len, num_free_blocks);
    const daac_status st = placer.run(trie.nodes());
    if (st != DAAC_OK) return st;
    p.match_kind = match_kind;
    p.num_states = static_cast<uint32_t>(trie.nodes().size() - 1);  // the dead state does not count
    out = std::move(p);
    return DAAC_OK;
}

}  // namespace daac
