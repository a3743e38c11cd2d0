// Host-side construction of the bytewise double-array automaton (C++17).
//
// Construction is a one-off, sequential, pointer-heavy job that stays on the CPU;
// it is here so that `DoubleArrayAhoCorasick::new(patterns)` works without the Rust crate.  The
// arrays it produces are byte-identical to the crate's (same trie numbering, same BASE search
// order, same CHECK values on vacant slots), so serialize() blobs are interchangeable:
//   sparse trie + failure links + merged output lists   reference src/nfa_builder.rs:78-222
//   label-sorted edge lists                             reference src/edge_map.rs
//   vacant-slot ring over the trailing blocks           reference src/build_helper.rs:16-227
//   double-array placement                              reference src/bytewise/builder.rs:204-400
#include <algorithm>
#include <limits>
#include <utility>

#include "build_common.hpp"
#include "pma.hpp"

namespace daac {
namespace {

using build::VacantRing;
using TrieNode = build::TrieNode<uint8_t>;
using SparseTrie = build::SparseTrie<uint8_t>;
constexpr uint32_t kU24Max = 0x00ffffffu;  // intpack.rs:15

inline void set_check(StateRec &s, uint8_t c) { s.opos_ch = (s.opos_ch & ~0xffu) | c; }  // intpack.rs:50-53

// ---------------------------------------------------------------------------- double-array placement
class Placer {
public:
    Placer(std::vector<StateRec> &slots, uint32_t window) : slots_(slots), ring_(kBlockLen, window) {}

    // bytewise/builder.rs:267-334
    daac_status run(const std::vector<TrieNode> &nodes) {
        slots_.assign(kBlockLen, StateRec{0, 0, 0});  // init_array, :336-344
        daac_status st = ring_.append_block();
        if (st != DAAC_OK) return st;
        ring_.take_index(kRoot);
        ring_.take_index(kDead);

        std::vector<uint32_t> slot_of(nodes.size(), kDead);
        slot_of[kRoot] = kRoot;
        std::vector<uint32_t> stack{kRoot};
        std::vector<uint8_t> labels;
        while (!stack.empty()) {
            const uint32_t id = stack.back();
            stack.pop_back();
            const TrieNode &s = nodes[id];
            if (s.edges.empty()) continue;
            labels.clear();
            for (const auto &e : s.edges) labels.push_back(e.first);
            const uint32_t base = pick_base(labels);
            if (base >= slots_.size() && (st = grow()) != DAAC_OK) return st;
            for (const auto &e : s.edges) {
                const uint32_t slot = base ^ e.first;  // XOR addressing keeps children in base's block
                ring_.take_index(slot);
                set_check(slots_[slot], e.first);
                slot_of[e.second] = slot;
                stack.push_back(e.second);
            }
            slots_[slot_of[id]].base = base;
            ring_.take_base(base);
        }
        for (size_t id = 0; id < nodes.size(); ++id) {  // :307-326
            if (id == kDead) continue;
            StateRec &rec = slots_[slot_of[id]];
            if (nodes[id].output_pos > kU24Max) {
                set_error("output_pos must be <= 2^24-1");
                return DAAC_ERR_AUTOMATON_SCALE;
            }
            rec.opos_ch = (nodes[id].output_pos << 8) | (rec.opos_ch & 0xffu);
            rec.fail = nodes[id].fail == kDead ? kDead : slot_of[nodes[id].fail];
        }
        for (uint32_t b = ring_.first_active_block(); b < ring_.blocks(); ++b) seal_block(b);  // :328-330
        slots_.shrink_to_fit();
        return DAAC_OK;
    }

private:
    // builder.rs:347-370: first vacant slot (ascending) whose implied BASE is free and non-zero and
    // whose sibling slots are all vacant; otherwise the first slot of a block yet to be added.
    uint32_t pick_base(const std::vector<uint8_t> &labels) const {
        if (ring_.has_vacant()) {
            uint32_t slot = ring_.head();
            do {
                const uint32_t base = slot ^ labels[0];
                bool ok = base != 0 && !ring_.base_taken(base);
                for (size_t k = 0; ok && k < labels.size(); ++k) ok = !ring_.index_taken(base ^ labels[k]);
                if (ok) return base;
                slot = ring_.next_of(slot);
            } while (slot != ring_.head());
        }
        return static_cast<uint32_t>(slots_.size());
    }

    // builder.rs:372-388
    daac_status grow() {
        if (slots_.size() > std::numeric_limits<uint32_t>::max() - kBlockLen) {
            set_error("states.len() must be <= u32::MAX");
            return DAAC_ERR_AUTOMATON_SCALE;
        }
        uint32_t closing;
        if (ring_.closing_block(closing)) seal_block(closing);
        const daac_status st = ring_.append_block();
        if (st != DAAC_OK) return st;
        slots_.resize(slots_.size() + kBlockLen, StateRec{0, 0, 0});
        return DAAC_OK;
    }

    // builder.rs:391-400: give every vacant slot (and ROOT / DEAD) a CHECK no real BASE can reach
    void seal_block(uint32_t blk) {
        uint32_t spare;
        if (!ring_.free_base_in_block(blk, spare)) return;
        for (uint32_t c = 0; c < 256; ++c) {
            const uint32_t slot = spare ^ c;
            if (slot == kRoot || slot == kDead || !ring_.index_taken(slot)) set_check(slots_[slot], static_cast<uint8_t>(c));
        }
    }

    std::vector<StateRec> &slots_;
    VacantRing ring_;
};

}  // namespace

// bytewise/builder.rs:152-244
daac_status build_bytewise(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n, uint8_t match_kind,
                           uint32_t num_free_blocks, HostPma &out) {
    if (match_kind > 2) { set_error("match_kind must be 0, 1 or 2"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (num_free_blocks < 1) { set_error("num_free_blocks must be >= 1"); return DAAC_ERR_INVALID_ARGUMENT; }  // builder.rs:113
    if (static_cast<uint64_t>(num_free_blocks) * kBlockLen > std::numeric_limits<uint32_t>::max()) {  // build_helper.rs:31-33
        set_error("block_len * num_free_blocks must be <= u32::MAX");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
    if (!values && n > std::numeric_limits<uint32_t>::max()) {  // builder.rs:160-165
        set_error("index cannot be converted to V");
        return DAAC_ERR_INVALID_CONVERSION;
    }
    SparseTrie trie(match_kind);
    for (size_t i = 0; i < n; ++i) {
        const size_t plen = static_cast<size_t>(offsets[i + 1] - offsets[i]);
        const daac_status st = trie.add(blob + offsets[i], plen, plen, values ? values[i] : static_cast<uint32_t>(i));
        if (st != DAAC_OK) return st;
    }
    if (trie.num_patterns() > kU24Max) {  // builder.rs:256-258
        set_error("patvals.len() must be <= 2^24-1");
        return DAAC_ERR_AUTOMATON_SCALE;
    }
    trie.link_failures();
    HostPma p;
    trie.merge_outputs(p.outputs);
    std::vector<StateRec> slots;
    Placer placer(slots, num_free_blocks);
    const daac_status st = placer.run(trie.nodes());
    if (st != DAAC_OK) return st;
    p.match_kind = match_kind;
    p.num_states = static_cast<uint32_t>(trie.nodes().size() - 1);  // the dead state does not count
    if (match_kind == DAAC_STANDARD) {
        p.states = std::move(slots);
        p.build_root_table();
    } else {  // builder.rs:220-231: split into 8-byte hot records + failure links
        p.lstates.reserve(slots.size());
        p.fails.reserve(slots.size());
        for (const StateRec &s : slots) {
            p.lstates.push_back(LStateRec{s.base, s.opos_ch});
            p.fails.push_back(s.fail);
        }
    }
    out = std::move(p);
    return DAAC_OK;
}

}  // namespace daac
