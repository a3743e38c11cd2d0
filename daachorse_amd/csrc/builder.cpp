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

#include "pma.hpp"

namespace daac {
namespace {

constexpr uint32_t kU24Max = 0x00ffffffu;  // intpack.rs:15

// ------------------------------------------------------------------------------------ sparse trie
struct TrieNode {
    std::vector<std::pair<uint8_t, uint32_t>> edges;  // sorted by label (edge_map.rs)
    std::vector<std::pair<uint32_t, uint32_t>> own;   // (value, length) of patterns ending here
    uint32_t fail = kRoot;
    uint32_t output_pos = 0;                           // 1-based head of the merged list, 0 = none

    const uint32_t *child(uint8_t c) const {
        auto it = std::lower_bound(edges.begin(), edges.end(), c, [](const auto &e, uint8_t key) { return e.first < key; });
        return (it != edges.end() && it->first == c) ? &it->second : nullptr;
    }
};

class SparseTrie {
public:
    explicit SparseTrie(uint8_t kind) : kind_(kind), nodes_(2) {}  // node 0 = root, node 1 = dead

    // nfa_builder.rs:78-113
    daac_status add(const uint8_t *pat, size_t len, uint32_t value) {
        if (len > std::numeric_limits<uint32_t>::max()) {
            set_error("pattern.len() must be <= u32::MAX");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
        uint32_t cur = kRoot;
        for (size_t i = 0; i < len; ++i) {
            // LeftmostFirst: nothing below an earlier-registered pattern can ever be reported
            if (kind_ == DAAC_LEFTMOST_FIRST && !nodes_[cur].own.empty()) return DAAC_OK;
            const uint8_t c = pat[i];
            if (const uint32_t *nx = nodes_[cur].child(c)) {
                cur = *nx;
                continue;
            }
            if (nodes_.size() >= std::numeric_limits<uint32_t>::max()) {
                set_error("state_id must be < u32::MAX");
                return DAAC_ERR_AUTOMATON_SCALE;
            }
            const uint32_t fresh = static_cast<uint32_t>(nodes_.size());
            auto &ed = nodes_[cur].edges;
            ed.insert(std::lower_bound(ed.begin(), ed.end(), c, [](const auto &e, uint8_t key) { return e.first < key; }),
                      std::make_pair(c, fresh));
            nodes_.emplace_back();
            cur = fresh;
        }
        nodes_[cur].own.emplace_back(value, static_cast<uint32_t>(len));
        ++num_patterns_;
        return DAAC_OK;
    }

    // nfa_builder.rs:115-144 (Standard) / 146-201 (leftmost kinds); fills bfs_ with the visit order
    void link_failures() {
        const bool leftmost = kind_ != DAAC_STANDARD;
        bfs_.clear();
        bfs_.reserve(nodes_.size());
        for (const auto &e : nodes_[kRoot].edges) bfs_.push_back(e.second);
        if (leftmost && !nodes_[kRoot].own.empty())
            for (const auto &e : nodes_[kRoot].edges) nodes_[e.second].fail = kDead;
        for (size_t qi = 0; qi < bfs_.size(); ++qi) {
            TrieNode &s = nodes_[bfs_[qi]];
            if (leftmost && !s.own.empty()) s.fail = kDead;  // an output state never falls back
            for (const auto &e : s.edges) {
                nodes_[e.second].fail = leftmost ? fail_target_leftmost(s.fail, e.first) : fail_target(s.fail, e.first);
                bfs_.push_back(e.second);
            }
        }
    }

    // nfa_builder.rs:203-222: own outputs (first-registered first) followed by the fail state's list
    void merge_outputs(std::vector<OutputRec> &outputs) {
        auto emit = [&](TrieNode &s, uint32_t tail) {
            uint32_t last = tail;
            for (auto it = s.own.rbegin(); it != s.own.rend(); ++it) {
                outputs.push_back(OutputRec{it->first, it->second, last});
                last = static_cast<uint32_t>(outputs.size());
            }
            s.output_pos = last;
        };
        emit(nodes_[kRoot], 0);
        for (uint32_t id : bfs_) emit(nodes_[id], nodes_[nodes_[id].fail].output_pos);
    }

    const std::vector<TrieNode> &nodes() const { return nodes_; }
    size_t num_patterns() const { return num_patterns_; }

private:
    uint32_t fail_target(uint32_t f, uint8_t c) const {
        for (;;) {
            if (const uint32_t *t = nodes_[f].child(c)) return *t;
            const uint32_t up = nodes_[f].fail;
            if (f == kRoot && up == kRoot) return kRoot;
            f = up;
        }
    }
    uint32_t fail_target_leftmost(uint32_t f, uint8_t c) const {
        if (f == kDead) return kDead;
        for (;;) {
            if (const uint32_t *t = nodes_[f].child(c)) return *t;
            const uint32_t up = nodes_[f].fail;
            if (up == kDead) return kDead;
            if (f == kRoot && up == kRoot) return kRoot;
            f = up;
        }
    }

    uint8_t kind_;
    std::vector<TrieNode> nodes_;
    std::vector<uint32_t> bfs_;
    size_t num_patterns_ = 0;
};

// ------------------------------------------------------------------------- vacant-slot bookkeeping
// Tracks, for the last `window` blocks only, which slots are taken, which BASE values are taken, and
// a circular list of the vacant slots in ascending order (build_helper.rs).  Indices are global;
// storage is a ring of window * 256 entries.
class VacantRing {
public:
    explicit VacantRing(uint32_t window) : window_(window), cap_(window * kBlockLen), cell_(cap_) {}

    uint32_t blocks() const { return blocks_; }
    uint32_t first_active_block() const { return blocks_ > window_ ? blocks_ - window_ : 0; }
    bool index_taken(uint32_t i) const { return at(i).taken; }
    bool base_taken(uint32_t b) const { return at(b).base_taken; }
    void take_base(uint32_t b) { at(b).base_taken = true; }
    bool has_vacant() const { return has_head_; }
    uint32_t head() const { return head_; }
    uint32_t next_of(uint32_t i) const { return at(i).next; }

    // build_helper.rs:118-130
    void take_index(uint32_t i) {
        Cell &c = at(i);
        c.taken = true;
        at(c.prev).next = c.next;
        at(c.next).prev = c.prev;
        if (head_ == i) {
            if (c.next != i) head_ = c.next; else has_head_ = false;
        }
    }

    // The block that leaves the window on the next append, if the window is full (:177-179).
    bool closing_block(uint32_t &blk) const {
        if (cap_ <= blocks_ * kBlockLen) { blk = first_active_block(); return true; }
        return false;
    }

    // build_helper.rs:133-173
    daac_status append_block() {
        if (blocks_ * static_cast<uint64_t>(kBlockLen) > std::numeric_limits<uint32_t>::max() - kBlockLen) {
            set_error("num_elements must be <= u32::MAX");
            return DAAC_ERR_AUTOMATON_SCALE;
        }
        uint32_t closing;
        if (closing_block(closing)) {
            const uint32_t limit = (closing + 1) * kBlockLen;  // retire what is still vacant there
            while (has_head_ && head_ < limit) take_index(head_);
        }
        const uint32_t lo = blocks_ * kBlockLen, hi = lo + kBlockLen;
        ++blocks_;
        for (uint32_t i = lo; i < hi; ++i) at(i) = Cell{i + 1, i - 1, false, false};
        if (has_head_) {
            const uint32_t tail = at(head_).prev;
            at(lo).prev = tail;
            at(tail).next = lo;
            at(hi - 1).next = head_;
            at(head_).prev = hi - 1;
        } else {
            at(lo).prev = hi - 1;
            at(hi - 1).next = lo;
            head_ = lo;
            has_head_ = true;
        }
        return DAAC_OK;
    }

    // build_helper.rs:76-80
    bool free_base_in_block(uint32_t blk, uint32_t &base) const {
        for (uint32_t b = blk * kBlockLen; b < (blk + 1) * kBlockLen; ++b)
            if (!base_taken(b)) { base = b; return true; }
        return false;
    }

private:
    struct Cell { uint32_t next = 0, prev = 0; bool base_taken = false, taken = false; };
    Cell &at(uint32_t i) { return cell_[i % cap_]; }
    const Cell &at(uint32_t i) const { return cell_[i % cap_]; }

    uint32_t window_, cap_;
    std::vector<Cell> cell_;
    uint32_t blocks_ = 0;
    uint32_t head_ = 0;
    bool has_head_ = false;
};

inline void set_check(StateRec &s, uint8_t c) { s.opos_ch = (s.opos_ch & ~0xffu) | c; }  // intpack.rs:50-53

// ---------------------------------------------------------------------------- double-array placement
class Placer {
public:
    Placer(std::vector<StateRec> &slots, uint32_t window) : slots_(slots), ring_(window) {}

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
        const daac_status st = trie.add(blob + offsets[i], static_cast<size_t>(offsets[i + 1] - offsets[i]),
                                        values ? values[i] : static_cast<uint32_t>(i));
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
