// Pieces shared by the bytewise and charwise host builders: the sparse trie with failure links
// and merged output lists (reference src/nfa_builder.rs, src/edge_map.rs) and the vacant-slot ring
// (reference src/build_helper.rs).
#pragma once

#include <algorithm>
#include <limits>
#include <utility>
#include <vector>

#include "pma.hpp"

namespace daac {
namespace build {

// ------------------------------------------------------------------------------------ sparse trie
template <class Label>
struct TrieNode {
    std::vector<std::pair<Label, uint32_t>> edges;    // sorted by label (edge_map.rs)
    std::vector<std::pair<uint32_t, uint32_t>> own;   // (value, length) of patterns ending here
    uint32_t fail = kRoot;
    uint32_t output_pos = 0;                           // 1-based head of the merged list, 0 = none

    const uint32_t *child(Label c) const {
        auto it = std::lower_bound(edges.begin(), edges.end(), c, [](const auto &e, Label key) { return e.first < key; });
        return (it != edges.end() && it->first == c) ? &it->second : nullptr;
    }
};

template <class Label>
class SparseTrie {
public:
    explicit SparseTrie(uint8_t kind) : kind_(kind), nodes_(2) {}  // node 0 = root, node 1 = dead

    // nfa_builder.rs:78-113; `byte_len` is what Output::length reports (bytes, also for char labels)
    daac_status add(const Label *pat, size_t len, uint64_t byte_len, uint32_t value) {
        if (byte_len > std::numeric_limits<uint32_t>::max()) {
            set_error("pattern.len() must be <= u32::MAX");
            return DAAC_ERR_INVALID_ARGUMENT;
        }
        uint32_t cur = kRoot;
        for (size_t i = 0; i < len; ++i) {
            // LeftmostFirst: nothing below an earlier-registered pattern can ever be reported
            if (kind_ == DAAC_LEFTMOST_FIRST && !nodes_[cur].own.empty()) return DAAC_OK;
            const Label c = pat[i];
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
            ed.insert(std::lower_bound(ed.begin(), ed.end(), c, [](const auto &e, Label key) { return e.first < key; }),
                      std::make_pair(c, fresh));
            nodes_.emplace_back();
            cur = fresh;
        }
        nodes_[cur].own.emplace_back(value, static_cast<uint32_t>(byte_len));
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
            TrieNode<Label> &s = nodes_[bfs_[qi]];
            if (leftmost && !s.own.empty()) s.fail = kDead;  // an output state never falls back
            for (const auto &e : s.edges) {
                nodes_[e.second].fail = leftmost ? fail_target_leftmost(s.fail, e.first) : fail_target(s.fail, e.first);
                bfs_.push_back(e.second);
            }
        }
    }

    // nfa_builder.rs:203-222: own outputs (first-registered first) followed by the fail state's list
    void merge_outputs(std::vector<OutputRec> &outputs) {
        auto emit = [&](TrieNode<Label> &s, uint32_t tail) {
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

    const std::vector<TrieNode<Label>> &nodes() const { return nodes_; }
    size_t num_patterns() const { return num_patterns_; }

private:
    uint32_t fail_target(uint32_t f, Label c) const {
        for (;;) {
            if (const uint32_t *t = nodes_[f].child(c)) return *t;
            const uint32_t up = nodes_[f].fail;
            if (f == kRoot && up == kRoot) return kRoot;
            f = up;
        }
    }
    uint32_t fail_target_leftmost(uint32_t f, Label c) const {
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
    std::vector<TrieNode<Label>> nodes_;
    std::vector<uint32_t> bfs_;
    size_t num_patterns_ = 0;
};

// ------------------------------------------------------------------------- vacant-slot bookkeeping
// Tracks, for the last `window` blocks only, which slots are taken, which BASE values are taken, and
// a circular list of the vacant slots in ascending order (build_helper.rs).  Indices are global;
// storage is a ring of window * 256 entries.
class VacantRing {
public:
    VacantRing(uint32_t block_len, uint32_t window) : bl_(block_len), window_(window), cap_(window * block_len), cell_(cap_) {}

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
        if (cap_ <= blocks_ * bl_) { blk = first_active_block(); return true; }
        return false;
    }

    // build_helper.rs:133-173
    daac_status append_block() {
        if (blocks_ * static_cast<uint64_t>(bl_) > std::numeric_limits<uint32_t>::max() - bl_) {
            set_error("num_elements must be <= u32::MAX");
            return DAAC_ERR_AUTOMATON_SCALE;
        }
        uint32_t closing;
        if (closing_block(closing)) {
            const uint32_t limit = (closing + 1) * bl_;  // retire what is still vacant there
            while (has_head_ && head_ < limit) take_index(head_);
        }
        const uint32_t lo = blocks_ * bl_, hi = lo + bl_;
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
        for (uint32_t b = blk * bl_; b < (blk + 1) * bl_; ++b)
            if (!base_taken(b)) { base = b; return true; }
        return false;
    }

private:
    struct Cell { uint32_t next = 0, prev = 0; bool base_taken = false, taken = false; };
    Cell &at(uint32_t i) { return cell_[i % cap_]; }
    const Cell &at(uint32_t i) const { return cell_[i % cap_]; }

    uint32_t bl_, window_, cap_;
    std::vector<Cell> cell_;
    uint32_t blocks_ = 0;
    uint32_t head_ = 0;
    bool has_head_ = false;
};


}  // namespace build
}  // namespace daac
