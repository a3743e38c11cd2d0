// Host-side automaton container for the MI355X scan path.
//
// Holds exactly the arrays of DoubleArrayAhoCorasick<u32> (reference src/bytewise.rs:54-68) as
// they arrive over the C ABI, validates them the way ::deserialize does (bytewise.rs:892-963)
// and exposes the two transition functions on the host — used only to derive the device
// re-pack (dense LDS rows are δ evaluated ahead of time), never to scan a haystack.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/daachorse_amd.h"

namespace daac {

constexpr uint32_t kRoot = 0;      // bytewise.rs:25
constexpr uint32_t kDead = 1;      // bytewise.rs:27
constexpr uint32_t kBlockLen = 256;

struct StateRec {   // State<u32>, bytewise.rs:1131-1137
    uint32_t base, fail, opos_ch;
};
struct LStateRec {  // State<Empty>
    uint32_t base, opos_ch;
};
struct OutputRec {  // Output<u32>, lib.rs:213-218
    uint32_t value, length, parent;
};

inline uint32_t output_pos_of(uint32_t opos_ch) { return opos_ch >> 8; }            // intpack.rs:34-36
inline uint8_t check_of(uint32_t opos_ch) { return static_cast<uint8_t>(opos_ch); }  // intpack.rs:39-41

void set_error(const std::string &msg);
const char *last_error_cstr();

// True when following `fail_at` from every slot ends at ROOT (or, with `dead_stops`, at DEAD): the
// transition loops of the reference (and of the kernels) only terminate on such links.  The
// reference's deserialize does not check this and would spin on a cyclic blob; a GPU must not.
bool fail_links_terminate(size_t n, const uint32_t *fail_at, size_t stride_words, bool dead_stops);

struct HostPma {
    std::vector<StateRec> states;        // Standard only
    std::vector<uint32_t> root_table;    // 256 entries, Standard only (bytewise.rs:1040-1056)
    std::vector<LStateRec> lstates;      // leftmost kinds only
    std::vector<uint32_t> fails;         // leftmost kinds only
    std::vector<OutputRec> outputs;
    uint8_t match_kind = DAAC_STANDARD;
    uint32_t num_states = 0;

    bool is_standard() const { return match_kind == DAAC_STANDARD; }
    size_t states_len() const { return is_standard() ? states.size() : lstates.size(); }
    size_t heap_bytes() const;           // bytewise.rs:764-770
    uint32_t max_pattern_len() const;

    // Accessors that work for both kinds (index into whichever array is populated).
    uint32_t base(uint32_t i) const { return is_standard() ? states[i].base : lstates[i].base; }
    uint32_t opos_ch(uint32_t i) const { return is_standard() ? states[i].opos_ch : lstates[i].opos_ch; }
    uint32_t fail(uint32_t i) const { return is_standard() ? states[i].fail : fails[i]; }

    // δ with failure links: bytewise.rs:1063-1088 (Standard) and 1094-1128 (leftmost).
    uint32_t next_state(uint32_t s, uint8_t c) const;
    uint32_t next_state_leftmost(uint32_t s, uint8_t c) const;

    void build_root_table();                         // bytewise.rs:1040-1056
    daac_status validate() const;                    // bytewise.rs:892-963
    void serialize(std::vector<uint8_t> &out) const; // bytewise.rs:801-820
    static daac_status deserialize(const uint8_t *src, size_t len, HostPma &out, size_t *consumed);  // 868-964
};

// bytewise/builder.rs:152-244 on the host CPU (builder.cpp)
daac_status build_bytewise(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n,
                           uint8_t match_kind, uint32_t num_free_blocks, HostPma &out);

}  // namespace daac
