// Host-side container of a CharwiseDoubleArrayAhoCorasick<u32> (reference src/charwise.rs:59-65):
// 16-byte states {base, check = PARENT index, fail, output_pos} (charwise.rs:1096-1101), the code
// mapper (src/charwise/mapper.rs: code point -> dense code by frequency), the output lists.
// Parsing / validation follow charwise.rs:896-952; construction (charwise_builder.cpp) follows
// src/charwise/builder.rs and produces byte-identical blobs.
#pragma once

#include <cstdint>
#include <vector>

#include "pma.hpp"
#include "repack.hpp"

namespace daac {

constexpr uint32_t kInvalidCode = 0xffffffffu;  // mapper.rs:7

struct CStateRec {  // charwise.rs:1096-1101, serialised base, check, fail, output_pos (:1162-1167)
    uint32_t base, check, fail, output_pos;
};

struct HostCharPma {
    std::vector<CStateRec> states;
    std::vector<uint32_t> table;   // code point -> code, kInvalidCode = unmapped (mapper.rs:10-13)
    uint32_t alphabet_size = 0;
    std::vector<OutputRec> outputs;
    uint8_t match_kind = DAAC_STANDARD;
    uint32_t num_states = 0;

    bool is_standard() const { return match_kind == DAAC_STANDARD; }
    size_t heap_bytes() const { return states.size() * 16 + table.size() * 4 + outputs.size() * 12; }  // charwise.rs:813-817
    uint32_t max_pattern_len() const;  // in BYTES (Output::length counts bytes, nfa_builder.rs:26-30)
    uint32_t code_of(uint32_t cp) const { return cp < table.size() ? table[cp] : kInvalidCode; }  // mapper.rs:36-42

    daac_status validate() const;                    // charwise.rs:912-950
    void serialize(std::vector<uint8_t> &out) const; // charwise.rs:831-848
    static daac_status deserialize(const uint8_t *src, size_t len, HostCharPma &out, size_t *consumed);
};

// src/charwise/builder.rs:178-359 on the host CPU; patterns are valid UTF-8
daac_status build_charwise(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n, uint8_t match_kind,
                           uint32_t num_free_blocks, HostCharPma &out);

// Device tables of the charwise engine (host copies)
struct CharTables {
    std::vector<CStateRec> states;       // one 16-byte record per slot (Standard: FAIL = DEAD of vacant slots -> ROOT)
    std::vector<uint32_t> fail_plain;    // leftmost kinds: classic failure links (sync points), else empty
    std::vector<uint32_t> table;         // mapper
    std::vector<OutSum> osum;            // per output record: {chain count, chain h32 sum}
    uint32_t root_flag = 0;
};
void build_char_tables(const HostCharPma &p, CharTables &out);

}  // namespace daac
