// GRAM engine tables (host side): a count/checksum scanner without a per-lane state chain.
//
// find_overlapping_iter reports every occurrence of every pattern; the ORDER matters for the
// materialised stream but not for count + checksum.  For that mode the Aho-Corasick recurrence is
// replaced by direct context lookups, which removes the serial dependency between bytes (every
// position is independent, loads are coalesced, nothing to warm up):
//
//   * the occurrences of length <= K ending at position p are a function of the last K byte classes:
//       CID[last K classes] -> id (u16) of the distinct {count, sum of h32} combination, 0 = nothing;
//       COMBO[id]           -> {count, sum of h32} of all patterns of length <= K ending here
//   * an occurrence of length > K is found from its START: the (K+1)-gram bitmap B_{K+1} says
//     whether the K+1 bytes ending at p are a trie prefix; its rank (popcount directory, also in
//     LDS) IS the offset of the depth-(K+1) state in the breadth-first numbering, so one 16-byte
//     read fetches {child bitmap, first child, own count, own h32 sum}; the state's own patterns
//     are counted, and only if the next byte also continues in the trie (rare) a walker is queued
//     that follows the goto function (no failure links) to the end of the branch.
//
// Everything is derived from the TIERED re-pack (repack.hpp), i.e. from the reference's own
// arrays.  K = 3 when the tables fit the LDS budget (<= 27-28 byte classes), else K = 2.
#pragma once

#include <cstdint>
#include <vector>

#include "repack.hpp"

namespace daac {

struct GramTables {
    bool available = false;
    uint32_t K = 0, C = 0, N = 0;
    uint8_t unused_byte = 0;        // a byte of class 0 (occurs in no pattern)
    bool has_short = false;         // any pattern of length <= K
    uint32_t level_start = 0;       // id of the first depth-(K+1) state
    std::vector<uint8_t> cls;       // 256
    std::vector<uint16_t> cid;      // C^K: combination id of the K-gram, 0 = no pattern ends here
    std::vector<U32x2> combo;       // per id: {count, hsum} of the patterns of length <= K ending here
    std::vector<uint32_t> bbits;    // ceil(C^(K+1) / 32): (K+1)-gram is a trie prefix
    std::vector<uint8_t> brank;     // per PAIR of bbits words (64 bits): set bits before the pair within its 8-word superblock
    std::vector<uint32_t> bsuper;   // per 8 words (256 bits): set bits before the superblock
    std::vector<U32x4> drec;        // N: {cmap, first_child, own_cnt, own_hsum}; from depth K + 3 on a single path with one pattern end is a tail record
                                    //    {h of the pattern, 1 << 31 | edges | index of the ending node << 4, path bytes 0-3, path bytes 4-7} (gram.cpp)
    uint32_t n_tail = 0;            // tail records among them
    std::vector<U32x2> dhit;        // per depth-(K+1) state, in rank order: {cmap, own_hsum}; own_cnt == (own_hsum != 0)
    std::vector<uint32_t> cfirst;   // same order: id of the state's first child (read only when a branch goes on)
    uint32_t lds_bytes = 0;
};

// Derives the GRAM tables; returns false if the automaton does not qualify (then the TIERED /
// DARRAY engines are used).  `tier` must come from build_tier_tables on the same automaton.
bool build_gram_tables(const HostPma &p, const TierTables &tier, uint32_t lds_budget, GramTables &out);

}  // namespace daac
