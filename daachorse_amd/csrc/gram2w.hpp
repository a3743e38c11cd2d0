// GRAM engine for wide alphabets (host side): 31 .. 62 byte classes.
//
// The 32-bit tables (gram.hpp, gram2.hpp) stop at 30-31 classes, and an automaton beyond that used to fall to the DARRAY
// engine (one dependent L2 gather per byte and lane).  Here the same method runs on 64-bit words with a context of K = 2:
//   M[2-gram] u64: bit d (1 <= d < C) = the 3-gram (g, d) is a trie prefix; bits 62-63 = number of patterns of length <= 2
//   that end after g.  C^2 * 8 bytes: 30 KB at 62 classes.  CID / H and the popcount directory S as in gram2.hpp; the walk
//   records carry 64-bit child bitmaps: drec[s] = {cmap lo, cmap hi, first_child, own h32 sum}, dhit[rank] = {cmap lo,
//   cmap hi, own h32 sum, first_child} (one 16-byte read per hit).
// A context of two classes filters less than three (a hit is a depth-3 trie prefix), so this engine is slower than the
// 32-bit one on the same dictionary; it is there so that a mixed-case / punctuated dictionary does not lose an order of
// magnitude.  Needs a tree-shaped trie, Standard kind, no "" pattern, every state's own patterns visible in its h32 sum.
#pragma once

#include <cstdint>
#include <vector>

#include "repack.hpp"

namespace daac {

struct Gram2WTables {
    bool available = false, exact_available = false;
    uint32_t C = 0, N = 0, level_start = 0;
    uint8_t unused_byte = 0;
    std::vector<uint8_t> cls;       // 256
    std::vector<uint64_t> m;        // C^2 (+ padding to a multiple of 4)
    std::vector<uint32_t> sdir;     // per 4 words of m: continuation bits set before the group
    std::vector<uint16_t> cid4;     // C^2: 4 * id
    std::vector<uint32_t> hsum;     // per id
    std::vector<U32x4> drec;        // N: {cmap lo, cmap hi, first_child, own_hsum}
    std::vector<U32x4> dhit;        // depth-3 states by rank: {cmap lo, cmap hi, own_hsum, first_child}
    uint32_t lds_count = 0, lds_exact = 0;
};

constexpr uint64_t kGram2WMaskBits = 0x3fffffffffffffffull;

bool build_gram2w_tables(const HostPma &p, uint32_t lds_budget, Gram2WTables &out);

}  // namespace daac
