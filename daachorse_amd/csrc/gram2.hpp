// GRAM engine, second table set (host side): one LDS word per position for filter AND count.
//
// Same method as gram.hpp (no state chain: every position is looked at on its own), re-cut after the round-2
// pipe measurements (tools/micro/pipes_bench.hip): a wave-wide random LDS lookup costs 7.5-9 CU
// cycles whatever its width, so the step is priced in LOOKUPS per position, not bytes.  v1 spends three
// (CID u16 -> COMBO b64, B bit) plus a queue store per position column; here
//
//   M[K-gram]  one 32-bit word per K-gram g of byte classes:
//                bit d (1 <= d < C)  the (K+1)-gram (g, d) is a trie prefix  (class 0 = byte of no pattern: never set)
//                bits 30-31          number of patterns of length <= K that end after g  (0..3)
//              -> `find_overlapping_iter(..).count()` needs this ONE lookup per position;
//   CID[K-gram] u16 = 4 * id, H[id] u32 = sum of h32 of those patterns: read only when the checksum is wanted;
//   S[g >> 2]  popcount directory over the continuation bits (one entry per 4 words of M): the rank of bit (g, d)
//              among all set bits IS the offset of that depth-(K+1) state in breadth-first order (as in gram.hpp),
//              so a hit costs one 16-byte read of its group of M words + one directory entry, on hits only.
//
// Built from the automaton itself (breadth-first walk of the double array + the reference's own delta for the
// K-gram contexts); the walk records (drec / dhit / cfirst) are those of gram.hpp.  K = 3 when the tables fit the
// LDS budget, else 2.  Needs <= 30 byte classes, a tree-shaped trie, Standard kind, no "" pattern, at most 3 short
// patterns per context (only duplicate patterns can exceed that) — otherwise the v1 tables / other engines serve.
#pragma once

#include <cstdint>
#include <vector>

#include "repack.hpp"

namespace daac { constexpr uint32_t kGram2OffMHost = 512; }  // LDS offset of M (= kGram2OffM of device_tables.hpp)

namespace daac {

struct Gram2Tables {
    bool available = false;
    bool exact_available = false;   // CID/H fit next to M (ids < 16384, LDS budget)
    uint32_t K = 0, C = 0, N = 0;
    uint8_t unused_byte = 0;        // a byte of class 0
    bool has_short = false;         // any pattern of length <= K
    bool s16 = false;               // directory entries fit u16 (fewer than 65536 depth-(K+1) states)
    uint32_t level_start = 0;       // id of the first depth-(K+1) state
    std::vector<uint8_t> cls;       // 256
    std::vector<uint32_t> m;        // C^K (+ padding to a multiple of 4)
    std::vector<uint32_t> sdir;     // per 4 words of m: continuation bits set before the group
    std::vector<uint16_t> cid4;     // C^K: 4 * id of the context's {sum of h32} (0 = nothing ends here)
    std::vector<uint32_t> hsum;     // per id
    std::vector<U32x4> drec;        // N: {cmap, first_child, own_cnt, own_hsum}
    // The same for `.count()`, with every state below which the trie is a single path of 1 .. 8 edges (and no duplicate
    // patterns on it) replaced by a TAIL record {1 << 31 | edges | word_ends << 4 | class of the first path byte << 13, 0, path bytes 0-3, path bytes 4-7}: the
    // walker compares the path with the next eight haystack bytes in one step instead of fetching a record per edge.
    // word_ends bit i: the state i edges down the path is the end of a pattern (bit 0: the state itself).
    std::vector<U32x4> drec_c;
    std::vector<U32x2> dhit;        // depth-(K+1) states by rank: {cmap, own_hsum}; own_cnt == (own_hsum != 0)
    std::vector<U32x2> dhit_c;      // the same for `.count()`: {cmap | 1 if the state ends a pattern, first_child} (no second look-up)
    std::vector<uint32_t> cfirst;   // same order: first child id
    // gram3_kernels.hip, TAIL: tail records wherever the trie below a state is one path — from the hit records on (16 bytes each:
    // {cmap | ends-a-pattern, first_child, 0, 0} or a tail record) and in the walk records from depth K + 2 on
    std::vector<U32x4> dhit_t, drec_t;
    uint32_t lds_count = 0, lds_exact = 0;  // table bytes in LDS per mode (without the hit rings)

    // ---- tuple emission (gram2_emit_kernels.hip): the same lookups, but every match has to come out as (start, end, value) ----
    // Needs: no duplicates among the patterns of at most K bytes (a context then holds at most one pattern per length) and <= 29
    // classes (three flag bits in the word).  Patterns longer than K + 16 bytes and the further copies of longer duplicate patterns
    // do not fit the u16 of deep-match lengths per position: the kernel places them as "extras" (a slower path, per tile).
    bool emit_available = false;
    uint32_t max_len = 0;
    std::vector<uint32_t> me;       // C^K: continuation bits 1..28 as in m; bit 28 + len: a pattern of length len (1..3) ends after the context
    std::vector<uint32_t> v1, v2, v3;  // value of the pattern that IS the 1-/2-/3-gram (C, C^2, C^3 entries; v3 only for K = 3)
    std::vector<U32x4> erec;        // N: {cmap | own (bit 0), first_child, own_value, depth | further copies << 24}
    std::vector<U32x2> ehit;        // depth-(K+1) states by rank: {cmap | own (bit 0), own_value}
    std::vector<uint32_t> ecopies;  // the same order: further copies of a duplicate pattern (0 = none)
    std::vector<uint32_t> dupo, dupv;  // per state: offset into dupv of the values of those copies, in registration order
};

constexpr uint32_t kGram2MaskBits = 0x3fffffffu;  // continuation bits of an M word

// `lds_budget` = bytes the tables may take (rings excluded).  Returns false if the automaton does not qualify.
bool build_gram2_tables(const HostPma &p, uint32_t lds_budget, Gram2Tables &out);

}  // namespace daac
