// GRAM engine, `.count()` tables in the numbering the round-5 kernel wants (gram4_kernels.hip); derived from gram2.hpp's.
//
// Why a second numbering: gram2.hpp numbers the byte classes 1 .. C-1 in byte order and keeps 0 for "a byte of no pattern".
// profiles/r04_pmc_sq.txt + tools/micro/pipes_bench.hip say the round-4 kernel sits on the LDS pipe as much as on the VALU
// (3.85 wave-lookups per haystack byte, one of them the 256-byte class table in the main path and 0.7 more in the hit path).
// With "no pattern" as the LAST class,
//
//      class(b) = min(b - lo, C - 1)           (two VALU: v_sub_u32 with an SDWA byte select, v_min_u32)
//
// is the whole map for every dictionary whose bytes are one contiguous range [lo, lo + C - 2] (a-z: cfg3) — no LDS access.
// Other dictionaries (<= 30 classes) keep a 256-byte table, in the same numbering, so one kernel serves both.
//
//   cls[256]   byte -> class; pattern bytes 0 .. C-2 in byte order, every other byte C-1
//   m[C^K]     bit d (0 <= d <= C-2): the (K+1)-gram (g, d) is a trie prefix; bits 30-31: patterns of length <= K ending after g
//   rfull[C^K] u16: continuation bits set in m[0 .. g)  (the rank of bit (g, d) is the offset of its depth-(K+1) state; present
//              when there are fewer than 65536 such states); sdir: the same per 4 words, u32
//   dhit_c     depth-(K+1) states by rank, 8 bytes: {cmap | ends-a-pattern << 30, first_child}
//   dhit_t     the same as 16-byte records; one path below the state: a tail record {1 << 31 | edges | word ends << 4 | first class << 13, 0, path bytes}
//   drec_c / drec_t   walk records {cmap, first_child, own_cnt, 0} or tail records (from depth K+3 / K+2 on), cmap bits 0 .. C-2
// Contexts holding the class "no pattern" have no continuation bits and every other context keeps its place among its peers, so ranks —
// and with them the order of the hit records — are those of gram2.hpp.
#pragma once

#include <cstdint>
#include <vector>

#include "gram2.hpp"
#include "gram4_filter.hpp"

namespace daac {

struct Gram4Tables {
    bool available = false;
    uint32_t K = 0, C = 0;
    bool arith = false;            // class(b) = min(b - lo, C - 1)
    uint32_t lo = 0;
    uint8_t unused_byte = 0;       // a byte of class C - 1
    bool s16 = false;              // rfull present
    std::vector<uint8_t> cls;      // 256
    std::vector<uint32_t> m;       // C^K, padded to a multiple of 4
    std::vector<uint16_t> rfull;   // per word (s16 only)
    std::vector<uint32_t> sdir;    // per 4 words
    std::vector<U32x2> dhit_c;
    std::vector<U32x4> dhit_t, drec_c, drec_t;
    // the filter in front of rank + gather (gram4_filter.hpp; build_gram4_filter): empty when it was not built
    std::vector<uint32_t> bloom;
    uint32_t filter_keys = 0;      // GO + ENDS keys in it
};

constexpr uint32_t kGram4EndsBit = 30;   // hit records: the depth-(K+1) state ends a pattern
constexpr uint32_t kGram4ChildBits = 0x3fffffffu;

// `g2` must be available.  Always succeeds for tables gram2.hpp accepts (<= 30 classes).
void build_gram4_tables(const Gram2Tables &g2, Gram4Tables &out);
// The Bloom array of gram4_filter.hpp over the keys the tables themselves name (M's continuation bits x the hit records' child maps and
// "ends a pattern" bits), in at most `max_bytes` of LDS.  Sized at 16 bits per key when there is room; not built (false) below 2 bits per key
// — it would pass nearly everything — or when a byte class stands for several bytes.
bool build_gram4_filter(Gram4Tables &t, uint32_t max_bytes);

}  // namespace daac
