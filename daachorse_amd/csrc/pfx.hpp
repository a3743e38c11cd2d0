// PFX engine (host side): `.count()` of find_overlapping_iter for bytewise Standard automata over ANY byte alphabet.
//
// The GRAM engines enumerate every K-gram of byte CLASSES in LDS, which ties them to dictionaries of at most 62 distinct
// bytes.  The transition function they replace has no such limit (reference src/bytewise.rs:1063-1088), and two of the four
// dictionaries the crate publishes numbers for are wide (UTF-8 Japanese, o200k_base: figures/overlapping.txt).  PFX looks at
// occurrences from their START instead (SURVEY §8a note C: a start-parallel formulation is exact for the overlapping stream):
//
//   G            = min(length of the shortest pattern of two or more bytes, 6): every such pattern begins with a depth-G trie path
//   BLOOM        two bits of one 32-bit word per hashed G-gram that is a depth-G trie path (as much of the LDS as the other tables
//                leave, one lookup per haystack byte): a position whose next G bytes fail it starts no pattern of two or more bytes;
//   CNT1[256]    patterns of ONE byte, counted per haystack byte straight from LDS (only when the dictionary has any);
//   DISP / SLOTS a hash-and-displace perfect hash of the depth-G paths: DISP (u16 per bucket, LDS) + ONE 16-byte record from L2
//                {bytes 0-3, bytes 4-5, BASE of the depth-G state, patterns that ARE this path}: a survivor of the filter is
//                settled by that one gather — a false positive shows as a key mismatch.  Where the trie below the path is a
//                single path of at most eight edges (most paths of a word list) the record is a TAIL record instead,
//                {bytes 0-3, bytes 4-5 | edges | pattern ends, path bytes 0-3, path bytes 4-7}: one compare with the text
//                behind the key settles everything below it, no walk;
//   WREC         the double array itself as 8-byte records {BASE, CHECK | patterns that end in this state << 8}: the branches
//                that go on below depth G walk it goto-only (child = BASE ^ byte, CHECK == byte, bytewise.rs:1070-1077),
//                no failure links: an occurrence is a root path, and every state met on it adds the patterns that end there.
//
// The count is exact: every occurrence starts at exactly one position, is found from there, and is counted once with its
// multiplicity (duplicate patterns: `own` > 1).  Needs: Standard kind, no "" pattern, at least one pattern of two or more bytes.
#pragma once

#include <cstdint>
#include <vector>

#include "repack.hpp"

namespace daac {

constexpr uint32_t kPfxMulBloom0 = 0x9E3779B1u, kPfxMulBloom1 = 0x85EBCA77u;   // m = k0 * A0 + k1 * A1: Bloom word (uint64(m) * words) >> 32; bits (m2 >> 27) and (m2 >> 22) & 31 of m2 = m * kPfxMulBits
constexpr uint32_t kPfxMulBucket0 = 0xC2B2AE3Du, kPfxMulBucket1 = 0x27D4EB2Fu; // bucket of DISP:         (uint64(mb) * buckets) >> 32
constexpr uint32_t kPfxMulSlot0 = 0x165667B1u, kPfxMulSlot1 = 0xD3A2646Du;     // slot: pfx_slot(ms, displacement of the key's bucket, slots)
constexpr uint32_t kPfxMulBits = 0x2C1B3C6Du, kPfxBit1 = 27, kPfxBit2 = 22;
constexpr uint32_t kPfxEmpty = 0x80000000u, kPfxTail = 0x40000000u;            // flags in word 1 of a slot record

struct PfxTables {
    bool available = false;
    uint32_t G = 0;                  // bytes of a key
    bool has_len1 = false;           // the dictionary has one-byte patterns (CNT1 is looked at)
    uint32_t bloom_words = 0;        // words of BLOOM (a multiple of 4)
    uint32_t buckets = 0;            // entries of DISP
    uint32_t n_slots = 0;            // entries of SLOTS
    uint32_t seed = 0;               // xor-ed into the second operand of the bucket / slot hashes (changed until the displacement search succeeds)
    uint32_t n_keys = 0;
    uint32_t n_distinct_bytes = 0;   // distinct byte values on the trie's edges (set even when the tables are declined after the walk)
    std::vector<uint32_t> bloom;
    std::vector<uint16_t> cnt1;      // 256
    std::vector<uint16_t> disp;
    // {k0, k1 | own << 16, base, filter}; an empty slot: k1 = kPfxEmpty; a tail record: {k0, k1 | kPfxTail | edges << 16 | ends << 20, path lo, path hi}.
    // own = patterns that ARE the key (< 16384); filter: bit pfx_pair_bit(c0, c1) is set for every two-byte path (c0, c1) below the key, all
    // bits when a pattern ends one byte below it — a branch whose next two text bytes miss the filter ends nowhere: no walker is queued
    std::vector<U32x4> slots;
    uint32_t n_tails = 0;
    std::vector<U32x2> wrec;         // per double-array slot {base, check | own << 8}
    // count + checksum (the kernel's EXACT variant): no tail records and no path filter — every match has to be met as its own state, because
    // its h32 goes into the sums with its own end position
    std::vector<U32x4> slots_x;      // {k0, k1 | own << 16, base, sum of h32 of the patterns that are the key}
    std::vector<U32x4> wrec_x;       // per double-array slot {base, check | own << 8, sum of h32 of the patterns that end there, VALUE of the pattern that ends there (own == 1)}
    std::vector<uint32_t> hs1;       // 256: sum of h32 of the one-byte patterns
    // tuple emission (pfx_emit_detect_kernel + EXPAND in its raw-haystack mode): every match is logged with its value, so no pattern may be
    // registered twice (emit_ok).  slots_e = slots_x with the key pattern's VALUE in word 3; wrec_x's word 3; v1 / has1 for the one-byte patterns
    bool emit_ok = false;
    std::vector<U32x4> slots_e;
    std::vector<uint32_t> v1;        // 256: value of the one-byte pattern
    std::vector<uint8_t> has1;       // 256: 0x20 where the byte is a pattern (the flag bit of EXPAND's stream bytes), else 0
    uint32_t lds_tables = 0;         // BLOOM + CNT1 + DISP bytes
};

// The slot of a key under displacement d: a fresh pseudo-random place for every d (two keys of one bucket that collide under one d
// part again under the next; with slot = home + d they would stay together for every d)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t pfx_slot(uint32_t ms, uint32_t d, uint32_t n_slots) {
    uint32_t x = ms + d * 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    return static_cast<uint32_t>((static_cast<uint64_t>(x) * n_slots) >> 32);
}

#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t pfx_pair_bit(uint32_t two_bytes) { return ((two_bytes & 0xffffu) * 0x9E3779B1u) >> 27; }

inline uint32_t pfx_mask0(uint32_t G) { return G >= 4 ? 0xffffffffu : ((1u << (8 * G)) - 1u); }
inline uint32_t pfx_mask1(uint32_t G) { return G <= 4 ? 0u : ((1u << (8 * (G - 4))) - 1u); }

// `lds_budget`: bytes the three LDS tables may take.  Returns false if the automaton does not qualify.
bool build_pfx_tables(const HostPma &p, uint32_t lds_budget, PfxTables &out);

}  // namespace daac
