// GPU re-pack of a daachorse automaton (host side).
//
// The reference keeps one 12-byte {BASE, FAIL, OUTPUT_POS|CHECK} record per double-array slot
// (src/bytewise.rs:1131-1137, src/intpack.rs:31-54) — ideal for one CPU core with a big L2, poor
// for 64-lane wavefronts: children of one state are XOR-scattered over a multi-MB array, so no
// useful subset fits the 160 KB LDS of a CU.  Two device layouts are derived from it:
//
//  DARRAY  the double array itself, split hot {base, opos_ch} (8 B) / cold fail (4 B) — the same
//          split the reference uses for leftmost automata (bytewise.rs:61-63).  Works for every
//          automaton; every transition is a random read of L2/MALL.
//
//  TIERED  states renumbered breadth-first so the children of a state are contiguous, each state
//          holding a child bitmap over byte *classes* (bytes that occur in no pattern share
//          class 0):  child(s, k) = first_child[s] + popcount(cmap[s] & ((1 << k) - 1)).
//            tier A  ids [0, NA)   depth <= DA: dense rows delta(s, k), failure links already
//                                  resolved, in LDS (one lookup per byte, no chain);
//            tier B  ids [NA, NB)  child bitmap + fail in LDS;
//            tier C  ids [NB, N)   16-byte {cmap, omap, first_child, fail} records in HBM/L2.
//          `omap` marks children that carry an output so the scanner knows without touching the
//          child.  Needs <= 32 byte classes and a tree-shaped trie; otherwise DARRAY is used.
//
// Both are pure functions of (trie, failure links, output lists), which is all the match tuples
// depend on (SURVEY.md §8a note F), and both are built by evaluating the reference's own
// transition function on the host.
#pragma once

#include <cstdint>
#include <vector>

#include "pma.hpp"

namespace daac {

struct U32x2 { uint32_t x, y; };
struct U32x4 { uint32_t x, y, z, w; };

// Per-output-list aggregates, indexed by output_pos - 1 (chain = the `parent` linked list,
// lib.rs:213-218): number of records on the chain and the sum of their h32 hashes.
struct OutSum { uint32_t cnt, hsum; };

uint32_t match_hash32(uint32_t value, uint32_t length);  // h of the checksum definition

struct DArrayTables {
    std::vector<U32x2> hot;        // {base, opos_ch} per slot
    std::vector<uint32_t> fail;    // per slot: the automaton's own failure links (leftmost kinds: DEAD = 1 marks "stop")
    std::vector<uint32_t> fail_plain;  // leftmost kinds only: classic Aho-Corasick failure links of the same trie,
                                       // used to find positions no occurrence spans (restart-scan sync points)
    std::vector<U32x4> root;       // 256 x {child idx, child base, child opos_ch, child fail}; no child: the ROOT record
    std::vector<OutSum> osum;      // per output record
    // chain walkers: bit (c & 31) of fmap[s] is set iff s has a child on some byte c' with c' & 31 == c & 31 — a probe of base ^ c that
    // the filter rules out is a miss without a memory request (failed probes land anywhere in the array: they were the walkers'
    // cache misses)
    std::vector<uint32_t> fmap;
    // ROOT's row for the chain walkers: {child, child.base, child.output_pos << 8 | child.fail (0 = ROOT, 1 = DEAD: a child of ROOT has
    // no other link; its CHECK byte is the index itself), child's filter}
    std::vector<U32x4> root_chain;
};

struct TierTables {
    bool available = false;
    uint32_t C = 0;                // classes incl. class 0
    uint32_t N = 0, NA = 0, NB = 0;
    uint32_t dense_depth = 0;
    bool row32 = false;            // row entries are u32 (id | flag<<31) instead of u16 (id | flag<<15)
    std::vector<uint8_t> cls;      // 256: byte -> class
    std::vector<uint16_t> rows16;  // NA x C
    std::vector<uint32_t> rows32;
    std::vector<uint32_t> bcmap;   // NB - NA
    std::vector<uint32_t> bfail;   // NB - NA   (new ids)
    std::vector<U32x4> grec;       // N: {cmap, omap, first_child, fail}
    std::vector<OutSum> ssum;      // N: per STATE {cnt, hsum} of its output list (0,0 if none)
    std::vector<uint32_t> sopos;   // N: output_pos (1-based) per state
    uint32_t lds_bytes() const;    // rows + bcmap + bfail + ssum[0..NA) + cls
    uint32_t root_flag = 0;        // root has an output ("" is a pattern)
    std::vector<uint32_t> old_of_new;  // N: double-array index of each renumbered state (host only)
};

struct RepackOptions {
    uint32_t lds_budget = 96 * 1024;  // bytes of LDS the tier tables may take per workgroup
    int dense_depth = -1;             // force DA (>= 0), -1 = choose
    uint32_t rows_share_pct = 45;     // at most this share of the budget goes to dense rows
};

void build_darray_tables(const HostPma &p, DArrayTables &out);
// Returns false (and leaves out.available = false) if the automaton does not qualify.
bool build_tier_tables(const HostPma &p, const RepackOptions &opt, TierTables &out);

}  // namespace daac
