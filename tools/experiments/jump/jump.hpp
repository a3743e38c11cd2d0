// JUMP engine (host side): find_iter of a Standard bytewise automaton without a state chain.
//
// FindIterator::next (reference src/bytewise/iter.rs:58-113) restarts at ROOT after every match and reports the head of the
// output list of the first state that has one.  Restarted at position e, the automaton's state after text[e..i) is the longest
// suffix of that text which is a trie path, so it has an output at i iff some pattern occurs at (s, i) with s >= e: the link
// that starts at e ends at
//        next(e) = min over s >= e of  s + L(s),      L(s) = length of the SHORTEST pattern that is a prefix of text[s..)
// and reports the longest pattern ending there with its start at or behind e — which, next(e) being minimal, is the shortest
// pattern starting at  s_min = the smallest s in [e, next(e)) with s + L(s) = next(e)  (its first-registered copy: the head of
// that state's own list, nfa_builder.rs:203-222).  L is a function of the POSITION alone, so it is computed for every
// position in parallel (jump_kernels.hip, pass A: K-gram table for patterns of up to K bytes, start-anchored walks over the
// trie for the rest), turned into per-position jumps {N = next(e) - e, D = s_min - e} by a suffix-minimum (pass B), and the
// chain through the matches is followed over that array — one 2-byte load per MATCH instead of a transition with failure
// links per BYTE (pass C: the speculate / reconcile / sum scheme of chain_scan.hpp over the new link function).
//
// Tables (all from the trie of the double array; the failure links are not used at all):
//   cls[256]        byte -> class (0 = byte of no pattern), C classes
//   MS[C^K]         bits 30-31: length (1..K) of the shortest pattern that is a prefix of the K-gram, 0 = none;
//                   bits 1..29 (only when that is 0): bit d = the (K+1)-gram (gram, d) is a trie path
//   SDIR            set continuation bits before every group of four MS words (the rank of a bit = index of its JHIT record)
//   JHIT[rank]      depth-(K+1) states in that order, JREC[state] every state breadth-first:
//                   {cmap | own (bit 0), first_child, h32 of the state's own first-registered pattern, depth}
//   H1 / H2 / H3    h32 of the first-registered pattern that IS the 1- / 2- / 3-gram
// Needs: Standard kind, no "" pattern, a tree-shaped trie, at most 29 byte classes with K = 3 (or 62 with K = 2 ... not built:
// K = 3 only), patterns of at most 127 bytes.
#pragma once

#include <cstdint>
#include <vector>

#include "repack.hpp"

namespace daac {

constexpr uint32_t kJumpMaxLen = 127;   // longest pattern: an escape (N = 255) then skips at least 128 bytes
constexpr uint32_t kJumpOffMS = 256;    // LDS offset of MS (classes at 0)

struct JumpTables {
    bool available = false;
    uint32_t K = 0, C = 0, N = 0;
    uint8_t unused_byte = 0;
    uint32_t max_len = 0;
    std::vector<uint8_t> cls;        // 256
    std::vector<uint32_t> ms;        // C^K, padded to a multiple of 4
    std::vector<uint32_t> sdir;      // ms.size() / 4
    std::vector<U32x4> jhit, jrec;
    std::vector<uint32_t> h1, h2, h3;
    uint32_t lds_bytes = 0;          // cls + MS + SDIR (without the hit queues)
};

bool build_jump_tables(const HostPma &p, JumpTables &out);

}  // namespace daac
