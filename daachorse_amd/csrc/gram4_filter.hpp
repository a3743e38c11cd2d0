// GRAM engine, `.count()` kernel (gram4_kernels.hip): the LDS filter in front of rank + gather (round 6).
//
// A hit of the count kernel — the (K+1)-gram ending at byte p is a trie prefix — is worth a record from the L2 only if the depth-(K+1)
// state ENDS a pattern or the (K+2)-gram ending at p + 1 is a trie prefix as well; on cfg3's uniform text 83 % of the hits are neither
// (0.110 hits per byte, 0.013 that end a pattern, 0.006 that go on).  The filter is a Bloom array of `words` 32-bit words holding two kinds
// of keys, both hashed from RAW text bytes (a byte of no pattern never occurs in a key, so it needs no class):
//   GO    x = the K+1 bytes p-K .. p (first byte lowest), y = byte p+1        two bits of the word x names
//   ENDS  x alone: the depth-(K+1) state ends a pattern                      four bits of the same word (few keys; their false positives add to GO's)
// No false negatives by construction (tests/native/gram4_check.cpp walks every hit of a text through it); what passes is re-compacted and
// only then ranked (coarse directory) and gathered.  Shared by the table builder (gram4.cpp), the kernel and the CPU check.
#pragma once
#include <cstdint>

#if defined(__HIP__) || defined(__HIPCC__)
#define DAAC_G4F_HD __host__ __device__
#else
#define DAAC_G4F_HD
#endif

namespace daac {

// ONE word per (K+1)-gram: both keys of a hit look at the same word, so a hit costs one LDS read; the arithmetic is 24-bit multiplies, shifts
// and a rotate — the integer ops gfx950 issues at full rate (the first form of this filter hashed with v_mul_lo / v_mul_hi_u32, quarter rate,
// and read two words: its filter stage took 0.83 ms per 4 GiB where the whole consumer it replaced took 0.53, profiles/r06_gram4_decomposition.txt).
struct G4Probe { uint32_t word, go, ends; };   // index of the word; the GO key's two bits; the ENDS key's four bits

DAAC_G4F_HD inline uint32_t g4f_mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }   // v_mul_u32_u24: low 32 bits of a 24 x 24 product
// x = the K+1 bytes p-K .. p (first byte lowest), y = byte p+1; `words` < 2^14
DAAC_G4F_HD inline G4Probe g4f_probe(uint32_t x, uint32_t y, uint32_t words) {
    const uint32_t h = g4f_mul24(x, 0x9E3779u) + g4f_mul24(x >> 24, 0x85EBCBu);   // (a further h ^= h >> 15 bought 1 % fewer passes for two instructions: dropped)
    const uint32_t g = g4f_mul24(y, 0x2545F5u) + h;
    G4Probe p;
    p.word = g4f_mul24(h >> 14, words) >> 18;
    p.go = (1u << (g & 31u)) | (1u << ((g >> 5) & 31u));
    const uint32_t r = (h >> 9) & 31u;
    p.ends = (0x00420811u >> r) | (0x00420811u << ((32u - r) & 31u));   // four bits, pairwise different distances, rotated (v_alignbit_b32)
    return p;
}

}  // namespace daac
