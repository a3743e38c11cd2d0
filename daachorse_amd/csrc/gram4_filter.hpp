// GRAM engine, `.count()` kernel (gram4_kernels.hip): the LDS filter in front of rank + gather (round 6).
//
// A hit of the count kernel — the (K+1)-gram ending at byte p is a trie prefix — is worth a record from the L2 only if the depth-(K+1)
// state ENDS a pattern or the (K+2)-gram ending at p + 1 is a trie prefix as well; on cfg3's uniform text 83 % of the hits are neither
// (0.110 hits per byte, 0.013 that end a pattern, 0.006 that go on).  The filter is a Bloom array of `words` 32-bit words holding two kinds
// of keys, both hashed from RAW text bytes (a byte of no pattern never occurs in a key, so it needs no class):
//   GO    x = the K+1 bytes p-K .. p (first byte lowest), y = byte p+1        two bits of one word
//   ENDS  x alone: the depth-(K+1) state ends a pattern                      four bits of one word (few keys; their false positives add to GO's)
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

struct G4Probe { uint32_t word, mask; };

DAAC_G4F_HD inline uint32_t g4f_mulhi(uint32_t a, uint32_t b) { return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32); }
// first round, shared by the two keys of a hit
DAAC_G4F_HD inline uint32_t g4f_base(uint32_t x) {
    uint32_t h = x * 0x9E3779B1u;
    return h ^ (h >> 15);
}
DAAC_G4F_HD inline G4Probe g4f_go(uint32_t base, uint32_t y, uint32_t words) {
    uint32_t h = (base + (y + 1u) * 0x7FEB352Du) * 0x846CA68Bu;
    h ^= h >> 16;
    return G4Probe{g4f_mulhi(h, words), (1u << (h & 31u)) | (1u << ((h >> 5) & 31u))};
}
DAAC_G4F_HD inline G4Probe g4f_ends(uint32_t base, uint32_t words) {
    uint32_t h = (base + 0x3C6EF372u) * 0x846CA68Bu;
    h ^= h >> 16;
    return G4Probe{g4f_mulhi(h, words), (1u << (h & 31u)) | (1u << ((h >> 5) & 31u)) | (1u << ((h >> 10) & 31u)) | (1u << ((h >> 15) & 31u))};
}

}  // namespace daac
