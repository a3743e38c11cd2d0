// GRAM engine, `.count()` tables renumbered for gram4_kernels.hip — see gram4.hpp.
#include "gram4.hpp"

#include <algorithm>

namespace daac {

void build_gram4_tables(const Gram2Tables &g2, Gram4Tables &out) {
    out = Gram4Tables{};
    if (!g2.available) return;
    const uint32_t K = g2.K, C = g2.C, OTH = C - 1;
    auto perm = [&](uint32_t c) -> uint32_t { return c == 0 ? OTH : c - 1; };
    out.K = K;
    out.C = C;
    out.unused_byte = g2.unused_byte;
    out.cls.resize(256);
    for (uint32_t b = 0; b < 256; ++b) out.cls[b] = static_cast<uint8_t>(perm(g2.cls[b]));
    // one contiguous range of pattern bytes?  (classes are handed out in byte order, so class(lo) = 0 settles the rest)
    {
        uint32_t lo = 256;
        for (uint32_t b = 0; b < 256; ++b) if (out.cls[b] == 0) lo = b;
        bool ok = lo < 256;
        for (uint32_t b = 0; b < 256 && ok; ++b) {
            const uint32_t u = b - lo;  // wraps for b < lo
            ok = out.cls[b] == (u < OTH ? u : OTH);
        }
        out.arith = ok;
        out.lo = ok ? lo : 0;
    }
    // M in the new index order
    uint64_t ngram = 1;
    for (uint32_t i = 0; i < K; ++i) ngram *= C;
    const uint32_t nm = static_cast<uint32_t>((ngram + 3) & ~3ull);
    out.m.assign(nm, 0);
    for (uint32_t g = 0; g < ngram; ++g) {
        uint32_t rest = g, gn = 0, mul = 1;
        for (uint32_t i = 0; i < K; ++i) {  // least significant class first
            gn += perm(rest % C) * mul;
            mul *= C;
            rest /= C;
        }
        const uint32_t w = g2.m[g];
        out.m[gn] = (w & 0xc0000000u) | ((w & kGram2MaskBits) >> 1);
    }
    out.s16 = g2.s16;
    out.sdir.assign(nm / 4, 0);
    if (out.s16) out.rfull.assign(nm, 0);
    uint32_t run = 0;
    for (uint32_t g = 0; g < nm; ++g) {
        if ((g & 3) == 0) out.sdir[g >> 2] = run;
        if (out.s16) out.rfull[g] = static_cast<uint16_t>(run);
        run += static_cast<uint32_t>(__builtin_popcount(out.m[g] & kGram4ChildBits));
    }
    auto hit_x = [&](uint32_t x) -> uint32_t { return ((x & kGram2MaskBits) >> 1) | ((x & 1u) << kGram4EndsBit); };
    auto tail_x = [&](uint32_t x) -> uint32_t { return (x & ~(0x1fu << 13)) | (perm((x >> 13) & 0x1fu) << 13); };
    out.dhit_c.resize(g2.dhit_c.size());
    for (size_t i = 0; i < g2.dhit_c.size(); ++i) out.dhit_c[i] = U32x2{hit_x(g2.dhit_c[i].x), g2.dhit_c[i].y};
    out.dhit_t.resize(g2.dhit_t.size());
    for (size_t i = 0; i < g2.dhit_t.size(); ++i) {
        const U32x4 r = g2.dhit_t[i];
        out.dhit_t[i] = (r.x >> 31) ? U32x4{tail_x(r.x), r.y, r.z, r.w} : U32x4{hit_x(r.x), r.y, 0u, 0u};
    }
    auto walk = [&](const std::vector<U32x4> &src, std::vector<U32x4> &dst) {
        dst.resize(src.size());
        for (size_t i = 0; i < src.size(); ++i) {
            const U32x4 r = src[i];
            dst[i] = (r.x >> 31) ? U32x4{tail_x(r.x), r.y, r.z, r.w} : U32x4{(r.x & kGram2MaskBits) >> 1, r.y, r.z, 0u};
        }
    };
    walk(g2.drec_c, out.drec_c);
    walk(g2.drec_t, out.drec_t);
    out.available = true;
}

bool build_gram4_filter(Gram4Tables &t, uint32_t max_bytes) {
    t.bloom.clear();
    t.filter_keys = 0;
    if (!t.available) return false;
    const uint32_t K = t.K, C = t.C, OTH = C - 1;
    // class -> its byte (the keys are hashed from text bytes): every pattern class must be exactly one byte value
    uint32_t byte_of[32];
    uint32_t seen[32] = {0};
    for (uint32_t b = 0; b < 256; ++b) {
        const uint32_t c = t.cls[b];
        if (c < OTH) { byte_of[c] = b; if (++seen[c] > 1) return false; }
    }
    for (uint32_t c = 0; c < OTH; ++c) if (seen[c] != 1) return false;
    uint64_t ngram = 1;
    for (uint32_t i = 0; i < K; ++i) ngram *= C;
    // the keys: one pass to count, one to insert
    uint64_t keys = 0;
    {
        uint32_t rank = 0;
        for (uint64_t g = 0; g < ngram; ++g)
            for (uint32_t w = t.m[g] & kGram4ChildBits; w != 0; w &= w - 1, ++rank) {
                const uint32_t x = t.dhit_c[rank].x;
                keys += static_cast<uint64_t>(__builtin_popcount(x & kGram4ChildBits)) + ((x >> kGram4EndsBit) & 1u);
            }
    }
    if (keys == 0) return false;
    uint64_t words = std::min<uint64_t>(max_bytes / 4, (keys * 16 + 31) / 32);
    words &= ~3ull;   // (staged in 16-byte pieces)
    if (words >= (1u << 14)) words = (1u << 14) - 4;   // (the word index is an 18 x 14-bit product)
    if (words < 64 || words * 32 < keys * 2) return false;
    t.bloom.assign(static_cast<size_t>(words), 0u);
    const uint32_t W = static_cast<uint32_t>(words);
    uint32_t rank = 0;
    for (uint64_t g = 0; g < ngram; ++g) {
        uint32_t w = t.m[g] & kGram4ChildBits;
        if (w == 0) continue;
        // the K context bytes, oldest first = lowest byte of x (the most significant class of the index is the oldest)
        uint32_t xc = 0;
        {
            uint64_t rest = g;
            for (uint32_t i = 0; i < K; ++i) {   // least significant class = the newest context byte
                xc |= byte_of[rest % C] << (8 * (K - 1 - i));
                rest /= C;
            }
        }
        for (; w != 0; w &= w - 1, ++rank) {
            const uint32_t d = static_cast<uint32_t>(__builtin_ctz(w));
            const uint32_t x = xc | (byte_of[d] << (8 * K));
            const uint32_t rx = t.dhit_c[rank].x;
            if ((rx >> kGram4EndsBit) & 1u) { const G4Probe p = g4f_probe(x, 0, W); t.bloom[p.word] |= p.ends; }
            for (uint32_t cm = rx & kGram4ChildBits; cm != 0; cm &= cm - 1) {
                const G4Probe p = g4f_probe(x, byte_of[__builtin_ctz(cm)], W);
                t.bloom[p.word] |= p.go;
            }
        }
    }
    t.filter_keys = static_cast<uint32_t>(std::min<uint64_t>(keys, 0xffffffffull));
    return true;
}

}  // namespace daac
