// Host-logic test for the PFX tables, pfx.hpp (no GPU needed): counts the find_overlapping stream from the tables, position by
// position with the rules of pfx_kernels.hip (Bloom bit, displacement, slot record, goto-only walk over WREC, CNT1), and
// compares with the literal automaton walk on the original double array; then the same for the count + checksum tables
// (slots_x / wrec_x / hs1: every match met as its own state) against the sums of h32 and h32 * end of that walk.
//   usage: pfx_check <blob> <lds_budget> <haystack-file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/pfx.hpp"
#include "../../daachorse_amd/csrc/pma.hpp"

using namespace daac;

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const std::vector<uint8_t> blob = slurp(argv[1]);
    HostPma p;
    if (HostPma::deserialize(blob.data(), blob.size(), p, nullptr) != DAAC_OK) { std::printf("BADBLOB\n"); return 1; }
    PfxTables g;
    if (!build_pfx_tables(p, static_cast<uint32_t>(std::atoi(argv[2])), g)) { std::printf("UNAVAILABLE pfx\n"); return 0; }
    const std::vector<uint8_t> hay = slurp(argv[3]);
    const long long n = static_cast<long long>(hay.size());

    uint64_t rc = 0;
    uint32_t st = 0, r1 = 0, r2 = 0;
    for (long long i = 0; i < n; ++i) {
        st = p.next_state(st, hay[i]);
        for (uint32_t op = output_pos_of(p.states[st].opos_ch); op != 0; op = p.outputs[op - 1].parent) {
            rc++;
            const uint32_t h = match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            r1 += h;
            r2 += h * static_cast<uint32_t>(i + 1);
        }
    }

    const uint32_t G = g.G, M = g.n_slots;
    uint64_t gc = 0, survivors = 0, false_pos = 0, walkers = 0;
    for (long long s = 0; s < n; ++s) {
        if (g.has_len1) gc += g.cnt1[hay[s]];
        if (s + G > n) continue;
        uint32_t k0 = 0, k1 = 0;
        for (uint32_t i = 0; i < G; ++i) {
            if (i < 4) k0 |= static_cast<uint32_t>(hay[s + i]) << (8 * i); else k1 |= static_cast<uint32_t>(hay[s + i]) << (8 * (i - 4));
        }
        const uint32_t m = k0 * kPfxMulBloom0 + k1 * kPfxMulBloom1;
        const uint32_t word = g.bloom[static_cast<uint32_t>((static_cast<uint64_t>(m) * g.bloom_words) >> 32)];
        const uint32_t m2 = m * kPfxMulBits;
        if (!((word >> ((m2 >> kPfxBit1) & 31u)) & (word >> ((m2 >> kPfxBit2) & 31u)) & 1u)) continue;
        ++survivors;
        const uint32_t mb = k0 * kPfxMulBucket0 + (k1 ^ g.seed) * kPfxMulBucket1, ms = k0 * kPfxMulSlot0 + (k1 ^ g.seed) * kPfxMulSlot1;
        const uint32_t bucket = static_cast<uint32_t>((static_cast<uint64_t>(mb) * g.buckets) >> 32);
        const U32x4 r = g.slots[pfx_slot(ms, g.disp[bucket], M)];
        if (r.x != k0 || (r.y & 0xffffu) != k1 || (r.y & kPfxEmpty)) { ++false_pos; continue; }
        long long vn = s + G;
        if (r.y & kPfxTail) {
            const uint32_t edges = (r.y >> 16) & 15u, ends = (r.y >> 20) & 0x1ffu;
            uint32_t same = 0;
            while (same < edges && vn + same < n && hay[vn + same] == ((same < 4 ? r.z >> (8 * same) : r.w >> (8 * (same - 4))) & 0xffu)) ++same;
            gc += static_cast<uint32_t>(__builtin_popcount(ends & ((2u << same) - 1u)));
            continue;
        }
        gc += (r.y >> 16) & 0x3fffu;
        uint32_t b = r.z;
        if (b != 0 && vn + 2 <= n && !((r.w >> pfx_pair_bit(hay[vn] | (static_cast<uint32_t>(hay[vn + 1]) << 8))) & 1u)) b = 0;  // the filter says: nothing below
        if (b != 0) ++walkers;
        while (b != 0 && vn < n) {
            const uint32_t c = hay[vn];
            const U32x2 w = g.wrec[b ^ c];
            if ((w.y & 0xffu) != c) break;
            gc += w.y >> 8;
            b = w.x;
            ++vn;
        }
    }
    // count + checksum tables
    uint64_t xc = 0;
    uint32_t x1 = 0, x2 = 0;
    for (long long s = 0; s < n; ++s) {
        if (g.has_len1) { xc += g.cnt1[hay[s]]; x1 += g.hs1[hay[s]]; x2 += g.hs1[hay[s]] * static_cast<uint32_t>(s + 1); }
        if (s + G > n) continue;
        uint32_t k0 = 0, k1 = 0;
        for (uint32_t i = 0; i < G; ++i) {
            if (i < 4) k0 |= static_cast<uint32_t>(hay[s + i]) << (8 * i); else k1 |= static_cast<uint32_t>(hay[s + i]) << (8 * (i - 4));
        }
        const uint32_t m = k0 * kPfxMulBloom0 + k1 * kPfxMulBloom1;
        const uint32_t word = g.bloom[static_cast<uint32_t>((static_cast<uint64_t>(m) * g.bloom_words) >> 32)];
        const uint32_t m2 = m * kPfxMulBits;
        if (!((word >> ((m2 >> kPfxBit1) & 31u)) & (word >> ((m2 >> kPfxBit2) & 31u)) & 1u)) continue;
        const uint32_t mb = k0 * kPfxMulBucket0 + (k1 ^ g.seed) * kPfxMulBucket1, ms = k0 * kPfxMulSlot0 + (k1 ^ g.seed) * kPfxMulSlot1;
        const U32x4 r = g.slots_x[pfx_slot(ms, g.disp[static_cast<uint32_t>((static_cast<uint64_t>(mb) * g.buckets) >> 32)], M)];
        if (r.x != k0 || (r.y & 0xffffu) != k1 || (r.y & kPfxEmpty)) continue;
        long long vn = s + G;
        xc += (r.y >> 16) & 0x3fffu;
        x1 += r.w;
        x2 += r.w * static_cast<uint32_t>(vn);
        uint32_t b = r.z;
        while (b != 0 && vn < n) {
            const uint32_t c = hay[vn];
            const U32x4 w = g.wrec_x[b ^ c];
            if ((w.y & 0xffu) != c) break;
            ++vn;
            xc += w.y >> 8;
            x1 += w.z;
            x2 += w.z * static_cast<uint32_t>(vn);
            b = w.x;
        }
    }
    if (xc != rc || x1 != r1 || x2 != r2) {
        std::printf("MISMATCH count+checksum tables: %llu %u %u != %llu %u %u\n", (unsigned long long)xc, x1, x2, (unsigned long long)rc, r1, r2);
        return 1;
    }
    if (gc != rc) { std::printf("MISMATCH count %llu != %llu\n", (unsigned long long)gc, (unsigned long long)rc); return 1; }
    std::printf("OK G=%u len1=%d keys=%u tails=%u bloom_words=%u buckets=%u slots=%u seed=%u lds=%u count=%llu survivors/byte=%.4f false_pos/byte=%.4f walkers/byte=%.4f\n",
                G, (int)g.has_len1, g.n_keys, g.n_tails, g.bloom_words, g.buckets, g.n_slots, g.seed, g.lds_tables, (unsigned long long)gc,
                n ? double(survivors) / n : 0.0, n ? double(false_pos) / n : 0.0, n ? double(walkers) / n : 0.0);
    return 0;
}
