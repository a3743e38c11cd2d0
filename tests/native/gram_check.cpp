// Host-logic test for the GRAM tables (no GPU needed): evaluates count + checksum of the
// find_overlapping stream from the GRAM tables, position by position exactly as the HIP kernel
// does, and compares with the literal automaton walk on the original double array.
//   usage: gram_check <blob> <lds_budget> <haystack-file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/gram.hpp"
#include "../../daachorse_amd/csrc/pma.hpp"
#include "../../daachorse_amd/csrc/repack.hpp"

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
    RepackOptions ro;
    TierTables t;
    if (!build_tier_tables(p, ro, t)) { std::printf("UNAVAILABLE tier\n"); return 0; }
    GramTables g;
    if (!build_gram_tables(p, t, static_cast<uint32_t>(std::atoi(argv[2])), g)) { std::printf("UNAVAILABLE gram\n"); return 0; }
    const std::vector<uint8_t> hay = slurp(argv[3]);
    const size_t n = hay.size();
    const uint32_t K = g.K, C = g.C;

    // reference: literal automaton walk, outputs by list walk
    uint64_t rc = 0;
    uint32_t r1 = 0, r2 = 0;
    uint32_t st = 0;
    for (size_t i = 0; i < n; ++i) {
        st = p.next_state(st, hay[i]);
        for (uint32_t op = output_pos_of(p.states[st].opos_ch); op != 0; op = p.outputs[op - 1].parent) {
            const uint32_t h = match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            rc++; r1 += h; r2 += h * static_cast<uint32_t>(i + 1);
        }
    }

    // GRAM evaluation
    auto cls = [&](long long pos) -> uint32_t { return (pos >= 0 && static_cast<size_t>(pos) < n) ? g.cls[hay[pos]] : 0u; };
    uint32_t pk1 = 1;
    for (uint32_t i = 0; i + 1 < K; ++i) pk1 *= C;
    uint64_t gc = 0, tails_met = 0;
    uint32_t g1 = 0, g2 = 0;
    for (size_t pz = 0; pz < n; ++pz) {
        const long long p0 = static_cast<long long>(pz);
        const uint32_t end = static_cast<uint32_t>(pz + 1);
        uint32_t iW = 0;
        for (uint32_t tt = 0; tt < K; ++tt) iW = iW * C + cls(p0 - (K - 1) + tt);
        const uint32_t iB = cls(p0 - K) * pk1 * C + iW;
        const U32x2 s = g.combo[g.cid[iW]];
        gc += s.x; g1 += s.y; g2 += s.y * end;
        const uint32_t bw = g.bbits[iB >> 5];
        if ((bw >> (iB & 31)) & 1u) {
            const uint32_t wi = iB >> 5;
            const uint32_t rank = g.bsuper[wi >> 3] + g.brank[wi >> 1] + ((wi & 1) ? __builtin_popcount(g.bbits[wi - 1]) : 0) +
                                  __builtin_popcount(bw & ((1u << (iB & 31)) - 1u));
            uint32_t id = g.level_start + rank;
            const U32x2 hrec = g.dhit[rank];  // what the fast pass reads: must agree with the full record
            if (hrec.x != g.drec[id].x || hrec.y != g.drec[id].w || g.drec[id].z != (hrec.y != 0 ? 1u : 0u)) { std::printf("MISMATCH dhit\n"); return 1; }
            // fast pass: the depth-(K+1) state's own pattern, then "does the branch go on with the next byte"
            gc += hrec.y != 0; g1 += hrec.y; g2 += hrec.y * end;
            const uint32_t k1 = cls(p0 + 1), k2 = cls(p0 + 2);
            if ((hrec.x >> k1) & 1u) {
                // walker, as queued by the kernel: the child state via cfirst, the class after next in the entry
                if (g.cfirst[rank] != g.drec[id].y) { std::printf("MISMATCH cfirst\n"); return 1; }
                id = g.cfirst[rank] + __builtin_popcount(hrec.x & ((1u << k1) - 1u));
                long long nx = p0 + 2;  // the state consumed the byte before nx
                uint32_t kn = k2;
                bool first_rec = true;
                for (;;) {
                    const U32x4 r = g.drec[id];
                    if (r.y >> 31) {   // a tail record (round 6): one path, one pattern end, compared with the next eight text bytes
                        if (first_rec) { std::printf("MISMATCH a tail record where the walkers start\n"); return 1; }
                        uint64_t text = 0;
                        for (int b = 7; b >= 0; --b) text = (text << 8) | ((nx + b >= 0 && nx + b < static_cast<long long>(n)) ? hay[nx + b] : g.unused_byte);
                        const uint64_t diff = ((static_cast<uint64_t>(r.w) << 32) | r.z) ^ text;
                        const uint32_t edges = r.y & 15u, at = (r.y >> 4) & 15u;
                        uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
                        same = same < edges ? same : edges;
                        if (at <= same) { gc += 1; g1 += r.x; g2 += r.x * static_cast<uint32_t>(nx + at); }
                        ++tails_met;
                        break;
                    }
                    first_rec = false;
                    gc += r.z; g1 += r.w; g2 += r.w * static_cast<uint32_t>(nx);
                    if (((r.x >> kn) & 1u) == 0) break;
                    id = r.y + __builtin_popcount(r.x & ((1u << kn) - 1u));
                    ++nx;
                    kn = cls(nx);
                }
            }
        }
    }
    if (gc != rc || g1 != r1 || g2 != r2) {
        std::printf("MISMATCH count %llu vs %llu, s1 %08x vs %08x, s2 %08x vs %08x\n", (unsigned long long)gc, (unsigned long long)rc, g1, r1, g2, r2);
        return 1;
    }
    std::printf("OK %zu K=%u C=%u count=%llu lds=%u short=%d combos=%zu tail_records=%u tails_met=%llu\n", n, K, C, (unsigned long long)gc, g.lds_bytes, int(g.has_short), g.combo.size(),
                g.n_tail, (unsigned long long)tails_met);
    return 0;
}
