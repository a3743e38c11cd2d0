// Host-logic test for the wide-alphabet GRAM tables, gram2w.hpp (no GPU needed): evaluates count + checksum of the
// find_overlapping stream from the tables, position by position with the rules of gram2w_kernels.hip, and
// compares with the literal automaton walk on the original double array.
//   usage: gram2_check <blob> <lds_budget> <haystack-file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/gram2w.hpp"
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
    Gram2WTables g;
    if (!build_gram2w_tables(p, static_cast<uint32_t>(std::atoi(argv[2])), g)) { std::printf("UNAVAILABLE gram2w\n"); return 0; }
    const std::vector<uint8_t> hay = slurp(argv[3]);
    const long long n = static_cast<long long>(hay.size());
    const uint32_t K = 2, C = g.C;

    // reference: literal automaton walk, outputs by list walk
    uint64_t rc = 0;
    uint32_t r1 = 0, r2 = 0, st = 0;
    for (long long i = 0; i < n; ++i) {
        st = p.next_state(st, hay[i]);
        for (uint32_t op = output_pos_of(p.states[st].opos_ch); op != 0; op = p.outputs[op - 1].parent) {
            const uint32_t h = match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            rc++; r1 += h; r2 += h * static_cast<uint32_t>(i + 1);
        }
    }

    auto cls = [&](long long pos) -> uint32_t { return (pos >= 0 && pos < n) ? g.cls[hay[pos]] : 0u; };
    auto kgram_ending_at = [&](long long pos) -> uint32_t {
        uint32_t w = 0;
        for (uint32_t t = 0; t < K; ++t) w = w * C + cls(pos - (K - 1) + t);
        return w;
    };
    uint64_t gc = 0;
    uint32_t g1 = 0, g2 = 0;
    for (long long pz = 0; pz < n; ++pz) {
        const uint32_t end = static_cast<uint32_t>(pz + 1);
        // (1) the short patterns ending at pz: count in the M word, sum of h32 through CID -> H
        const uint32_t gw = kgram_ending_at(pz);
        const uint64_t word = g.m[gw];
        gc += static_cast<uint32_t>(word >> 62);
        if ((g.cid4[gw] & 3u) != 0 || g.cid4[gw] / 4 >= g.hsum.size()) { std::printf("MISMATCH cid4\n"); return 1; }
        const uint32_t hs = g.hsum[g.cid4[gw] / 4];
        if (g.exact_available) { g1 += hs; g2 += hs * end; }
        // (2) a (K+1)-gram ending at pz that is a trie prefix: bit cls(pz) of the word of the K-gram ending at pz - 1
        const uint32_t gp = kgram_ending_at(pz - 1), d = cls(pz);
        const uint64_t wp = g.m[gp];
        if (d != 0 && ((wp >> d) & 1ull)) {
            uint32_t rank = g.sdir[gp >> 2] + __builtin_popcountll(wp & kGram2WMaskBits & ((1ull << d) - 1ull));
            for (uint32_t i = gp & ~3u; i < gp; ++i) rank += __builtin_popcountll(g.m[i] & kGram2WMaskBits);
            uint32_t id = g.level_start + rank;
            const U32x4 hrec = g.dhit[rank];  // {cmap lo, cmap hi, own_hsum, first_child}
            if (hrec.x != g.drec[id].x || hrec.y != g.drec[id].y || hrec.z != g.drec[id].w || hrec.w != g.drec[id].z) { std::printf("MISMATCH dhit\n"); return 1; }
            gc += hrec.z != 0; g1 += hrec.z; g2 += hrec.z * end;
            const uint32_t k1 = cls(pz + 1), k2 = cls(pz + 2);
            const uint64_t hc = (static_cast<uint64_t>(hrec.y) << 32) | hrec.x;
            if ((hc >> k1) & 1ull) {
                id = hrec.w + __builtin_popcountll(hc & ((1ull << k1) - 1ull));
                long long nx = pz + 2;  // the state consumed the byte before nx
                uint32_t kn = k2;
                for (;;) {
                    const U32x4 r = g.drec[id];
                    const uint64_t cm = (static_cast<uint64_t>(r.y) << 32) | r.x;
                    gc += r.w != 0; g1 += r.w; g2 += r.w * static_cast<uint32_t>(nx);
                    if (((cm >> kn) & 1ull) == 0) break;
                    id = r.z + __builtin_popcountll(cm & ((1ull << kn) - 1ull));
                    ++nx;
                    kn = cls(nx);
                }
            }
        }
    }
    if (!g.exact_available) {  // count only: the short patterns' share of the checksum is not in the tables that would be staged
        if (gc != rc) { std::printf("MISMATCH count %llu vs %llu\n", (unsigned long long)gc, (unsigned long long)rc); return 1; }
        std::printf("OK-COUNT %lld K=%u C=%u count=%llu lds=%u\n", n, K, C, (unsigned long long)gc, g.lds_count);
        return 0;
    }
    if (gc != rc || g1 != r1 || g2 != r2) {
        std::printf("MISMATCH count %llu vs %llu, s1 %08x vs %08x, s2 %08x vs %08x\n", (unsigned long long)gc, (unsigned long long)rc, g1, r1, g2, r2);
        return 1;
    }
    std::printf("OK %lld K=%u C=%u count=%llu lds=%u/%u ids=%zu deep=%zu\n", n, K, C, (unsigned long long)gc, g.lds_count, g.lds_exact,
                g.hsum.size(), g.dhit.size());
    return 0;
}
