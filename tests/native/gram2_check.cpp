// Host-logic test for the second GRAM table set (no GPU needed): evaluates count + checksum of the
// find_overlapping stream from the tables, position by position with the rules of gram2_kernels.hip, and
// compares with the literal automaton walk on the original double array.
//   usage: gram2_check <blob> <lds_budget> <haystack-file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/gram2.hpp"
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
    Gram2Tables g;
    if (!build_gram2_tables(p, static_cast<uint32_t>(std::atoi(argv[2])), g)) { std::printf("UNAVAILABLE gram2\n"); return 0; }
    const std::vector<uint8_t> hay = slurp(argv[3]);
    const long long n = static_cast<long long>(hay.size());
    const uint32_t K = g.K, C = g.C;

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
    auto byte_at = [&](long long pos) -> int { return (pos >= 0 && pos < n) ? hay[pos] : -1; };  // -1: nothing there
    uint64_t gc = 0, gc_walk = 0, gc_tail = 0;
    uint32_t g1 = 0, g2 = 0;
    for (long long pz = 0; pz < n; ++pz) {
        const uint32_t end = static_cast<uint32_t>(pz + 1);
        // (1) the short patterns ending at pz: count in the M word, sum of h32 through CID -> H
        const uint32_t gw = kgram_ending_at(pz), word = g.m[gw];
        gc += word >> 30;
        if ((g.cid4[gw] & 3u) != 0 || g.cid4[gw] / 4 >= g.hsum.size()) { std::printf("MISMATCH cid4\n"); return 1; }
        const uint32_t hs = g.hsum[g.cid4[gw] / 4];
        if (g.exact_available) { g1 += hs; g2 += hs * end; }
        // (2) a (K+1)-gram ending at pz that is a trie prefix: bit cls(pz) of the word of the K-gram ending at pz - 1
        const uint32_t gp = kgram_ending_at(pz - 1), wp = g.m[gp], d = cls(pz);
        if (d != 0 && ((wp >> d) & 1u)) {
            uint32_t rank = g.sdir[gp >> 2] + __builtin_popcount(wp & kGram2MaskBits & ((1u << d) - 1u));
            for (uint32_t i = gp & ~3u; i < gp; ++i) rank += __builtin_popcount(g.m[i] & kGram2MaskBits);
            if (g.s16 && g.sdir[gp >> 2] >= 65536) { std::printf("MISMATCH s16\n"); return 1; }
            uint32_t id = g.level_start + rank;
            const U32x2 hrec = g.dhit[rank];
            if (hrec.x != g.drec[id].x || hrec.y != g.drec[id].w || g.drec[id].z != (hrec.y != 0 ? 1u : 0u)) { std::printf("MISMATCH dhit\n"); return 1; }
            if (g.dhit_c[rank].x != (hrec.x | (hrec.y != 0 ? 1u : 0u)) || g.dhit_c[rank].y != g.cfirst[rank]) { std::printf("MISMATCH dhit_c\n"); return 1; }
            gc += hrec.y != 0; g1 += hrec.y; g2 += hrec.y * end;
            const uint32_t k1 = cls(pz + 1), k2 = cls(pz + 2);
            if ((hrec.x >> k1) & 1u) {
                if (g.cfirst[rank] != g.drec[id].y) { std::printf("MISMATCH cfirst\n"); return 1; }
                id = g.cfirst[rank] + __builtin_popcount(hrec.x & ((1u << k1) - 1u));
                long long nx = pz + 2;  // the state consumed the byte before nx
                uint32_t kn = k2;
                {   // the `.count()` walk over drec_c: tail records compare the path with the text in one step
                    uint32_t id_c = id;
                    long long nc = nx;
                    uint32_t kc = kn;
                    for (;;) {
                        const U32x4 r = g.drec_c[id_c];
                        if (r.x >> 31) {
                            const uint32_t edges = r.x & 15u;
                            uint32_t same = 0;
                            while (same < edges && byte_at(nc + same) == static_cast<int>(((same < 4 ? r.z >> (8 * same) : r.w >> (8 * (same - 4))) & 0xffu))) ++same;
                            gc_tail += __builtin_popcount((r.x >> 4) & ((2u << same) - 1u) & 0x1ffu);
                            break;
                        }
                        gc_tail += r.z;
                        if (((r.x >> kc) & 1u) == 0) break;
                        id_c = r.y + __builtin_popcount(r.x & ((1u << kc) - 1u));
                        ++nc;
                        kc = cls(nc);
                    }
                }
                for (bool first_rec = true;; first_rec = false) {
                    const U32x4 r = g.drec[id];
                    if (r.y >> 31) {   // a tail record with the pattern's h inline (round 6): one path, one pattern end
                        if (first_rec) { std::printf("MISMATCH a tail record where the walkers start\n"); return 1; }
                        uint64_t text = 0;
                        for (int b = 7; b >= 0; --b) text = (text << 8) | ((nx + b >= 0 && nx + b < n) ? hay[nx + b] : g.unused_byte);
                        const uint64_t diff = ((static_cast<uint64_t>(r.w) << 32) | r.z) ^ text;
                        const uint32_t edges = r.y & 15u, at = (r.y >> 4) & 15u;
                        uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
                        same = same < edges ? same : edges;
                        if (at <= same) { gc += 1; g1 += r.x; g2 += r.x * static_cast<uint32_t>(nx + at); gc_walk += 1; }
                        break;
                    }
                    gc += r.z; g1 += r.w; g2 += r.w * static_cast<uint32_t>(nx);
                    gc_walk += r.z;
                    if (((r.x >> kn) & 1u) == 0) break;
                    id = r.y + __builtin_popcount(r.x & ((1u << kn) - 1u));
                    ++nx;
                    kn = cls(nx);
                }
            }
        }
    }
    if (gc_tail != gc_walk) { std::printf("MISMATCH tail records %llu vs %llu\n", (unsigned long long)gc_tail, (unsigned long long)gc_walk); return 1; }
    if (!g.exact_available) {  // count only: the short patterns' share of the checksum is not in the tables that would be staged
        if (gc != rc) { std::printf("MISMATCH count %llu vs %llu\n", (unsigned long long)gc, (unsigned long long)rc); return 1; }
        std::printf("OK-COUNT %lld K=%u C=%u count=%llu lds=%u\n", n, K, C, (unsigned long long)gc, g.lds_count);
        return 0;
    }
    if (gc != rc || g1 != r1 || g2 != r2) {
        std::printf("MISMATCH count %llu vs %llu, s1 %08x vs %08x, s2 %08x vs %08x\n", (unsigned long long)gc, (unsigned long long)rc, g1, r1, g2, r2);
        return 1;
    }
    std::printf("OK %lld K=%u C=%u count=%llu lds=%u/%u ids=%zu deep=%zu s16=%d\n", n, K, C, (unsigned long long)gc, g.lds_count, g.lds_exact,
                g.hsum.size(), g.dhit.size(), int(g.s16));
    return 0;
}
