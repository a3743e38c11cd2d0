// Host-logic test for the renumbered `.count()` tables (gram4.hpp; no GPU needed): evaluates the count of the
// find_overlapping stream from the tables, position by position with the rules of gram4_kernels.hip — both record
// sets (plain / tail records from the hit record on), both rank directories, the arithmetic class map against the
// class table — and compares with the literal automaton walk on the original double array.  Round 6: the filter in front of rank +
// gather (gram4_filter.hpp) — every hit of the text that ends a pattern or goes on must pass (no false negatives), and the count with
// the hits that do not pass dropped must be the same; prints how many hits pass and how many of those are false positives.
//   usage: gram4_check <blob> <lds_budget> <haystack-file> [filter-bytes]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/gram4.hpp"
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
    Gram2Tables g2;
    if (!build_gram2_tables(p, static_cast<uint32_t>(std::atoi(argv[2])), g2)) { std::printf("UNAVAILABLE gram2\n"); return 0; }
    Gram4Tables g;
    build_gram4_tables(g2, g);
    if (!g.available) { std::printf("MISMATCH gram4 not built\n"); return 1; }
    const std::vector<uint8_t> hay = slurp(argv[3]);
    const long long n = static_cast<long long>(hay.size());
    const uint32_t K = g.K, C = g.C, OTH = C - 1;

    uint64_t rc = 0;
    uint32_t st = 0;
    for (long long i = 0; i < n; ++i) {
        st = p.next_state(st, hay[i]);
        for (uint32_t op = output_pos_of(p.states[st].opos_ch); op != 0; op = p.outputs[op - 1].parent) rc++;
    }

    // the class map: table and arithmetic must agree on every byte
    if (g.cls[g.unused_byte] != OTH) { std::printf("MISMATCH unused byte\n"); return 1; }
    for (uint32_t b = 0; b < 256; ++b) {
        if (g.cls[b] > OTH) { std::printf("MISMATCH class range\n"); return 1; }
        if (g.arith) {
            const uint32_t u = b - g.lo;
            if (g.cls[b] != (u < OTH ? u : OTH)) { std::printf("MISMATCH arith\n"); return 1; }
        }
    }
    auto cls = [&](long long pos) -> uint32_t { return (pos >= 0 && pos < n) ? g.cls[hay[pos]] : OTH; };
    auto ctx_ending_at = [&](long long pos) -> uint32_t {
        uint32_t w = 0;
        for (uint32_t t = 0; t < K; ++t) w = w * C + cls(pos - (K - 1) + t);
        return w;
    };
    auto text8 = [&](long long pos) -> uint64_t {  // eight bytes from pos on, bytes outside the haystack = the unused byte
        uint64_t x = 0;
        for (int b = 7; b >= 0; --b) x = (x << 8) | ((pos + b >= 0 && pos + b < n) ? hay[pos + b] : g.unused_byte);
        return x;
    };
    auto tail_count = [&](const U32x4 &rr, uint64_t text) -> uint32_t {
        const uint32_t edges = rr.x & 15u;
        const uint64_t path = (static_cast<uint64_t>(rr.w) << 32) | rr.z, diff = path ^ text;
        uint32_t same = diff ? static_cast<uint32_t>(__builtin_ctzll(diff)) >> 3 : 8u;
        same = same < edges ? same : edges;
        return static_cast<uint32_t>(__builtin_popcount((rr.x >> 4) & ((2u << same) - 1u) & 0x1ffu));
    };
    auto walk = [&](const std::vector<U32x4> &recs, uint32_t state, long long vn) -> uint64_t {  // state consumed the byte before vn
        uint64_t c = 0;
        U32x4 rr = recs[state];
        for (;;) {
            if (rr.x >> 31) { c += tail_count(rr, text8(vn)); break; }
            const uint32_t k = cls(vn);
            c += rr.z;
            if (((rr.x >> k) & 1u) == 0) break;
            rr = recs[rr.y + __builtin_popcount(rr.x & ((1u << k) - 1u))];
            ++vn;
        }
        return c;
    };
    // the filter, sized as the upload sizes it (or by the fourth argument)
    const uint32_t fbytes = argc > 4 ? static_cast<uint32_t>(std::atoi(argv[4])) : 33000u;
    const bool have_filter = build_gram4_filter(g, fbytes);
    const uint32_t W = static_cast<uint32_t>(g.bloom.size());
    auto raw = [&](long long pos) -> uint32_t { return (pos >= 0 && pos < n) ? hay[pos] : g.unused_byte; };
    uint64_t c_plain = 0, c_tail = 0, hits = 0, c_filt = 0, passed = 0, useful = 0;
    for (long long pz = 0; pz < n; ++pz) {
        const uint32_t word = g.m[ctx_ending_at(pz)];
        c_plain += word >> 30;
        c_tail += word >> 30;
        const uint32_t gp = ctx_ending_at(pz - 1), wp = g.m[gp], d = cls(pz);
        if (d == OTH && ((wp >> d) & 1u)) { std::printf("MISMATCH bit of the last class\n"); return 1; }
        if (!((wp >> d) & 1u)) continue;
        ++hits;
        const uint32_t under = __builtin_popcount(wp & ((1u << d) - 1u));
        uint32_t rank = g.sdir[gp >> 2] + under;
        for (uint32_t i = gp & ~3u; i < gp; ++i) rank += __builtin_popcount(g.m[i] & kGram4ChildBits);
        if (g.s16 && rank != static_cast<uint32_t>(g.rfull[gp]) + under) { std::printf("MISMATCH rfull\n"); return 1; }
        if (rank >= g.dhit_c.size()) { std::printf("MISMATCH rank\n"); return 1; }
        const uint32_t k1 = cls(pz + 1);
        {   // plain records: gram4_body<.., TAIL = false>
            const U32x2 r = g.dhit_c[rank];
            if (r.x >> 31) { std::printf("MISMATCH dhit_c flag\n"); return 1; }
            c_plain += (r.x >> kGram4EndsBit) & 1u;
            uint64_t below = 0;
            if ((r.x >> k1) & 1u) below = walk(g.drec_c, r.y + __builtin_popcount(r.x & ((1u << k1) - 1u)), pz + 2);
            c_plain += below;
            if (have_filter) {   // gram4_body<.., FILT = true>: the probes on the raw bytes pz-K .. pz and pz+1
                uint32_t x = 0;
                for (uint32_t i = 0; i <= K; ++i) x |= raw(pz - K + i) << (8 * i);
                const G4Probe pr = g4f_probe(x, raw(pz + 1), W);
                const uint32_t fw = g.bloom[pr.word];
                const bool pass = (fw & pr.go) == pr.go || (fw & pr.ends) == pr.ends;
                const bool is_useful = ((r.x >> kGram4EndsBit) & 1u) || ((r.x >> k1) & 1u);
                if (is_useful && !pass) { std::printf("MISMATCH filter: a hit that ends a pattern or goes on does not pass (position %lld)\n", pz); return 1; }
                if (pass) { ++passed; c_filt += ((r.x >> kGram4EndsBit) & 1u) + below; }
                if (is_useful) ++useful;
            }
        }
        {   // tail records from the hit record on
            const U32x4 r = g.dhit_t[rank];
            if (r.x >> 31) {
                c_tail += tail_count(r, text8(pz + 1));
            } else {
                c_tail += (r.x >> kGram4EndsBit) & 1u;
                if ((r.x >> k1) & 1u) c_tail += walk(g.drec_t, r.y + __builtin_popcount(r.x & ((1u << k1) - 1u)), pz + 2);
            }
        }
    }
    if (c_plain != rc || c_tail != rc) {
        std::printf("MISMATCH count plain %llu tail %llu want %llu\n", (unsigned long long)c_plain, (unsigned long long)c_tail, (unsigned long long)rc);
        return 1;
    }
    if (have_filter) {
        uint64_t shorts = 0;
        for (long long pz = 0; pz < n; ++pz) shorts += g.m[ctx_ending_at(pz)] >> 30;
        if (shorts + c_filt != rc) { std::printf("MISMATCH count through the filter %llu want %llu\n", (unsigned long long)(shorts + c_filt), (unsigned long long)rc); return 1; }
    }
    std::printf("OK K=%u C=%u arith=%d lo=%u count=%llu hits=%llu filter=%d words=%u keys=%u passed=%llu useful=%llu\n", K, C, g.arith ? 1 : 0, g.lo, (unsigned long long)rc,
                (unsigned long long)hits, have_filter ? 1 : 0, W, g.filter_keys, (unsigned long long)passed, (unsigned long long)useful);
    return 0;
}
