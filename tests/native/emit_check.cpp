// Host-logic test for the GRAM tuple-emission tables (no GPU needed): builds the (start, end, value) stream from the
// tables with the rules of gram2_emit_kernels.hip — short patterns from the flag bits and the v1/v2/v3 value tables,
// deep ones from hits and walks over ehit / erec, deep before short and longest first at every end — and compares it,
// tuple by tuple, with the literal automaton's list walk (bytewise/iter.rs:133-176).
//   usage: emit_check <blob> <lds_budget> <haystack-file>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <tuple>
#include <vector>

#include "../../daachorse_amd/csrc/gram2.hpp"
#include "../../daachorse_amd/csrc/pma.hpp"

using namespace daac;
typedef std::tuple<uint64_t, uint64_t, uint32_t> Tup;  // start, end, value

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
    if (!build_gram2_tables(p, static_cast<uint32_t>(std::atoi(argv[2])), g) || !g.emit_available) { std::printf("UNAVAILABLE emit\n"); return 0; }
    const std::vector<uint8_t> hay = slurp(argv[3]);
    const long long n = static_cast<long long>(hay.size());
    const uint32_t K = g.K, C = g.C;

    std::vector<Tup> want;
    uint32_t st = 0;
    for (long long i = 0; i < n; ++i) {
        st = p.next_state(st, hay[i]);
        for (uint32_t op = output_pos_of(p.states[st].opos_ch); op != 0; op = p.outputs[op - 1].parent)
            want.emplace_back(static_cast<uint64_t>(i + 1 - p.outputs[op - 1].length), static_cast<uint64_t>(i + 1), p.outputs[op - 1].value);
    }

    auto cls = [&](long long pos) -> uint32_t { return (pos >= 0 && pos < n) ? g.cls[hay[pos]] : 0u; };
    auto kgram_ending_at = [&](long long pos) -> uint32_t {
        uint32_t w = 0;
        for (uint32_t t = 0; t < K; ++t) w = w * C + cls(pos - (K - 1) + t);
        return w;
    };
    // deep matches by the byte they end at: (length, copy, value) — copy > 0: a further copy of a duplicate pattern
    struct Deep { uint32_t len, copy, value; };
    std::vector<std::vector<Deep>> deep(static_cast<size_t>(n));
    auto add_state = [&](long long at, uint32_t len, uint32_t value, uint32_t copies, uint32_t state) {
        deep[at].push_back(Deep{len, 0u, value});
        for (uint32_t k = 0; k < copies; ++k) deep[at].push_back(Deep{len, k + 1, g.dupv[g.dupo[state] + k]});
    };
    for (long long pz = 0; pz < n; ++pz) {
        const uint32_t gp = kgram_ending_at(pz - 1), wp = g.me[gp], d = cls(pz);
        if (d == 0 || ((wp >> d) & 1u) == 0) continue;
        uint32_t rank = g.sdir[gp >> 2] + __builtin_popcount(wp & 0x1ffffffeu & ((1u << d) - 1u));
        for (uint32_t i = gp & ~3u; i < gp; ++i) rank += __builtin_popcount(g.me[i] & 0x1ffffffeu);
        const U32x2 h = g.ehit[rank];
        if (h.x & 1u) add_state(pz, K + 1, h.y, g.ecopies[rank], g.level_start + rank);
        uint32_t k1 = cls(pz + 1);
        if (k1 == 0 || ((h.x >> k1) & 1u) == 0) continue;
        uint32_t id = g.cfirst[rank] + __builtin_popcount(h.x & ((1u << k1) - 1u) & ~1u);
        long long nx = pz + 2;
        uint32_t kn = cls(nx);
        for (;;) {
            const U32x4 r = g.erec[id];
            if (r.x & 1u) { if ((r.w & 0xffffffu) != static_cast<uint32_t>(nx - 1 - (pz - K) + 1)) { std::printf("MISMATCH depth\n"); return 1; } add_state(nx - 1, r.w & 0xffffffu, r.z, r.w >> 24, id); }
            if (kn == 0 || ((r.x >> kn) & 1u) == 0) break;
            id = r.y + __builtin_popcount(r.x & ((1u << kn) - 1u) & ~1u);
            ++nx;
            kn = cls(nx);
        }
    }
    std::vector<Tup> got;
    for (long long pz = 0; pz < n; ++pz) {
        std::sort(deep[pz].begin(), deep[pz].end(), [](const Deep &a, const Deep &b) { return a.len != b.len ? a.len > b.len : a.copy < b.copy; });
        for (size_t i = 0; i + 1 < deep[pz].size(); ++i)
            if (deep[pz][i].len == deep[pz][i + 1].len && deep[pz][i].copy == deep[pz][i + 1].copy) { std::printf("MISMATCH two deep matches of one length\n"); return 1; }
        const uint64_t end = static_cast<uint64_t>(pz + 1);
        for (const auto &dv : deep[pz]) got.emplace_back(end - dv.len, end, dv.value);
        const uint32_t gw = kgram_ending_at(pz), f = g.me[gw] >> 29;
        if (K == 3 && (f & 4u)) got.emplace_back(end - 3, end, g.v3[gw]);
        if (f & 2u) got.emplace_back(end - 2, end, g.v2[cls(pz - 1) * C + cls(pz)]);
        if (f & 1u) got.emplace_back(end - 1, end, g.v1[cls(pz)]);
    }
    if (got != want) {
        size_t i = 0;
        while (i < got.size() && i < want.size() && got[i] == want[i]) ++i;
        std::printf("MISMATCH at tuple %zu of %zu/%zu\n", i, got.size(), want.size());
        return 1;
    }
    std::printf("OK %lld K=%u C=%u tuples=%zu maxlen=%u\n", n, K, C, got.size(), g.max_len);
    return 0;
}
