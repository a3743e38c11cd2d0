// Host-logic test for the JUMP tables, jump.hpp (no GPU needed): L(s) from the tables with the rules of jump_kernels.hip (MS word,
// rank, JHIT / JREC walk), N / D by the saturating suffix minimum, the chain over them with H1 / H2 / H3 / the walk's own h32 —
// against the literal FindIterator (reference src/bytewise/iter.rs:58-113) on the original double array.
//   usage: jump_check <blob> <haystack-file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/jump.hpp"
#include "../../daachorse_amd/csrc/pma.hpp"

using namespace daac;

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::vector<uint8_t> blob = slurp(argv[1]);
    HostPma p;
    if (HostPma::deserialize(blob.data(), blob.size(), p, nullptr) != DAAC_OK) { std::printf("BADBLOB\n"); return 1; }
    JumpTables g;
    if (!build_jump_tables(p, g)) { std::printf("UNAVAILABLE jump\n"); return 0; }
    const std::vector<uint8_t> hay = slurp(argv[2]);
    const long long n = static_cast<long long>(hay.size());

    // the reference: restart at ROOT after every match, report the list head
    uint64_t rc = 0;
    uint32_t r1 = 0, r2 = 0, st = 0;
    for (long long i = 0; i < n; ++i) {
        st = p.next_state(st, hay[i]);
        const uint32_t op = output_pos_of(p.states[st].opos_ch);
        if (op != 0) {
            const uint32_t h = match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            rc++; r1 += h; r2 += h * static_cast<uint32_t>(i + 1);
            st = 0;
        }
    }

    const uint32_t C = g.C;
    auto cls = [&](long long i) -> uint32_t { return (i >= 0 && i < n) ? g.cls[hay[i]] : 0u; };
    std::vector<uint8_t> L(n + 2, 0);
    std::vector<uint32_t> hdeep(n + 2, 0);
    uint64_t deep = 0;
    for (long long s = 0; s < n; ++s) {
        const uint32_t idx = (cls(s) * C + cls(s + 1)) * C + cls(s + 2);
        const uint32_t m = g.ms[idx];
        L[s] = static_cast<uint8_t>(m >> 30);
        const uint32_t d = cls(s + 3);
        if (!((m >> d) & 1u)) continue;
        if (d == 0 || (m >> 30) != 0) { std::printf("MISMATCH table: continuation bit of class 0 / beside a short pattern\n"); return 1; }
        uint32_t rank = g.sdir[idx >> 2] + static_cast<uint32_t>(__builtin_popcount(m & ((1u << d) - 2u)));
        for (uint32_t q = idx & ~3u; q < idx; ++q) rank += static_cast<uint32_t>(__builtin_popcount(g.ms[q] & 0x3ffffffeu));
        U32x4 rec = g.jhit[rank];
        long long pos = s + 4;
        ++deep;
        for (;;) {
            if (rec.x & 1u) { L[s] = static_cast<uint8_t>(rec.w); hdeep[s] = rec.z; break; }
            const uint32_t k = cls(pos);
            if (pos >= n || k == 0 || !((rec.x >> k) & 1u)) break;
            rec = g.jrec[rec.y + static_cast<uint32_t>(__builtin_popcount(rec.x & ((1u << k) - 2u)))];
            ++pos;
        }
    }
    // N, D: right to left, saturating
    std::vector<uint8_t> N(n + 1, 255), D(n + 1, 0);
    uint32_t nn = 255, dd = 0;
    for (long long e = n - 1; e >= 0; --e) {
        const uint32_t other = nn < 255 ? nn + 1 : 255;
        const bool mine = L[e] != 0 && L[e] <= other;
        dd = mine ? 0 : dd + 1;
        nn = mine ? L[e] : other;
        N[e] = static_cast<uint8_t>(nn);
        D[e] = static_cast<uint8_t>(dd);
    }
    uint64_t gc = 0;
    uint32_t s1 = 0, s2 = 0;
    long long e = 0;
    while (e < n) {
        if (N[e] == 255) { e += 255 - g.max_len; continue; }
        const long long smin = e + D[e], end = e + N[e];
        const uint32_t len = N[e] - D[e];
        uint32_t h;
        if (len == 1) h = g.h1[cls(smin)];
        else if (len == 2) h = g.h2[cls(smin) * C + cls(smin + 1)];
        else if (len == 3) h = g.h3[(cls(smin) * C + cls(smin + 1)) * C + cls(smin + 2)];
        else h = hdeep[smin];
        gc++; s1 += h; s2 += h * static_cast<uint32_t>(end);
        e = end;
    }
    if (gc != rc || s1 != r1 || s2 != r2) {
        std::printf("MISMATCH find_iter: %llu %u %u != %llu %u %u\n", (unsigned long long)gc, s1, s2, (unsigned long long)rc, r1, r2);
        return 1;
    }
    std::printf("OK C=%u max_len=%u states=%u deep_starts/byte=%.4f matches/byte=%.4f\n", C, g.max_len, g.N, n ? double(deep) / n : 0.0, n ? double(gc) / n : 0.0);
    return 0;
}
