// Host-logic test for the GPU re-pack (no GPU needed): walks the TIERED tables with the same
// step rule the HIP TierEngine uses and checks, byte by byte, that it lands on the same state as
// the reference transition function on the original double array.
//   usage: repack_check <blob> <lds_budget> <dense_depth|-1> <haystack-file>
// prints "OK <steps> N=<N> NA=<NA> NB=<NB> C=<C> row32=<0|1>" or "UNAVAILABLE" / "MISMATCH ...".
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/pma.hpp"
#include "../../daachorse_amd/csrc/repack.hpp"

using namespace daac;

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static bool tier_step(const TierTables &t, uint32_t &st, uint8_t c) {
    const uint32_t k = t.cls[c];
    uint32_t s = st;
    for (;;) {
        if (s < t.NA) {
            const uint32_t e = t.row32 ? t.rows32[size_t(s) * t.C + k] : t.rows16[size_t(s) * t.C + k];
            const uint32_t sh = t.row32 ? 31 : 15;
            st = e & ((1u << sh) - 1u);
            return (e >> sh) != 0;
        }
        const U32x4 r = t.grec[s];
        const uint32_t cmap = s < t.NB ? t.bcmap[s - t.NA] : r.x;
        const uint32_t fail = s < t.NB ? t.bfail[s - t.NA] : r.w;
        if (((cmap >> k) & 1u) == 0) { s = fail; continue; }
        st = r.z + __builtin_popcount(cmap & ((1u << k) - 1u));
        return ((r.y >> k) & 1u) != 0;
    }
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::vector<uint8_t> blob = slurp(argv[1]);
    HostPma p;
    size_t consumed = 0;
    if (HostPma::deserialize(blob.data(), blob.size(), p, &consumed) != DAAC_OK) { std::printf("BADBLOB\n"); return 1; }
    RepackOptions ro;
    ro.lds_budget = static_cast<uint32_t>(std::atoi(argv[2]));
    ro.dense_depth = std::atoi(argv[3]);
    TierTables t;
    if (!build_tier_tables(p, ro, t)) { std::printf("UNAVAILABLE\n"); return 0; }
    if (t.lds_bytes() > ro.lds_budget + 64) { std::printf("MISMATCH lds_bytes %u > budget\n", t.lds_bytes()); return 1; }
    const std::vector<uint8_t> hay = slurp(argv[4]);
    uint32_t ref = 0, st = 0;
    for (size_t i = 0; i < hay.size(); ++i) {
        ref = p.next_state(ref, hay[i]);
        const bool flag = tier_step(t, st, hay[i]);
        const uint32_t op = output_pos_of(p.states[ref].opos_ch);
        if (t.old_of_new[st] != ref || flag != (op != 0) || t.sopos[st] != op) {
            std::printf("MISMATCH at %zu: tier state %u (old %u) vs ref %u, flag %d, opos %u/%u\n", i, st, t.old_of_new[st], ref, int(flag),
                        t.sopos[st], op);
            return 1;
        }
    }
    std::printf("OK %zu N=%u NA=%u NB=%u C=%u row32=%d lds=%u\n", hay.size(), t.N, t.NA, t.NB, t.C, int(t.row32), t.lds_bytes());
    return 0;
}
