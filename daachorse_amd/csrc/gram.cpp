// GRAM engine tables (host side) — see gram.hpp.
#include "gram.hpp"

#include <algorithm>

namespace daac {

namespace {
uint64_t ipow(uint64_t b, uint32_t e) {
    uint64_t r = 1;
    while (e--) r *= b;
    return r;
}
}  // namespace

bool build_gram_tables(const HostPma &p, const TierTables &tier, uint32_t lds_budget, GramTables &out) {
    out = GramTables{};
    // "" as a pattern makes every position (and end = 0) a match: leave that to the AC engines
    if (!tier.available || tier.root_flag != 0 || !p.is_standard()) return false;
    const uint32_t C = tier.C, N = tier.N;

    // byte class representatives
    uint8_t rep[32] = {0};
    bool have[32] = {false};
    for (uint32_t c = 0; c < 256; ++c) {
        const uint32_t k = tier.cls[c];
        if (!have[k]) { have[k] = true; rep[k] = static_cast<uint8_t>(c); }
    }
    if (!have[0]) return false;

    // depth and own patterns of every state (ids are breadth-first: parents come first)
    std::vector<uint32_t> depth(N, 0), own_cnt(N, 0), own_hs(N, 0);
    std::vector<uint64_t> gram(N, 0);  // class string as a base-C number (meaningful while short)
    uint32_t max_depth = 0;
    for (uint32_t s = 0; s < N; ++s) {
        const U32x4 r = tier.grec[s];
        uint32_t j = 0;
        for (uint32_t k = 0; k < C; ++k) {
            if (((r.x >> k) & 1u) == 0) continue;
            const uint32_t ch = r.z + j++;
            depth[ch] = depth[s] + 1;
            gram[ch] = depth[ch] <= 6 ? gram[s] * C + k : 0;
            max_depth = std::max(max_depth, depth[ch]);
        }
        uint32_t op = tier.sopos[s];
        while (op != 0 && p.outputs[op - 1].length == depth[s]) {  // own = list entries as long as the state is deep
            own_cnt[s]++;
            own_hs[s] += match_hash32(p.outputs[op - 1].value, p.outputs[op - 1].length);
            op = p.outputs[op - 1].parent;
        }
    }

    // ---- choose K -----------------------------------------------------------------------------
    auto pad16 = [](uint64_t x) { return static_cast<uint32_t>((x + 15) & ~15ull); };
    uint32_t K = 0, lds = 0;
    for (uint32_t cand : {3u, 2u}) {
        const uint64_t wwords = (ipow(C, cand) + 31) / 32, bwords = (ipow(C, cand + 1) + 31) / 32;
        if (bwords * 32 >= (1ull << 31)) continue;
        uint64_t nword = 0;
        for (uint32_t s = 0; s < N; ++s) nword += depth[s] == cand && own_cnt[s] != 0;
        if (nword >= 65536) continue;  // wrank is u16
        const uint64_t bytes = 256 + pad16(ipow(C, cand - 1) * 8) + pad16(wwords * 4) + pad16(wwords * 2) + pad16(nword * 8) +
                               pad16(bwords * 4) + pad16(bwords * 2) + pad16(((bwords + 63) / 64) * 4);
        if (bytes > lds_budget) continue;
        // the 8-byte hit record of a depth-(K+1) state carries no count: it must be 0/1 and visible in the sum
        bool regular = true;
        for (uint32_t s = 0; s < N && regular; ++s)
            if (depth[s] == cand + 1 && own_cnt[s] != (own_hs[s] != 0 ? 1u : 0u)) regular = false;
        if (!regular) continue;
        K = cand;
        lds = static_cast<uint32_t>(bytes);
        break;
    }
    if (K == 0) return false;

    out.K = K;
    out.C = C;
    out.N = N;
    out.unused_byte = rep[0];
    out.cls = tier.cls;
    out.lds_bytes = lds;

    std::vector<uint32_t> new_of_old(p.states.size(), 0xffffffffu);
    for (uint32_t s = 0; s < N; ++s) new_of_old[tier.old_of_new[s]] = s;

    // ---- T_{K-1}: everything of length <= K-1 that ends after these K-1 classes ------------------
    const uint32_t nshort = static_cast<uint32_t>(ipow(C, K - 1));
    out.tshort.assign(nshort, U32x2{0, 0});
    for (uint32_t g = 0; g < nshort; ++g) {
        uint32_t st = kRoot;
        for (uint32_t i = 0; i < K - 1; ++i) {
            const uint32_t k = (g / static_cast<uint32_t>(ipow(C, K - 2 - i))) % C;
            st = p.next_state(st, rep[k]);  // the reference's own delta; a class-0 byte resets to ROOT
        }
        const OutSum s = tier.ssum[new_of_old[st]];
        out.tshort[g] = U32x2{s.cnt, s.hsum};
        if (s.cnt != 0) out.has_short = true;
    }

    // ---- W_K and B_{K+1} with their rank directories; per-state walk records ------------------------
    // Breadth-first ids order each level lexicographically by class string (children are numbered
    // parent by parent, class-ascending), i.e. by gram index: the rank of a set bit in B_{K+1} is
    // the state's offset within its level, and W_K's compact records follow the same order.
    out.wbits.assign(static_cast<size_t>((ipow(C, K) + 31) / 32), 0);
    out.bbits.assign(static_cast<size_t>((ipow(C, K + 1) + 31) / 32), 0);
    out.drec.resize(N);
    out.level_start = N;
    uint64_t prev_gram = 0;
    bool first_in_level = true;
    for (uint32_t s = 0; s < N; ++s) {
        const U32x4 r = tier.grec[s];
        out.drec[s] = U32x4{r.x, r.z, own_cnt[s], own_hs[s]};
        if (depth[s] == K && own_cnt[s] != 0) {
            const uint32_t g = static_cast<uint32_t>(gram[s]);
            out.wbits[g >> 5] |= 1u << (g & 31);
            out.wown.push_back(U32x2{own_cnt[s], own_hs[s]});
            out.has_word = true;
        } else if (depth[s] == K + 1) {
            const uint32_t g = static_cast<uint32_t>(gram[s]);
            if (first_in_level) { out.level_start = s; first_in_level = false; }
            else if (gram[s] <= prev_gram) return false;  // numbering is not lexicographic: do not trust ranks
            prev_gram = gram[s];
            out.bbits[g >> 5] |= 1u << (g & 31);
            out.dhit.push_back(U32x2{r.x, own_hs[s]});
        }
    }
    out.wrank.resize(out.wbits.size());
    uint32_t run = 0;
    for (size_t w = 0; w < out.wbits.size(); ++w) { out.wrank[w] = static_cast<uint16_t>(run); run += __builtin_popcount(out.wbits[w]); }
    out.brank.resize(out.bbits.size());
    out.bsuper.assign((out.bbits.size() + 63) / 64, 0);
    run = 0;
    uint32_t in_super = 0;
    for (size_t w = 0; w < out.bbits.size(); ++w) {
        if ((w & 63) == 0) { out.bsuper[w >> 6] = run; in_super = 0; }
        out.brank[w] = static_cast<uint16_t>(in_super);
        const uint32_t c = __builtin_popcount(out.bbits[w]);
        in_super += c;
        run += c;
    }
    out.available = true;
    return true;
}

}  // namespace daac
