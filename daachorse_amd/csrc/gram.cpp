// GRAM engine tables (host side) — see gram.hpp.
#include "gram.hpp"

#include <algorithm>
#include <utility>

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
    std::vector<uint32_t> new_of_old(p.states.size(), 0xffffffffu);
    for (uint32_t s = 0; s < N; ++s) new_of_old[tier.old_of_new[s]] = s;

    uint32_t K = 0, lds = 0;
    for (uint32_t cand : {3u, 2u}) {
        const uint64_t ngram = ipow(C, cand), bwords = (ipow(C, cand + 1) + 31) / 32;
        if (bwords * 32 >= (1ull << 31) || ngram >= (1ull << 24)) continue;
        // the 8-byte hit record of a depth-(K+1) state carries no count: it must be 0/1 and visible in the sum
        bool regular = true;
        for (uint32_t s = 0; s < N && regular; ++s)
            if (depth[s] == cand + 1 && own_cnt[s] != (own_hs[s] != 0 ? 1u : 0u)) regular = false;
        if (!regular) continue;
        // CID / COMBO: everything of length <= K that ends after these K classes, found with the
        // reference's own delta (a class-0 byte resets to ROOT); distinct {count, hsum} pairs share an id
        std::vector<uint16_t> cid(static_cast<size_t>(ngram), 0);
        std::vector<U32x2> combo{U32x2{0, 0}};
        std::vector<std::pair<uint64_t, uint16_t>> seen;  // sorted (count << 32 | hsum) -> id
        bool ok = true;
        for (uint32_t g = 0; g < ngram && ok; ++g) {
            uint32_t st = kRoot;
            for (uint32_t i = 0; i < cand; ++i) st = p.next_state(st, rep[(g / static_cast<uint32_t>(ipow(C, cand - 1 - i))) % C]);
            const OutSum o = tier.ssum[new_of_old[st]];
            if (o.cnt == 0) continue;
            const uint64_t key = (static_cast<uint64_t>(o.cnt) << 32) | o.hsum;
            auto it = std::lower_bound(seen.begin(), seen.end(), std::make_pair(key, uint16_t(0)));
            if (it == seen.end() || it->first != key) {
                if (combo.size() >= 65535) { ok = false; break; }
                it = seen.insert(it, std::make_pair(key, static_cast<uint16_t>(combo.size())));
                combo.push_back(U32x2{o.cnt, o.hsum});
            }
            cid[g] = it->second;
        }
        if (!ok) continue;
        const uint64_t bytes = 1024 + pad16(ngram * 2) + pad16(combo.size() * 8) + pad16(bwords * 4 + 8) + pad16((bwords + 1) / 2) +
                               pad16(((bwords + 7) / 8) * 4) + 64;
        if (bytes > lds_budget) continue;
        K = cand;
        lds = static_cast<uint32_t>(bytes);
        out.cid = std::move(cid);
        out.combo = std::move(combo);
        break;
    }
    if (K == 0) return false;

    out.K = K;
    out.C = C;
    out.N = N;
    out.unused_byte = rep[0];
    out.cls = tier.cls;
    out.lds_bytes = lds;
    out.has_short = out.combo.size() > 1;

    // ---- B_{K+1} with its rank directory; per-state walk records ------------------------------------
    // Breadth-first ids order each level lexicographically by class string (children are numbered
    // parent by parent, class-ascending), i.e. by gram index: the rank of a set bit in B_{K+1} is
    // the state's offset within its level.
    out.bbits.assign(static_cast<size_t>((ipow(C, K + 1) + 31) / 32), 0);
    out.drec.resize(N);
    out.level_start = N;
    uint64_t prev_gram = 0;
    bool first_in_level = true;
    for (uint32_t s = 0; s < N; ++s) {
        const U32x4 r = tier.grec[s];
        out.drec[s] = U32x4{r.x, r.z, own_cnt[s], own_hs[s]};
        if (depth[s] == K + 1) {
            const uint32_t g = static_cast<uint32_t>(gram[s]);
            if (first_in_level) { out.level_start = s; first_in_level = false; }
            else if (gram[s] <= prev_gram) return false;  // numbering is not lexicographic: do not trust ranks
            prev_gram = gram[s];
            out.bbits[g >> 5] |= 1u << (g & 31);
            out.dhit.push_back(U32x2{r.x, own_hs[s]});
            out.cfirst.push_back(r.z);
        }
    }
    // ---- tail records with the pattern's h inline (round 6) ------------------------------------------------
    // Below the records the walkers start on (depth K + 2: most walkers of uniform text end there, and that record stays as it was), a
    // state under which the trie is ONE path of at most eight edges with exactly ONE pattern end on it — a dictionary word's last
    // letters — becomes {h of that pattern, 1 << 31 | edges | index of the ending node << 4, path bytes 0-3, path bytes 4-7}: the walker
    // compares the path with the next eight text bytes in one step instead of fetching a record per letter (text made of dictionary
    // words: every word walks its letters).  first_child is below 2^27, so bit 31 of the second word marks the form.  Children carry
    // larger numbers than their parents: one pass from the back knows every subtree.
    {
        std::vector<uint8_t> plen(N, 0xff), nend(N, 0);   // edges of the single path below s (0xff: it branches, is too long, or a node ends two patterns)
        for (uint32_t s = N; s-- > 0;) {
            const U32x4 r = tier.grec[s];
            const uint32_t kids = static_cast<uint32_t>(__builtin_popcount(r.x));
            if (own_cnt[s] > 1) continue;
            if (kids == 0) { plen[s] = 0; nend[s] = static_cast<uint8_t>(own_cnt[s]); continue; }
            if (kids != 1) continue;
            const uint32_t c = r.z;
            if (plen[c] == 0xff || plen[c] >= 8) continue;
            plen[s] = static_cast<uint8_t>(plen[c] + 1);
            nend[s] = static_cast<uint8_t>(std::min<uint32_t>(3u, own_cnt[s] + nend[c]));
        }
        for (uint32_t s = 0; s < N; ++s) {
            if (depth[s] < K + 3 || plen[s] == 0xff || plen[s] == 0 || nend[s] != 1) continue;
            uint64_t bytes = 0;
            uint32_t cur = s, at = own_cnt[s] ? 0u : 0xffu, h = own_cnt[s] ? own_hs[s] : 0u;
            for (uint32_t i = 0; i < plen[s]; ++i) {
                const U32x4 r = tier.grec[cur];
                bytes |= static_cast<uint64_t>(rep[__builtin_ctz(r.x)]) << (8 * i);
                cur = r.z;
                if (own_cnt[cur]) { at = i + 1; h = own_hs[cur]; }
            }
            out.drec[s] = U32x4{h, 0x80000000u | plen[s] | (at << 4), static_cast<uint32_t>(bytes), static_cast<uint32_t>(bytes >> 32)};
            ++out.n_tail;
        }
    }
    if (out.bbits.size() & 1) out.bbits.push_back(0);  // whole 64-bit pairs
    out.brank.resize(out.bbits.size() / 2);
    out.bsuper.assign((out.bbits.size() + 7) / 8, 0);
    uint32_t run = 0, in_super = 0;
    for (size_t w = 0; w < out.bbits.size(); ++w) {
        if ((w & 7) == 0) { out.bsuper[w >> 3] = run; in_super = 0; }
        if ((w & 1) == 0) out.brank[w >> 1] = static_cast<uint8_t>(in_super);  // <= 6 * 32
        const uint32_t c = __builtin_popcount(out.bbits[w]);
        in_super += c;
        run += c;
    }
    out.available = true;
    return true;
}

}  // namespace daac
