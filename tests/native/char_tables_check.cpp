// Host-logic test for the charwise device tables (no GPU needed).
//   usage: char_tables_check <leftmost-longest blob> <standard blob of the same patterns>
// Both automata have the same trie and the same slot layout (the placement only looks at the trie
// and the code mapper), so the classic failure links the device tables derive for the leftmost
// automaton must equal the Standard automaton's own links on every real slot, and the output chain
// sums must match a literal walk.  Prints "OK <slots checked>" or "MISMATCH ...".
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../daachorse_amd/csrc/charwise.hpp"

using namespace daac;

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::vector<uint8_t> lb = slurp(argv[1]), sb = slurp(argv[2]);
    HostCharPma lm, sd;
    size_t used = 0;
    if (HostCharPma::deserialize(lb.data(), lb.size(), lm, &used) != DAAC_OK || used != lb.size()) { std::printf("BADBLOB leftmost: %s\n", last_error_cstr()); return 1; }
    if (HostCharPma::deserialize(sb.data(), sb.size(), sd, &used) != DAAC_OK || used != sb.size()) { std::printf("BADBLOB standard: %s\n", last_error_cstr()); return 1; }
    if (lm.states.size() != sd.states.size() || lm.table != sd.table) { std::printf("MISMATCH layouts differ\n"); return 1; }
    CharTables t, ts;
    build_char_tables(lm, t);
    build_char_tables(sd, ts);
    if (!ts.fail_plain.empty()) { std::printf("MISMATCH standard automaton got fail_plain\n"); return 1; }
    if (t.fail_plain.size() != lm.states.size()) { std::printf("MISMATCH fail_plain size\n"); return 1; }
    size_t checked = 0;
    // walk the trie from ROOT through CHECK-confirmed children only
    std::vector<uint32_t> stack{kRoot};
    while (!stack.empty()) {
        const uint32_t s = stack.back();
        stack.pop_back();
        ++checked;
        if (sd.states[s].base != lm.states[s].base) { std::printf("MISMATCH base at %u\n", s); return 1; }
        if (t.fail_plain[s] != sd.states[s].fail) {
            std::printf("MISMATCH classic link of slot %u: %u vs %u\n", s, t.fail_plain[s], sd.states[s].fail);
            return 1;
        }
        if (lm.states[s].base == 0) continue;
        for (uint32_t code = 0; code < lm.alphabet_size; ++code) {
            const uint32_t child = lm.states[s].base ^ code;
            if (child != kRoot && lm.states[child].check == s) stack.push_back(child);
        }
    }
    for (size_t i = 0; i < lm.outputs.size(); ++i) {
        uint32_t cnt = 0, hs = 0;
        for (uint32_t op = static_cast<uint32_t>(i) + 1; op != 0; op = lm.outputs[op - 1].parent) {
            ++cnt;
            hs += match_hash32(lm.outputs[op - 1].value, lm.outputs[op - 1].length);
        }
        if (t.osum[i].cnt != cnt || t.osum[i].hsum != hs) { std::printf("MISMATCH osum at %zu\n", i); return 1; }
    }
    std::printf("OK %zu\n", checked);
    return 0;
}
