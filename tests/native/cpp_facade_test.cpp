// The C++ façade (include/daachorse_amd.hpp) driven the way the reference's own doc tests drive the
// crate (README.md:57-192, src/bytewise/builder.rs:38-55, tests/matchkind_mismatch_test.rs).
//   usage: cpp_facade_test host   (no GPU: construction, serialisation, errors)
//          cpp_facade_test gpu    (+ the scans)
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/daachorse_amd.hpp"

using namespace daachorse;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static bool same(const std::vector<Match> &got, const std::vector<Match> &want) { return got == want; }

int main(int argc, char **argv) {
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    auto pma = DoubleArrayAhoCorasick::new_({"bcd", "ab", "a"}).unwrap();
    CHECK(pma.num_states() == 6);       // src/bytewise.rs:777-783
    CHECK(pma.heap_bytes() == 4132);    // src/bytewise.rs:756-762
    CHECK(pma.match_kind() == MatchKind::Standard);
    const std::string blob = pma.serialize();
    auto again = DoubleArrayAhoCorasick::deserialize(blob + "xyz").unwrap();
    CHECK(again.second == blob.size() && again.first.serialize() == blob);
    CHECK(DoubleArrayAhoCorasick::deserialize(std::string(21, '\0')).is_err());  // src/bytewise.rs:1496-1507
    CHECK(DoubleArrayAhoCorasickBuilder().num_free_blocks(0xffffffffu).build({"pattern"}).is_err());  // tests/invalid_option_test.rs
    // the charwise twin (README.md:171-193): same surface, UTF-8 patterns, blobs of its own format
    auto cw = CharwiseDoubleArrayAhoCorasick::new_({"全世界", "世界", "に"}).unwrap();
    CHECK(cw.match_kind() == MatchKind::Standard);
    const std::string cblob = cw.serialize();
    auto cagain = CharwiseDoubleArrayAhoCorasick::deserialize(cblob).unwrap();
    CHECK(cagain.second == cblob.size() && cagain.first.serialize() == cblob);
    CHECK(DoubleArrayAhoCorasick::deserialize(cblob).is_err());
    CHECK(CharwiseDoubleArrayAhoCorasickBuilder().build({"\xff\xfe"}).is_err());  // not UTF-8
    if (!gpu) { std::printf("OK host\n"); return 0; }

    CHECK(same(pma.find_overlapping_iter("abcd").collect(), {Match(0, 1, 2), Match(0, 2, 1), Match(1, 4, 0)}));  // README.md:57-71
    CHECK(same(pma.find_iter("abcd").collect(), {Match(0, 1, 2), Match(1, 4, 0)}));                               // README.md:82-93
    auto it = pma.find_iter("abcd");  // lazy, one next() at a time like the doc test
    auto m = it.next();
    CHECK(m && m->start() == 0 && m->end() == 1 && m->value() == 2);
    m = it.next();
    CHECK(m && m->start() == 1 && m->end() == 4 && m->value() == 0);
    CHECK(!it.next());
    auto ll = DoubleArrayAhoCorasickBuilder().match_kind(MatchKind::LeftmostLongest).build({"ab", "a", "abcd"}).unwrap();
    CHECK(same(ll.leftmost_find_iter("abcd").collect(), {Match(0, 4, 2)}));  // README.md:104-115
    auto lf = DoubleArrayAhoCorasickBuilder().match_kind(MatchKind::LeftmostFirst).build({"ab", "a", "abcd"}).unwrap();
    CHECK(same(lf.leftmost_find_iter("abcd").collect(), {Match(0, 2, 0)}));  // README.md:130-141
    auto wv = DoubleArrayAhoCorasick::with_values({{"bcd", 0}, {"ab", 10}, {"a", 20}}).unwrap();
    CHECK(same(wv.find_overlapping_iter("abcd").collect(), {Match(0, 1, 20), Match(0, 2, 10), Match(1, 4, 0)}));  // README.md:152-166
    bool panicked = false;
    try { ll.find_iter(""); } catch (const PanicError &e) { panicked = std::string(e.what()) == "Error: match_kind must be standard."; }
    CHECK(panicked);  // tests/matchkind_mismatch_test.rs:5-11
    panicked = false;
    try { pma.leftmost_find_iter(""); } catch (const PanicError &e) { panicked = std::string(e.what()) == "Error: match_kind must be leftmost."; }
    CHECK(panicked);  // tests/matchkind_mismatch_test.rs:65-71
    auto cit = cw.find_iter("全世界中に");  // README.md:184-192, byte offsets
    m = cit.next();
    CHECK(m && m->start() == 0 && m->end() == 9 && m->value() == 0);
    m = cit.next();
    CHECK(m && m->start() == 12 && m->end() == 15 && m->value() == 2);
    CHECK(!cit.next());
    auto cll = CharwiseDoubleArrayAhoCorasickBuilder().match_kind(MatchKind::LeftmostLongest).build({"世界", "全世界", "世"}).unwrap();
    CHECK(same(cll.leftmost_find_iter("全世界中に世").collect(), {Match(0, 9, 1), Match(15, 18, 2)}));
    // Iterator::count() on the four iterators without materialising, and the list left in device memory
    CHECK(pma.find_overlapping_iter_count("abcd") == 3 && pma.find_iter_count("abcd") == 2);
    CHECK(pma.find_overlapping_no_suffix_iter_count("abcd") == 3 && ll.leftmost_find_iter_count("abcd") == 1);
    CHECK(cw.find_overlapping_iter_count("全世界中に") == 3);
    panicked = false;
    try { ll.find_overlapping_iter_count("abcd"); } catch (const PanicError &) { panicked = true; }
    CHECK(panicked);
    auto dm = pma.find_overlapping_device("abcdabcd");
    CHECK(dm.size() == 6 && same(dm.to_host(0, 3), {Match(0, 1, 2), Match(0, 2, 1), Match(1, 4, 0)}));
    CHECK(same(dm.to_host(3, 3), {Match(4, 5, 2), Match(4, 6, 1), Match(5, 8, 0)}));
    CHECK(pma.find_overlapping_device("zzz").size() == 0);
    auto stepper = pma.find_overlapping_stepper();  // the text in three pieces, a pattern across each cut
    std::vector<Match> fed;
    for (const char *piece : {"a", "bc", "d"})
        for (const Match &x : stepper.feed(piece)) fed.push_back(x);
    CHECK(same(fed, {Match(0, 1, 2), Match(0, 2, 1), Match(1, 4, 0)}));
    std::printf("OK gpu\n");
    return 0;
}
