#!/usr/bin/env python3
"""Transcribes the reference's golden test VECTORS into tests/golden/*.json (data only).

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py

Sources (all under /root/reference):
  * tests/aho_corasick_crate_test.rs:63-382 — the seven SearchTest tables
    (name, patterns, haystack, [(value, start, end)]) and the four collections (:50-59).
  * src/bytewise.rs:1243-1416, 761, 782 — layout pins (BASE/CHECK/FAIL arrays, block
    allocation cases, heap_bytes, num_states); README.md / doc known answers.
The second group is small enough that it is written out by hand below, with citations;
the first group is parsed from the `t!(...)` invocations.  Only the vectors are stored —
no reference source text.
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_rust_str(tok):
    assert tok[0] == '"' and tok[-1] == '"', tok
    body = tok[1:-1]
    # the tables only use plain ASCII without escapes; make sure of it
    assert "\\" not in body, tok
    return body


def parse_tables(path):
    src = open(path, encoding="utf-8").read()
    tables = {}
    for m in re.finditer(r"const (\w+): &'static \[SearchTest\] = &\[(.*?)\n\];", src, re.S):
        name, body = m.group(1), m.group(2)
        cases = []
        i = 0
        while True:
            j = body.find("t!(", i)
            if j < 0:
                break
            depth, k = 0, j + 2
            while True:
                ch = body[k]
                if ch == '"':
                    k = body.index('"', k + 1)
                elif ch in "([":
                    depth += 1
                elif ch in ")]":
                    depth -= 1
                    if depth == 0:
                        break
                k += 1
            inner = body[j + 3:k]
            i = k + 1
            mm = re.match(r"\s*(\w+)\s*,\s*&\[(.*?)\]\s*,\s*(\"[^\"]*\")\s*,\s*&\[(.*)\]\s*$", inner, re.S)
            assert mm, inner
            cname, pats, hay, matches = mm.groups()
            patterns = [parse_rust_str(t) for t in re.findall(r'"[^"]*"', pats)]
            triples = [[int(a), int(b), int(c)] for a, b, c in re.findall(r"\((\d+),\s*(\d+),\s*(\d+)\)", matches)]
            cases.append({"name": cname, "patterns": patterns, "haystack": parse_rust_str(hay), "matches": triples})
        tables[name] = cases
    return tables


def main():
    tables = parse_tables(os.path.join(REF, "tests/aho_corasick_crate_test.rs"))
    expect = {"BASICS": 42, "STANDARD": 11, "LEFTMOST": 24, "LEFTMOST_FIRST": 17,
              "LEFTMOST_LONGEST": 19, "NON_OVERLAPPING": 8, "OVERLAPPING": 15}
    for k, v in expect.items():
        assert len(tables[k]) == v, (k, len(tables[k]), v)
    doc = {
        "_source": "daachorse 4.0.0 tests/aho_corasick_crate_test.rs:63-382 (vectors), :50-59 (collections), :537-589 (runners)",
        "_triple_order": "(value, start, end)",
        "tables": tables,
        "collections": {
            "AC_STANDARD_NON_OVERLAPPING": ["BASICS", "NON_OVERLAPPING", "STANDARD"],
            "AC_STANDARD_OVERLAPPING": ["BASICS", "OVERLAPPING"],
            "AC_LEFTMOST_LONGEST": ["BASICS", "NON_OVERLAPPING", "LEFTMOST", "LEFTMOST_LONGEST"],
            "AC_LEFTMOST_FIRST": ["BASICS", "NON_OVERLAPPING", "LEFTMOST", "LEFTMOST_FIRST"],
        },
        "runners": [
            {"api": "find_iter", "kind": "Standard", "collection": "AC_STANDARD_NON_OVERLAPPING"},
            {"api": "find_stepper", "kind": "Standard", "collection": "AC_STANDARD_NON_OVERLAPPING"},
            {"api": "find_overlapping_iter", "kind": "Standard", "collection": "AC_STANDARD_OVERLAPPING"},
            {"api": "find_overlapping_stepper", "kind": "Standard", "collection": "AC_STANDARD_OVERLAPPING"},
            {"api": "leftmost_find_iter", "kind": "LeftmostLongest", "collection": "AC_LEFTMOST_LONGEST"},
            {"api": "leftmost_find_iter", "kind": "LeftmostFirst", "collection": "AC_LEFTMOST_FIRST"},
        ],
    }
    with open(os.path.join(HERE, "aho_corasick_vectors.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")

    # ---- hand-transcribed pins (values only), each with its citation ---------------------
    X = None
    pins = {
        "_source": "daachorse 4.0.0 in-module tests and doc examples; see `cite` on each entry",
        "double_array": {  # src/bytewise.rs:1243-1310
            "cite": "src/bytewise.rs:1243-1310",
            "patterns_bytes": [[0, 0], [0, 2], [1, 2], [2]],
            "base": [4, X, X, X, 8, 1, X, X, X, X, X],
            "check": [0, 1, 2, 2, 0, 1, 2, 7, 0, 9, 2],
            "fail": [0, 0, 0, 6, 0, 0, 0, 0, 4, 0, 6],
        },
        "num_states": [  # src/bytewise.rs:1313-1325, 782
            {"cite": "src/bytewise.rs:1313-1325", "patterns": ["abba", "baaba", "ababa"], "num_states": 13},
            {"cite": "src/bytewise.rs:777-783", "patterns": ["bcd", "ab", "a"], "num_states": 6},
        ],
        "heap_bytes": [  # src/bytewise.rs:756-762
            {"cite": "src/bytewise.rs:756-762", "patterns": ["bcd", "ab", "a"], "heap_bytes": 4132},
        ],
        "input_order": {  # src/bytewise.rs:1328-1337
            "cite": "src/bytewise.rs:1328-1337",
            "sorted": [["ababa", 0], ["abba", 1], ["baaba", 2]],
            "unsorted": [["abba", 1], ["baaba", 2], ["ababa", 0]],
        },
        "n_blocks": [  # src/bytewise.rs:1340-1416; patterns are described, not listed
            {"cite": "src/bytewise.rs:1340-1354", "name": "1_1",
             "gen": [{"prefix": [], "ranges": [[0, 253]]}],
             "num_states": 255, "states_len": 256, "base_of": {"0": 254}},
            {"cite": "src/bytewise.rs:1356-1372", "name": "1_2",
             "gen": [{"prefix": [], "ranges": [[0, 0], [2, 2], [4, 255]]}],
             "num_states": 255, "states_len": 512, "base_of": {"0": 256}},
            {"cite": "src/bytewise.rs:1374-1393", "name": "2_1",
             "gen": [{"prefix": [], "ranges": [[0, 127]]}, {"prefix": [0], "ranges": [[0, 125]]}],
             "num_states": 255, "states_len": 256, "base_of": {"0": 128, "128": 126}},
            {"cite": "src/bytewise.rs:1395-1416", "name": "2_2",
             "gen": [{"prefix": [], "ranges": [[0, 127]]}, {"prefix": [0], "ranges": [[0, 0], [2, 2], [4, 127]]}],
             "num_states": 255, "states_len": 512, "base_of": {"0": 128, "128": 256}},
        ],
        "invalid_blob": {  # src/bytewise.rs:1496-1507: 21 zero bytes must be rejected
            "cite": "src/bytewise.rs:1496-1507", "blob": [0] * 21,
        },
        "serialize_roundtrip": [  # src/bytewise.rs:1451-1493
            {"cite": "src/bytewise.rs:1451-1462", "kind": "Standard", "patterns": ["abba", "baaba", "ababa"]},
            {"cite": "src/bytewise.rs:1464-1481", "kind": "LeftmostLongest", "patterns": ["abba", "baaba", "ababa"]},
            {"cite": "src/bytewise.rs:1483-1493", "kind": "Standard", "patterns": []},
        ],
        "known_answers": [  # (start, end, value) triples from README.md / crate docs
            {"cite": "README.md:57-71", "api": "find_overlapping_iter", "kind": "Standard",
             "patterns": ["bcd", "ab", "a"], "haystack": "abcd", "matches_sev": [[0, 1, 2], [0, 2, 1], [1, 4, 0]]},
            {"cite": "README.md:82-93", "api": "find_iter", "kind": "Standard",
             "patterns": ["bcd", "ab", "a"], "haystack": "abcd", "matches_sev": [[0, 1, 2], [1, 4, 0]]},
            {"cite": "README.md:104-115", "api": "leftmost_find_iter", "kind": "LeftmostLongest",
             "patterns": ["ab", "a", "abcd"], "haystack": "abcd", "matches_sev": [[0, 4, 2]]},
            {"cite": "README.md:130-141", "api": "leftmost_find_iter", "kind": "LeftmostFirst",
             "patterns": ["ab", "a", "abcd"], "haystack": "abcd", "matches_sev": [[0, 2, 0]]},
            {"cite": "README.md:152-166", "api": "find_overlapping_iter", "kind": "Standard",
             "patvals": [["bcd", 0], ["ab", 10], ["a", 20]], "haystack": "abcd",
             "matches_sev": [[0, 1, 20], [0, 2, 10], [1, 4, 0]]},
            {"cite": "src/bytewise.rs:397-408", "api": "find_overlapping_no_suffix_iter", "kind": "Standard",
             "patterns": ["bcd", "cd", "abc"], "haystack": "abcd", "matches_sev": [[0, 3, 2], [1, 4, 0]]},
            {"cite": "src/bytewise/iter.rs:484-509", "api": "find_overlapping_no_suffix_iter", "kind": "Standard",
             "patterns": ["a", "ab", ""], "haystack": "ab", "matches_sev": [[0, 0, 2], [0, 1, 0], [0, 2, 1]]},
            {"cite": "src/bytewise/builder.rs:188-202", "api": "find_iter", "kind": "Standard",
             "patvals": [["bcd", 0], ["ab", 1], ["a", 2], ["e", 1]], "haystack": "abcde",
             "matches_sev": [[0, 1, 2], [1, 4, 0], [4, 5, 1]]},
            {"cite": "src/bytewise/builder.rs:76-87", "api": "leftmost_find_iter", "kind": "LeftmostLongest",
             "patterns": ["ab", "abcd"], "haystack": "abcd", "matches_sev": [[0, 4, 1]]},
        ],
        "empty_pattern_set": {"cite": "src/bytewise.rs:1418-1431",
                              "note": "no match for every 1- and 2-byte haystack"},
        "invalid_option": {"cite": "tests/invalid_option_test.rs:3-9", "patterns": ["pattern"],
                           "num_free_blocks": 4294967295, "must_fail": True},
        "matchkind_mismatch": {  # tests/matchkind_mismatch_test.rs:3-71
            "cite": "tests/matchkind_mismatch_test.rs:3-71",
            "must_fail": [
                {"kind": "Standard", "api": "leftmost_find_iter"},
                {"kind": "LeftmostLongest", "api": "find_iter"},
                {"kind": "LeftmostLongest", "api": "find_overlapping_iter"},
                {"kind": "LeftmostLongest", "api": "find_overlapping_no_suffix_iter"},
                {"kind": "LeftmostFirst", "api": "find_iter"},
                {"kind": "LeftmostFirst", "api": "find_overlapping_iter"},
                {"kind": "LeftmostFirst", "api": "find_overlapping_no_suffix_iter"},
            ],
        },
    }
    with open(os.path.join(HERE, "bytewise_pins.json"), "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
        f.write("\n")
    # ---- charwise pins (values only), each with its citation ------------------------------------------
    FW = {"A": "\uff21", "B": "\uff22", "C": "\uff23"}  # fullwidth letters used by the reference's layout test
    cpins = {
        "_source": "daachorse 4.0.0 src/charwise.rs / src/charwise/*.rs in-module tests and docs; see `cite`",
        "runners": "the six charwise runners of tests/aho_corasick_crate_test.rs:592-645 use the same vector tables",
        "double_array": {  # src/charwise.rs:1200-1267
            "cite": "src/charwise.rs:1200-1267",
            "patterns": [FW["A"] * 2, FW["A"] + FW["C"], FW["B"] + FW["C"], FW["C"]],
            "base": [4, X, X, X, 8, X, 3, X, X, X, X],
            "check": [1, 1, 6, 1, 0, 0, 0, 1, 4, 4, 1],
            "fail": [0, 1, 5, 1, 0, 0, 0, 1, 4, 5, 1],
        },
        "num_states": [{"cite": "src/charwise.rs:1269-1281", "patterns": ["\uff41\uff42\uff42\uff41", "\uff42\uff41\uff41\uff42\uff41", "\uff41\uff42\uff41\uff42\uff41"], "num_states": 13},
                       {"cite": "src/charwise.rs:769-781", "patterns": ["bcd", "ab", "a"], "num_states": 6}],
        "num_elements": [{"cite": "src/charwise.rs:785-796", "patterns": ["bcd", "ab", "a"], "num_elements": 8}],
        "heap_bytes": [{"cite": "src/charwise.rs:800-811", "patterns": ["bcd", "ab", "a"], "heap_bytes": 568}],
        "input_order": {"cite": "src/charwise.rs:1283-1294",
                        "sorted": [["\uff41\uff42\uff41\uff42\uff41", 0], ["\uff41\uff42\uff42\uff41", 1], ["\uff42\uff41\uff41\uff42\uff41", 2]],
                        "unsorted": [["\uff41\uff42\uff42\uff41", 1], ["\uff42\uff41\uff41\uff42\uff41", 2], ["\uff41\uff42\uff41\uff42\uff41", 0]]},
        "n_blocks": [  # src/charwise.rs:1297-1370: single-char patterns U+0000.., described not listed
            {"cite": "src/charwise.rs:1297-1311", "name": "1_1", "gen": [{"prefix": [], "range": [0, 0x7d]}],
             "num_states": 127, "states_len": 128, "base_of": {"0": 0x7e}},
            {"cite": "src/charwise.rs:1313-1327", "name": "1_2", "gen": [{"prefix": [], "range": [0, 0x7e]}],
             "num_states": 128, "states_len": 256, "base_of": {"0": 0x80}},
            {"cite": "src/charwise.rs:1329-1349", "name": "2_1", "gen": [{"prefix": [], "range": [0, 0x7f]}, {"prefix": [0], "range": [0, 0x7d]}],
             "num_states": 255, "states_len": 256, "base_of": {"0": 0x80, "128": 0x7e}},
            {"cite": "src/charwise.rs:1351-1370", "name": "2_2", "gen": [{"prefix": [], "range": [0, 0x7f]}, {"prefix": [0], "range": [0, 0x7e]}],
             "num_states": 256, "states_len": 384, "base_of": {"0": 0x80, "128": 0x100}},
        ],
        "multibyte_zero_length": [  # src/charwise.rs:1373-1456
            {"cite": "src/charwise.rs:1373-1408", "api": "find_overlapping_iter", "kind": "Standard",
             "patterns": ["a", "\u00e6", "\u3042", ""], "haystack": "\u3044\u3042abc\u00c6\u00e6\u3046",
             "matches_sev": [[0, 0, 3], [3, 3, 3], [3, 6, 2], [6, 6, 3], [6, 7, 0], [7, 7, 3], [8, 8, 3], [9, 9, 3], [11, 11, 3],
                             [11, 13, 1], [13, 13, 3], [16, 16, 3]]},
            {"cite": "src/charwise.rs:1410-1433", "api": "find_iter", "kind": "Standard",
             "patterns": ["a", "\u00e6", "\u3042", ""], "haystack": "\u3044\u3042abc\u00c6\u00e6\u3046",
             "matches_sev": [[0, 0, 3], [3, 3, 3], [6, 6, 3], [7, 7, 3], [8, 8, 3], [9, 9, 3], [11, 11, 3], [13, 13, 3], [16, 16, 3]]},
            {"cite": "src/charwise.rs:1435-1456", "api": "leftmost_find_iter", "kind": "LeftmostLongest",
             "patterns": ["a", "\u00e6", "\u3042", ""], "haystack": "\u3044\u3042abc\u00c6\u00e6\u3046",
             "matches_sev": [[0, 0, 3], [3, 6, 2], [6, 7, 0], [8, 8, 3], [9, 9, 3], [11, 13, 1], [16, 16, 3]]},
        ],
        "known_answers": [
            {"cite": "README.md:181-192", "api": "find_iter", "kind": "Standard",
             "patterns": ["\u5168\u4e16\u754c", "\u4e16\u754c", "\u306b"], "haystack": "\u5168\u4e16\u754c\u4e2d\u306b",
             "matches_sev": [[0, 9, 0], [12, 15, 2]]},
            {"cite": "src/charwise/iter.rs:595-620", "api": "find_overlapping_no_suffix_iter", "kind": "Standard",
             "patterns": ["a", "ab", ""], "haystack": "ab", "matches_sev": [[0, 0, 2], [0, 1, 0], [0, 2, 1]]},
        ],
        "decoder": {  # src/charwise/iter.rs:543-593: code points and their END offsets in the concatenated string
            "cite": "src/charwise/iter.rs:543-593",
            "code_points": [0x0, 0x1, 0x2, 0x4, 0x8, 0x10, 0x1f, 0x20, 0x40, 0x7f, 0x80, 0x100, 0x1ff, 0x200, 0x400, 0x7ff, 0x800, 0x1000,
                            0x1fff, 0x2000, 0x4000, 0x8000, 0xffff, 0x10000, 0x1ffff, 0x20000, 0x40000, 0x80000, 0x100000, 0x10ffff],
            "end_offsets": [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 22, 25, 28, 31, 34, 37, 40, 43, 47, 51, 55, 59, 63, 67, 71],
        },
        "mapper": {"cite": "src/charwise/mapper.rs:87-99", "freqs": [3, 6, 0, 2, 3, 0, 3], "codes": [1, 0, X, 4, 2, X, 3]},
    }
    cpins = json.loads(json.dumps(cpins).replace("\\\\u", "\\u"))
    with open(os.path.join(HERE, "charwise_pins.json"), "w") as f:
        json.dump(cpins, f, indent=1, sort_keys=True)
        f.write("\n")
    n = sum(len(v) for v in tables.values())
    print(f"wrote {n} vector cases in {len(tables)} tables + pins")


if __name__ == "__main__":
    main()
