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
    n = sum(len(v) for v in tables.values())
    print(f"wrote {n} vector cases in {len(tables)} tables + pins")


if __name__ == "__main__":
    main()
