#!/usr/bin/env python3
"""Randomised soak of the host side (no GPU): the C++ builders against the oracle (byte-identical blobs), and the device
tables against the literal automaton through the native checkers (tests/native/*.cpp).
usage: python tools/host_soak.py [seconds] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import daachorse_amd as da
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
tmp = tempfile.mkdtemp()
csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
exes = {}
for name, srcs in (("repack_check", ["pma.cpp", "repack.cpp"]), ("gram_check", ["pma.cpp", "repack.cpp", "gram.cpp"]),
                   ("char_tables_check", ["pma.cpp", "repack.cpp", "charwise.cpp"])):
    exes[name] = os.path.join(tmp, name)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exes[name], os.path.join(ROOT, "tests", "native", name + ".cpp")] +
                          [os.path.join(csrc, s) for s in srcs])
ALPHAS = [list("ab"), list("abcdef"), [chr(c) for c in range(97, 123)], [chr(c) for c in range(0x3041, 0x3060)], list("aé世") + ["\U0001F600"]]
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    A = ALPHAS[int(rng.integers(0, len(ALPHAS)))]
    ascii_only = all(len(c.encode()) == 1 for c in A)
    npat = int(rng.choice([1, 3, 30, 300, 3000]))
    maxlen = int(rng.choice([2, 5, 9, 20]))
    pats = ["".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(1, maxlen + 1)))) for _ in range(npat)]
    if rng.random() < 0.3:
        pats += [pats[0] + q for q in pats[:4]] + pats[:3]  # prefix families and duplicates
    if rng.random() < 0.1:
        pats.insert(int(rng.integers(0, len(pats) + 1)), "")
    vals = [int(v) for v in rng.integers(0, 2**32, size=len(pats), dtype=np.uint64)] if rng.random() < 0.5 else None
    nfb = int(rng.choice([1, 2, 16, 64]))
    for kind in (0, 1, 2):
        oc = orc.OracleCharwisePma.build(pats, values=vals, kind=kind, num_free_blocks=nfb)
        b = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(kind).num_free_blocks(nfb)
        pc = b.build_with_values(zip(pats, vals)) if vals else b.build(pats)
        assert pc.serialize() == oc.serialize(), ("charwise", kind, pats[:5])
        ob = orc.OraclePma.build(pats, values=vals, kind=kind, num_free_blocks=nfb)
        b = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).num_free_blocks(nfb)
        pb = b.build_with_values(zip(pats, vals)) if vals else b.build(pats)
        assert pb.serialize() == ob.serialize(), ("bytewise", kind, pats[:5])
        assert da.DoubleArrayAhoCorasick.deserialize(pb.serialize())[0].serialize() == ob.serialize()
    # device tables vs the literal automaton (Standard kind), text made of patterns and noise
    text = "".join(pats[int(i)] if rng.random() < 0.6 else A[int(rng.integers(0, len(A)))] for i in rng.integers(0, len(pats), size=3000))
    hay = os.path.join(tmp, "h.bin")
    open(hay, "wb").write(text.encode()[:60000] or b"x")
    blob = os.path.join(tmp, "a.blob")
    open(blob, "wb").write(orc.OraclePma.build(pats).serialize())
    out = subprocess.check_output([exes["repack_check"], blob, str(int(rng.choice([1024, 8192, 98304]))), str(int(rng.choice([-1, 0, 1, 2]))), hay]).decode()
    assert out.startswith(("OK", "UNAVAILABLE")), out
    out = subprocess.check_output([exes["gram_check"], blob, str(int(rng.choice([9216, 40000, 161792]))), hay]).decode()
    assert out.startswith(("OK", "UNAVAILABLE")), out
    lb, sb = os.path.join(tmp, "l.blob"), os.path.join(tmp, "s.blob")
    open(lb, "wb").write(orc.OracleCharwisePma.build(pats, kind=1).serialize())
    open(sb, "wb").write(orc.OracleCharwisePma.build(pats, kind=0).serialize())
    out = subprocess.check_output([exes["char_tables_check"], lb, sb]).decode()
    assert out.startswith("OK"), out
    n += 1
print(f"host soak ok: {n} pattern sets x 3 kinds x 2 builders + table checks in {time.time() - t0:.0f} s (seed {seed})")
