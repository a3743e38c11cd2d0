import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth
from oracle import oracle as orc
pats3 = synth.patterns_cfg3(30000)
long_pats = [b"abcdefghijklmnop", b"bcdefghijklmnopq", b"mnopqrs", b"ponmlkjihg", b"qrstuv", b"a", b"op", b"nop", b"lmnopqrstuvwxyzabc"]
cases = [(synth.patterns_cfg1(), synth.uniform_haystack(9000, 3, synth.ALPHA_ABCD)),
         (long_pats, np.frombuffer((b"abcdefghijklmnopqrstuvwxyzabc" * 400)[:11000], dtype=np.uint8)),
         (long_pats, synth.uniform_haystack(20000, 4, b"abcdefghijklmnopqrstuvwxyz")),
         (synth.patterns_cfg2(500), synth.wordsoup_haystack(300000, 8, synth.patterns_cfg2(500), 13, noise_256=30)),
         (pats3, synth.uniform_haystack((1 << 20) + 777, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
         (pats3, synth.wordsoup_haystack(1 << 20, synth.SEEDS["cfg3_dense"], pats3, 20))]
for ci, (pats, hay) in enumerate(cases):
    o = orc.OraclePma.build(pats)
    want = o.find_overlapping_iter(hay)
    for tiles, budget, shift in ((64, 158 * 1024, 0), (1, 158 * 1024, 3), (2, 24 * 1024, 9)):
        print("case", ci, tiles, budget, shift, flush=True)
        da.set_option("emit_tiles", tiles)
        da.set_option("gram_lds_budget", budget)
        p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        dev = torch.from_numpy(np.concatenate([np.zeros(shift, dtype=np.uint8), hay])).cuda()[shift:]
        try:
            got = p.scan(ScanMode.FindOverlapping, dev, engine=Engine.Gram)
        except da.DaachorseError as e:
            print("  error", e, p.info().gram2_available, p.info().gram2_k, flush=True)
            continue
        ok = len(got) == len(want) and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and np.array_equal(got["value"], want["value"])
        print("  scan", len(got), len(want), ok, "k", p.info().gram2_k, flush=True)
        if not ok:
            n = min(len(got), len(want))
            bad = np.nonzero((got["start"][:n] != want["start"][:n]) | (got["end"][:n] != want["end"][:n]) | (got["value"][:n] != want["value"][:n]))[0]
            if len(bad):
                b = int(bad[0]); print("first bad", b, "of", n, "nbad", len(bad)); print(got[max(0,b-2):b+4]); print(want[max(0,b-2):b+4])
