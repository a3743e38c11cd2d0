import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth
from oracle import oracle as orc
pats = synth.patterns_cfg1()
hay = synth.uniform_haystack(int(sys.argv[1]) if len(sys.argv) > 1 else 9000, 3, synth.ALPHA_ABCD)
o = orc.OraclePma.build(pats)
p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
p.upload()
i = p.info()
print("info", i.gram2_available, i.gram2_k, flush=True)
want = o.find_overlapping_iter(hay)
dev = torch.from_numpy(hay).cuda()
print("count", p.count(ScanMode.FindOverlapping, dev), len(want), flush=True)
got = p.scan(ScanMode.FindOverlapping, dev, engine=Engine.Gram)
print("scan ok", len(got), len(want), flush=True)
ok = len(got) == len(want) and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and np.array_equal(got["value"], want["value"])
print("equal", ok)
if not ok:
    n = min(len(got), len(want))
    bad = np.nonzero((got["start"][:n] != want["start"][:n]) | (got["end"][:n] != want["end"][:n]) | (got["value"][:n] != want["value"][:n]))[0]
    print("first bad", bad[:5], got[bad[0] - 2:bad[0] + 3] if len(bad) else None, want[bad[0] - 2:bad[0] + 3] if len(bad) else None)
