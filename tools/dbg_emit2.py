import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth
from oracle import oracle as orc
pats = synth.patterns_cfg1()
hay = synth.uniform_haystack(9000, 3, synth.ALPHA_ABCD)
o = orc.OraclePma.build(pats)
want = o.find_overlapping_iter(hay)
for tiles, budget, shift in ((64, 158 * 1024, 0), (1, 158 * 1024, 3), (2, 24 * 1024, 9)):
    print("cfg", tiles, budget, shift, flush=True)
    da.set_option("emit_tiles", tiles)
    da.set_option("gram_lds_budget", budget)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    dev = torch.from_numpy(np.concatenate([np.zeros(shift, dtype=np.uint8), hay])).cuda()[shift:]
    got = p.scan(ScanMode.FindOverlapping, dev, engine=Engine.Gram)
    print("scan", len(got), len(want), flush=True)
    dm = p.scan_device(ScanMode.FindOverlapping, dev)
    print("dev", dm.count, da.last_engine(), flush=True)
    a = dm.to_numpy()
    print("eq", np.array_equal(a["start"], want["start"]) and np.array_equal(a["value"], want["value"]), flush=True)
    dm.free()
