#!/usr/bin/env python3
"""tools/cw_taxonomy.sh run: cfg5 through the instrumented library (abtmp/cwprof/libdaachorse_amd.so), the counters per symbol."""
import ctypes as C, os, shutil, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
lib = R + "/daachorse_amd/lib/libdaachorse_amd.so"
shutil.copy(lib, "/tmp/_orig_cw.so")
shutil.copy(R + "/abtmp/cwprof/libdaachorse_amd.so", lib)
try:
    import torch, daachorse_amd as da
    from daachorse_amd import ScanMode, synth
    L = C.CDLL(lib)
    names = ["own_turns", "symbols", "done", "reports", "probe", "probe_hit", "follow", "by_row", "fell_at_root", "dead", "phase1", "ret_bytes", "runs"]
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for mode_name in ("leftmost", "find"):
        kind = da.MatchKind.LeftmostLongest if mode_name == "leftmost" else da.MatchKind.Standard
        mode = {"leftmost": ScanMode.LeftmostFind, "find": ScanMode.Find}[mode_name]
        pma = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(kind).build(synth.patterns_cfg5())
        pma.upload(0)
        n = synth.cfg5_haystack_bytes(mib << 20)
        hay = torch.empty(n, dtype=torch.uint8, device="cuda")
        synth.device_zipf_text(hay)
        buf = (C.c_ulonglong * 16)()
        L.daac_cw_prof(buf, 1)
        r = pma.scan_count(mode, hay)
        torch.cuda.synchronize()
        L.daac_cw_prof(buf, 0)
        v = dict(zip(names, list(buf)))
        print(f"cfg5 {mode_name}: {n} bytes, {r[0]} matches; all passes of the chain walkers (speculation + reconciliation), lane-turns summed: {v}")
        print("   per symbol taken: turns %.3f  probes %.3f  probe hits %.3f  failure links followed %.3f  settled by ROOT's row %.3f  dead %.3f  "
              "failed probe -> link next turn %.3f  reports %.3f ; haystack bytes per symbol taken %.3f" % tuple(
                  [v[k] / v["symbols"] for k in ("own_turns", "probe", "probe_hit", "follow", "by_row", "dead", "phase1", "reports")] + [n / v["symbols"]]))
finally:
    shutil.copy("/tmp/_orig_cw.so", lib)
