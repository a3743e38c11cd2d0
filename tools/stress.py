#!/usr/bin/env python3
"""Randomised parity soak on one MI355X: many small automata / haystacks / segment sizes, all four iterators of
both automaton flavours against the CPU oracle (test infrastructure).  usage: python tools/stress.py [seconds] [seed] [gram]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import daachorse_amd as da
from daachorse_amd import ScanMode
from oracle import oracle as orc

APIS = {0: [("find_overlapping_iter", ScanMode.FindOverlapping), ("find_overlapping_no_suffix_iter", ScanMode.FindOverlappingNoSuffix),
            ("find_iter", ScanMode.Find)], 1: [("leftmost_find_iter", ScanMode.LeftmostFind)], 2: [("leftmost_find_iter", ScanMode.LeftmostFind)]}
ALPHAS = [list("ab"), list("abc"), list("abcde"), [chr(c) for c in range(0x3041, 0x3046)], list("aé世") + ["\U0001F600"]]


def sev(m):
    return [(int(x["start"]), int(x["end"]), int(x["value"])) for x in m]




def iter_soak(budget, seed, max_cases=None):
    """all iterators and steppers of both automaton flavours against the oracle, random everything.
    `max_cases` set: exactly that many automata whatever the box's speed (the GPU suite); else `budget` seconds."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    cases = checks = 0
    while (cases < max_cases) if max_cases is not None else (time.time() - t0 < budget):
        A = ALPHAS[int(rng.integers(0, len(ALPHAS)))]
        multibyte = any(len(c.encode()) > 1 for c in A)
        npat = int(rng.integers(1, 40))
        maxlen = int(rng.choice([3, 7, 7, 15, 40]))
        pats = ["".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(1, maxlen + 1)))) for _ in range(npat)]
        if rng.random() < 0.3:  # families with common prefixes / suffixes
            pats += [pats[0] + q for q in pats[:5]] + [q + pats[-1] for q in pats[:5]]
        if rng.random() < 0.15 and not (multibyte and len({len(c.encode()) for c in A}) > 1):
            pats.insert(int(rng.integers(0, len(pats) + 1)), "")
        text = "".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(0, 6000))))
        if rng.random() < 0.2:
            text = (pats[0] + pats[-1]) * int(rng.integers(1, 300))  # periodic: chains that never fall in step
        elif rng.random() < 0.4:  # text made of patterns and pieces of patterns: deep states, long failure chains
            parts = []
            for _ in range(int(rng.integers(1, 800))):
                w = pats[int(rng.integers(0, len(pats)))]
                parts.append(w[:int(rng.integers(0, len(w) + 1))] if rng.random() < 0.5 else w)
            text = "".join(parts)
        if rng.random() < 0.04:  # now and then something bigger: thousands of patterns, megabytes of text
            big = [chr(c) for c in (range(0x61, 0x61 + 12) if not multibyte else range(0x3041, 0x3041 + 40))]
            pats = list({"".join(big[i] for i in rng.integers(0, len(big), size=int(rng.integers(2, 9)))) for _ in range(3000)})
            words = [pats[i] for i in rng.integers(0, len(pats), size=200_000)]
            text = ("" if rng.random() < 0.5 else " ").join(words)[:int(rng.integers(100_000, 1_500_000))]
            A = big
        charwise = multibyte or rng.random() < 0.4
        kind = int(rng.integers(0, 3))
        da.set_option("seg_bytes", int(rng.choice([0, 16, 32, 48, 256, 1024])))
        da.set_option("restart_chain", int(rng.random() < 0.8))
        da.set_option("chain_rounds", int(rng.choice([1, 2, 24])))
        da.set_option("iter_window", int(rng.choice([4096, 64 << 20])))
        if charwise:
            o = orc.OracleCharwisePma.build(pats, kind=kind)
            p, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o.serialize())
        else:
            o = orc.OraclePma.build(pats, kind=kind)
            p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        cases += 1
        for api, mode in APIS[kind]:
            try:
                want = getattr(o, api)(text)
            except orc.OracleError as e:
                assert e.code == 6
                try:
                    p.scan(mode, text)
                    raise SystemExit(f"expected Unsupported: {pats!r} {text[:80]!r} {api}")
                except da.DaachorseError as e2:
                    assert e2.code == 6
                continue
            got = p.scan(mode, text)
            ctx = (charwise, kind, api, pats, text[:200], len(text))
            assert sev(got) == sev(want), ctx
            assert p.scan_count(mode, text) == (len(want), orc.matches_checksum(want)), ctx
            if len(text) < 3000:
                assert [(m.start(), m.end(), m.value()) for m in getattr(p, api)(text)] == sev(want), ctx
            if kind == 0 and rng.random() < 0.5:  # the same through a chunk-fed stepper, random cuts (also inside characters)
                raw = text.encode()
                cuts = sorted(int(x) for x in rng.integers(0, len(raw) + 1, size=int(rng.integers(0, 6))))
                st = getattr(p, api.replace("_iter", "_stepper"))()
                fed, prev = [], 0
                for c in cuts + [len(raw)]:
                    fed += sev(st.feed(raw[prev:c]))
                    prev = c
                assert fed == sev(want), ("stepper", cuts) + ctx
            checks += 1
    if cases:
        print(f"stress ok: {cases} automata, {checks} iterator checks in {time.time() - t0:.0f} s (seed {seed})")


    for k, v in (("seg_bytes", 0), ("restart_chain", 1), ("chain_rounds", 24), ("iter_window", 64 << 20)):
        da.set_option(k, v)


def gram_soak(seconds, seed, max_cases=None):
    """count + checksum of the GRAM engine (and TIERED / DARRAY) against the oracle on random dictionaries.
    `max_cases` set: exactly that many automata whatever the box's speed (the GPU suite); else `seconds`."""
    import torch
    from daachorse_amd import Engine
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_auto = n_gram = 0
    while (n_auto < max_cases) if max_cases is not None else (time.time() - t0 < seconds):
        nsym = int(rng.integers(2, 27))
        syms = rng.choice(np.arange(97, 123), size=nsym, replace=False).astype(np.uint8)
        npat = int(rng.choice([1, 5, 50, 500, 5000]))
        lo, hi = int(rng.integers(1, 5)), int(rng.integers(5, 14))
        pats = [bytes(syms[rng.integers(0, nsym, size=int(rng.integers(lo, hi + 1)))]) for _ in range(npat)]
        noise = rng.choice(np.concatenate([syms, np.frombuffer(b" .,\n", dtype=np.uint8)]), size=int(rng.integers(1000, 3_000_000)))
        hay = noise.astype(np.uint8)
        if rng.random() < 0.5:  # text made of the patterns themselves (deep trie walks, many walkers)
            idx = rng.integers(0, npat, size=max(1, len(hay) // max(1, (lo + hi) // 2 + 1)))
            sep = b" " if rng.random() < 0.5 else b""
            hay = np.frombuffer(sep.join(pats[i] for i in idx.tolist())[:len(hay)] or b"x", dtype=np.uint8).copy()
        for _ in range(int(rng.integers(0, 200))):  # plant some patterns
            w = np.frombuffer(pats[int(rng.integers(0, npat))], dtype=np.uint8)
            at = int(rng.integers(0, max(1, len(hay) - len(w))))
            hay[at:at + len(w)] = w[:len(hay) - at]
        o = orc.OraclePma.build(pats)
        da_budget = int(rng.choice([158 * 1024, 40 * 1024, 9216]))
        da.set_option("gram_lds_budget", da_budget)
        opts = {"gram_region": int(rng.choice([2048, 16384, 65536])), "gram_slab": int(rng.choice([0, 4096, 20000])),
                "threads": int(rng.choice([1024, 1024, 768, 512, 256, 64])), "blocks_per_cu": int(rng.choice([0, 0, 1, 2])),
                "gram_ppl": int(rng.choice([0, 16])), "gram_dense": int(rng.choice([-1, 0, 1])), "seg_bytes": int(rng.choice([0, 0, 64, 4096])),
                "gram_version": int(rng.choice([0, 0, 1, 2])), "gram2_dpp": int(rng.choice([1, 1, 0]))}
        for k, v in opts.items():
            da.set_option(k, v)
        p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        dev = torch.from_numpy(hay).cuda()[int(rng.integers(0, 16)):]
        want = o.overlapping_count(dev.cpu().numpy(), threads=8)
        n_auto += 1
        for eng in (Engine.Auto, Engine.Tiered, Engine.DArray, Engine.Gram):
            try:
                got = p.scan_count(ScanMode.FindOverlapping, dev, engine=eng)
            except da.DaachorseError as e:
                assert e.code == 6 and eng in (Engine.Gram, Engine.Tiered), (eng, str(e))
                continue
            if got != want:
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez(f"gpurun_out/gram_fail_{seed}_{n_auto}.npz", hay=dev.cpu().numpy(), blob=np.frombuffer(o.serialize(), dtype=np.uint8),
                         budget=np.array([da_budget]), eng=np.array([int(eng)]))
            assert got == want, (eng, nsym, npat, lo, hi, len(hay), got, want, opts, da_budget)
            assert p.count(ScanMode.FindOverlapping, dev, engine=eng) == want[0], ("count", eng, nsym, npat, lo, hi, len(hay), opts, da_budget)
            n_gram += eng == Engine.Gram
        begin = int(rng.integers(1, len(dev)))
        head = p.scan_count(ScanMode.FindOverlapping, dev[:begin])
        tail = p.scan_count(ScanMode.FindOverlapping, dev, begin=begin)
        s = lambda c: ((c >> 32) & 0xFFFFFFFF, c & 0xFFFFFFFF)
        tot = (head[0] + tail[0], (((s(head[1])[0] + s(tail[1])[0]) & 0xFFFFFFFF) << 32) | ((s(head[1])[1] + s(tail[1])[1]) & 0xFFFFFFFF))
        assert tot == want, ("shards", begin, tot, want)
    for k, v in (("gram_lds_budget", 158 * 1024), ("gram_region", 0), ("gram_slab", 4096), ("threads", 1024), ("blocks_per_cu", 0),
                 ("gram_ppl", 0), ("gram_dense", -1), ("seg_bytes", 0), ("gram_version", 0), ("gram2_dpp", 1)):
        da.set_option(k, v)
    print(f"gram soak ok: {n_auto} automata ({n_gram} on the GRAM engine) in {time.time() - t0:.0f} s (seed {seed})")


def engines_soak(seconds, seed, max_cases=None):
    """The engines of round 3 against the oracle on random dictionaries over random ALPHABETS (2 .. 256 byte values): `.count()` on the gram4
    kernel in every body / launch shape, `.count()` and count + checksum on PFX, and the tuple list of the emitter in both device formats —
    with duplicate patterns, one-byte patterns and patterns of up to 60 bytes in the mix."""
    import torch
    from daachorse_amd import Engine
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_auto = n_g3 = n_pfx = n_emit = n_pfx_emit = 0
    da.set_option("pfx", 2)
    while (n_auto < max_cases) if max_cases is not None else (time.time() - t0 < seconds):
        nsym = int(rng.choice([2, 5, 12, 26, 29, 31, 60, 256]))
        syms = rng.choice(np.arange(256), size=nsym, replace=False).astype(np.uint8)
        if nsym < 256 and rng.random() < 0.4:   # one contiguous byte range (what gram4's arithmetic class map takes), also at either end of the byte values
            first = int(rng.choice([0, 256 - nsym, int(rng.integers(0, 257 - nsym))]))
            syms = np.arange(first, first + nsym).astype(np.uint8)
        npat = int(rng.choice([3, 40, 600, 6000]))
        lo = int(rng.integers(1, 5))
        hi = int(rng.choice([lo + 1, 8, 14, 24, 60]))
        pats = [bytes(syms[rng.integers(0, nsym, size=int(rng.integers(lo, hi + 1)))]) for _ in range(npat)]
        if rng.random() < 0.5:
            pats += [pats[int(i)] for i in rng.integers(0, npat, size=max(1, npat // 10))]  # duplicates: every copy is a match of its own
        n = int(rng.integers(1000, 1_500_000))
        if rng.random() < 0.5:
            hay = syms[rng.integers(0, nsym, size=n)]
            if rng.random() < 0.5:   # bytes of no pattern in between (just below / above a range, anything)
                noise = rng.random(n) < 0.1
                hay = np.where(noise, rng.choice(np.array([int(syms.min()) - 1, int(syms.max()) + 1, 0, 255, int(rng.integers(0, 256))]) % 256, size=n), hay).astype(np.uint8)
        else:  # text made of the patterns themselves
            hay = np.frombuffer(b"".join(pats[int(i)] for i in rng.integers(0, len(pats), size=n // max(1, (lo + hi) // 2) + 1))[:n], dtype=np.uint8).copy()
        o = orc.OraclePma.build(pats)
        da.set_option("gram_lds_budget", int(rng.choice([158 * 1024, 40 * 1024])))
        opts = {"gram_region": int(rng.choice([0, 2048, 65536])), "gram_ppl": int(rng.choice([0, 16, 32])), "gram3_tail": int(rng.choice([-1, 0, 1])),
                "gram_version": int(rng.choice([0, 4])), "gram4_arith": int(rng.choice([1, 1, 0])), "gram2_rfull": int(rng.choice([0, 1])), "threads": int(rng.choice([1024, 512])),
                "gram4_filter": int(rng.choice([1, 1, 0]))}
        for k, v in opts.items():
            da.set_option(k, v)
        p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        info = p.upload().info()
        dev = torch.from_numpy(hay).cuda()[int(rng.integers(0, 16)):]
        host = dev.cpu().numpy()
        want = o.overlapping_count(host, threads=8)
        n_auto += 1
        ctx = (nsym, npat, lo, hi, len(host), opts)
        which = "count"
        try:
            assert p.count(ScanMode.FindOverlapping, dev) == want[0], ("auto count", ctx, p.explain())
            which = "count + checksum"
            assert p.scan_count(ScanMode.FindOverlapping, dev) == want, ("auto checksum", ctx, p.explain())
        except da.DaachorseError as e:
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez(f"gpurun_out/engines_fail_{seed}_{n_auto}.npz", hay=host, blob=np.frombuffer(o.serialize(), dtype=np.uint8),
                     opts=np.array([f"{k}={v}" for k, v in opts.items()]))
            raise AssertionError(("auto", which, str(e), ctx, p.explain(), {f: getattr(info, f) for f in ("num_classes", "gram_available", "gram2_available", "gram2_exact", "pfx_available", "pfx_key_bytes", "pfx_lds_bytes")}))
        if info.gram2_available:
            assert p.count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want[0], ("gram count", ctx)
            n_g3 += 1
        if info.pfx_available:
            assert p.count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx) == want[0], ("pfx count", ctx)
            assert p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx) == want, ("pfx checksum", ctx)
            begin = int(rng.integers(1, len(host)))
            assert p.count(ScanMode.FindOverlapping, dev[:begin], engine=Engine.Pfx) + p.count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx, begin=begin) == want[0], ("pfx shards", begin, ctx)
            n_pfx += 1
        # the tuple list, both device formats, against the oracle's (a prefix keeps the oracle's list small)
        m = max(1, min(len(host), 200_000, int(4e6 / max(want[0] / len(host), 1e-9))))  # (dictionaries of one- and two-byte duplicates match thousands of times per byte)
        ref = o.find_overlapping_iter(host[:m])
        for fmt16 in (True, False):
            dm = p.scan_device(ScanMode.FindOverlapping, dev[:m], fmt16=fmt16)
            got = dm.to_numpy()
            dm.free()
            assert len(got) == len(ref), ("tuples", fmt16, len(got), len(ref), ctx)
            if fmt16:
                ok = np.array_equal(got["end"], ref["end"]) and np.array_equal(got["length"], ref["end"] - ref["start"]) and np.array_equal(got["value"], ref["value"])
            else:
                ok = all(np.array_equal(got[f], ref[f]) for f in ("start", "end", "value"))
            assert ok, ("tuples", fmt16, ctx)
        n_emit += da.last_engine() == int(Engine.Gram)
        # round 4: the same list from the PFX engine's emitter where it applies (no duplicate patterns; a tile's extras may make it say no),
        # and through the compact lazy iterator (8-byte tuples, windows of 4-64 KiB)
        if info.pfx_available:
            try:
                dm = p.scan_device(ScanMode.FindOverlapping, dev[:m], engine=Engine.Pfx, fmt16=True)
                got = dm.to_numpy()
                dm.free()
                assert len(got) == len(ref) and np.array_equal(got["end"], ref["end"]) and np.array_equal(got["length"], ref["end"] - ref["start"]) and \
                    np.array_equal(got["value"], ref["value"]), ("pfx tuples", ctx)
                n_pfx_emit += 1
            except da.DaachorseError as e:
                assert e.code == 6, (str(e), ctx)
        if max(len(w) for w in pats) < 2000:
            da.set_option("iter_window", int(rng.choice([4096, 20000, 65536])))
            it = p.find_overlapping_iter(host[:m], compact=True)
            e_, l_, v_ = [np.zeros(0, dtype=np.uint64)], [np.zeros(0, dtype=np.uint32)], [np.zeros(0, dtype=np.uint32)]
            while True:
                got = it.next_batch8()
                if got is None:
                    break
                run, base, eb = got
                e_.append((run["end_len"] & np.uint32((1 << eb) - 1)).astype(np.uint64) + np.uint64(base)); l_.append(run["end_len"] >> np.uint32(eb)); v_.append(run["value"].copy())
            it.close()
            da.set_option("iter_window", 64 << 20)
            assert np.array_equal(np.concatenate(e_), ref["end"]) and np.array_equal(np.concatenate(l_), (ref["end"] - ref["start"]).astype(np.uint32)) and \
                np.array_equal(np.concatenate(v_), ref["value"]), ("compact iterator", ctx)
    for k, v in (("gram_lds_budget", 158 * 1024), ("gram_region", 0), ("gram_ppl", 0), ("gram3_tail", -1), ("gram4_arith", 1), ("gram4_filter", 1), ("gram_version", 0), ("gram2_rfull", 1), ("threads", 1024),
                 ("pfx", 1)):
        da.set_option(k, v)
    print(f"engines soak ok: {n_auto} automata ({n_g3} with GRAM tables, {n_pfx} with PFX tables, {n_emit} tuple lists from the GRAM emitter, {n_pfx_emit} from PFX's) in {time.time() - t0:.0f} s (seed {seed})")


def select_soak(seconds, seed, max_cases=None):
    """find3 / left3 (the restart iterators' count + checksum as a selection over the emitter's detection) against the oracle's iterators on
    random dictionaries over small alphabets: patterns of 1 .. 19 bytes (sometimes longer: the engine declines), with and without one-byte
    patterns, texts of random symbols / of the patterns themselves / mixed, haystacks at odd addresses, windows of 8 KiB .. 1 GiB, restarts
    inside the haystack; Standard, LeftmostLongest and LeftmostFirst.  The engines are told to try whatever the text (option = 2): what
    they cannot settle they hand to the chain walkers, and the answer is the oracle's either way."""
    import torch
    from daachorse_amd import Engine
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_auto = n_find = n_left = n_decl = n_list = 0
    da.set_option("find3", 2); da.set_option("left3", 2)
    while (n_auto < max_cases) if max_cases is not None else (time.time() - t0 < seconds):
        nsym = int(rng.choice([2, 3, 5, 12, 26, 29]))
        syms = rng.choice(np.arange(256), size=nsym, replace=False).astype(np.uint8)
        npat = int(rng.choice([3, 40, 600, 6000, 30000]))
        lo = int(rng.integers(1, 5))
        hi = int(rng.choice([lo + 1, 4, 8, 14, 19, 19, 24]))
        hi = max(hi, lo)
        pats = list({bytes(syms[rng.integers(0, nsym, size=int(rng.integers(lo, hi + 1)))]) for _ in range(npat)})
        order = rng.permutation(len(pats))
        pats = [pats[int(i)] for i in order]
        n = int(rng.integers(1000, 1_500_000))
        kind_text = rng.random()
        if kind_text < 0.4:
            hay = syms[rng.integers(0, nsym, size=n)]
        elif kind_text < 0.7:
            hay = np.frombuffer(b"".join(pats[int(i)] for i in rng.integers(0, len(pats), size=n // max(1, (lo + hi) // 2) + 1))[:n], dtype=np.uint8).copy()
        else:   # words between stretches of noise (some of it no pattern byte at all)
            parts = []
            while sum(len(x) for x in parts) < n:
                parts.append(pats[int(rng.integers(0, len(pats)))])
                parts.append(bytes(syms[rng.integers(0, nsym, size=int(rng.integers(0, 40)))]) if rng.random() < 0.7 else b"\x00" * int(rng.integers(0, 3000)))
            hay = np.frombuffer(b"".join(parts)[:n], dtype=np.uint8).copy()
        win = int(rng.choice([8192, 50000, 1 << 20, 1 << 30]))
        da.set_option("find3_window", win)
        dev = torch.from_numpy(hay).cuda()[int(rng.integers(0, 16)):]
        host = dev.cpu().numpy()
        n_auto += 1
        for kind in (orc.STANDARD, orc.LEFTMOST_LONGEST, orc.LEFTMOST_FIRST):
            o = orc.OraclePma.build(pats, kind=kind)
            p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
            p.upload()
            mode = ScanMode.Find if kind == orc.STANDARD else ScanMode.LeftmostFind
            ms = o.find_iter(host) if kind == orc.STANDARD else o.leftmost_find_iter(host)
            want = (len(ms), orc.matches_checksum(ms))
            ctx = (kind, nsym, len(pats), lo, hi, len(host), win, round(kind_text, 2))
            got = p.scan_count(mode, dev)
            served = da.last_engine() == int(Engine.Gram)
            assert got == want, ("count + checksum", ctx, served)
            assert p.count(mode, dev) == want[0], ("count", ctx)
            if kind == orc.STANDARD:
                n_find += served
            else:
                n_left += served
            n_decl += not served
            # the list itself (the selection kernels' emitting form where they serve; the walkers' otherwise): device formats, then the lazy
            # iterator over small windows
            fmt16 = bool(rng.random() < 0.7)
            dm = p.scan_device(mode, dev, fmt16=fmt16)
            got = dm.to_numpy()
            dm.free()
            if fmt16:
                assert len(got) == len(ms) and np.array_equal(got["end"], ms["end"]) and np.array_equal(got["length"], ms["end"] - ms["start"]) and \
                    np.array_equal(got["value"], ms["value"]), ("list16", ctx)
            else:
                assert len(got) == len(ms) and np.array_equal(got["start"], ms["start"]) and np.array_equal(got["end"], ms["end"]) and \
                    np.array_equal(got["value"], ms["value"]), ("list24", ctx)
            n_list += da.last_engine() == int(Engine.Gram)
            if rng.random() < 0.5:
                da.set_option("iter_window", int(rng.choice([4096, 20000, 65536])))
                it = (p.find_iter if kind == orc.STANDARD else p.leftmost_find_iter)(host)
                runs = [np.zeros(0, dtype=da.bytewise.MATCH16_DTYPE)]
                while True:
                    run = it.next_batch()
                    if run is None:
                        break
                    runs.append(run.copy())
                it.close()
                da.set_option("iter_window", 64 << 20)
                g2 = np.concatenate(runs)
                assert len(g2) == len(ms) and np.array_equal(g2["end"], ms["end"]) and np.array_equal(g2["length"], ms["end"] - ms["start"]) and \
                    np.array_equal(g2["value"], ms["value"]), ("iterator", ctx)
            if len(ms) > 2:   # a restart inside the haystack: at the end of a match the iterator returned
                b = int(ms[int(rng.integers(0, len(ms) - 1))]["end"])
                rest = ms[ms["end"] > b] if kind == orc.STANDARD else ms[ms["start"] >= b]
                assert p.scan_count(mode, dev, begin=b) == (len(rest), orc.matches_checksum(rest)), ("restart", b, ctx)
    da.set_option("find3", 1); da.set_option("left3", 1); da.set_option("find3_window", 1 << 30)
    print(f"select soak ok: {n_auto} dictionaries x 3 kinds ({n_find} find_iter scans served by find3, {n_left} leftmost scans by left3, {n_decl} handed to the chain walkers; {n_list} tuple lists from the selection kernels) "
          f"in {time.time() - t0:.0f} s (seed {seed})")


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "gram":
        gram_soak(budget, seed + 1000)
    elif len(sys.argv) > 3 and sys.argv[3] == "engines":
        engines_soak(budget, seed + 2000)
    elif len(sys.argv) > 3 and sys.argv[3] == "select":
        select_soak(budget, seed + 3000)
    else:
        iter_soak(budget, seed)
