"""The restart iterators' tuple LIST from the selection kernels (find3_kernels.hip / left3_kernels.hip in their emitting form): find_iter and
leftmost_find_iter (both leftmost kinds) against the oracle's lists — both device formats, the lazy iterator over small windows (every
window restarts where the one before ended), windows inside one device call, and the list of 1 GiB of cfg3 against its own count + checksum."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth


def _same(got, want):
    return len(got) == len(want) and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and \
        np.array_equal(got["value"], want["value"])


def _same16(got16, want):
    return len(got16) == len(want) and np.array_equal(got16["end"], want["end"]) and np.array_equal(got16["value"], want["value"]) and \
        np.array_equal(got16["length"].astype(np.uint64), want["end"] - want["start"])


def _cases():
    pats3 = synth.patterns_cfg3(30000)
    with1 = synth.patterns_cfg3(5000) + [b"a", b"e", b"q"]
    deepish = [b"abcd", b"bcdefg", b"cdefghijklmnopqrs", b"defg", b"ghij", b"xy", b"yz", b"zab", b"nopqrstuvwxyzabcdef", b"ab", b"abc"]
    return [(pats3, synth.uniform_haystack((3 << 20) + 7, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
            (pats3, synth.wordsoup_haystack(1 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
            (with1, synth.uniform_haystack(1 << 20, 5, synth.ALPHA_LOWER_SPACE)),
            (with1, synth.wordsoup_haystack(1 << 19, 6, with1, 20)),
            (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" + b"-" * 300) * 3000, dtype=np.uint8)),
            (deepish, synth.uniform_haystack(1 << 20, 9, b"abcdefghijklmnopqrstuvwxyz"))]


def test_select_tuple_lists_against_the_oracle():
    import torch
    served = 0
    for kind, mode, api in ((orc.STANDARD, ScanMode.Find, "find_iter"), (orc.LEFTMOST_LONGEST, ScanMode.LeftmostFind, "leftmost_find_iter"),
                            (orc.LEFTMOST_FIRST, ScanMode.LeftmostFind, "leftmost_find_iter")):
        for pats, hay in _cases():
            o = orc.OraclePma.build(pats, kind=kind)
            p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
            p.set_option("find3", 2).set_option("left3", 2)   # (the handle's own settings: daac_pma_set_option)
            want = getattr(o, api)(hay)
            dev = torch.from_numpy(np.concatenate([np.zeros(3, dtype=np.uint8), hay])).cuda()[3:]
            for win in (1 << 30, 70000):   # one window / windows inside the call (a haystack "beyond one window")
                p.set_option("find3_window", win)
                dm = p.scan_device(mode, dev, fmt16=True)
                served += da.last_engine() == int(Engine.Gram)
                assert _same16(dm.to_numpy(), want), (api, kind, len(pats), len(hay), win)
                dm.free()
                dm = p.scan_device(mode, dev)
                assert _same(dm.to_numpy(), want), (api, kind, len(pats), len(hay), win)
                dm.free()
            p.set_option("find3_window")
            # the eager host list and the lazy iterator over small windows
            assert _same(p.scan(mode, hay), want), (api, kind, "daac_scan")
            p.set_option("iter_window", 40000)
            it = getattr(p, api)(hay)
            runs = []
            while True:
                run = it.next_batch()
                if run is None:
                    break
                runs.append(run.copy())
            it.close()
            assert len(runs) > 3 and _same16(np.concatenate(runs), want), (api, kind, "iterator")
            # the walkers' list is the same list
            p.set_option("select_emit", 0)
            dm = p.scan_device(mode, dev, fmt16=True)
            assert da.last_engine() != int(Engine.Gram) and _same16(dm.to_numpy(), want)
            dm.free()
    assert served >= 30, served


def test_select_tuple_list_one_gib_of_cfg3():
    """1 GiB (+ a little: two windows for the leftmost kind) of cfg3's random text: the list left in HBM has the count + checksum the counting
    request reports, its ends ascend, and no two of its matches overlap."""
    import torch
    from test_gpu_configs import _DeviceWords, _checksum_of_device_tuples16
    pats = synth.patterns_cfg3()
    dev = torch.empty((1 << 30) + 12345, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    for kind, mode in ((da.MatchKind.Standard, ScanMode.Find), (da.MatchKind.LeftmostLongest, ScanMode.LeftmostFind)):
        p = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).build(pats)
        want = p.scan_count(mode, dev)
        dm = p.scan_device(mode, dev, fmt16=True)
        assert da.last_engine() == int(Engine.Gram)
        assert _checksum_of_device_tuples16(dm) == want, kind
        w = torch.as_tensor(_DeviceWords(dm.ptr, 2 * dm.count), device="cuda").view(dm.count, 2)
        ends = w[:, 0]
        starts = ends - (w[:, 1] & 0xFFFFFFFF)
        assert bool((starts[1:] >= ends[:-1]).all()), kind   # in order, and a match begins where the one before has ended or later
        assert int(ends[-1].item()) <= dev.numel()
        dm.free()
        del w, ends, starts
        torch.cuda.empty_cache()
