"""Synthetic patterns and haystacks of BASELINE.json's configs.

Everything is a pure function of (seed, index): the numpy definitions here are the reference the
device generators (include/daac_synth.h, csrc/synth.hip) are tested against, and pattern sets are
always generated on the host (they are tiny).  Bench and tests use the same seeds.
"""
import ctypes as C

import numpy as np

from . import _ffi

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)

SEEDS = {
    "cfg1_hay": 0xDAAC0001,
    "cfg2_pat": 0xDAAC0002, "cfg2_hay": 0xDAAC0012, "cfg2_dense": 0xDAAC0022,
    "cfg3_pat": 0xDAAC0003, "cfg3_hay": 0xDAAC0013, "cfg3_dense": 0xDAAC0023,
    "cfg4_hay": 0xDAAC0014,
    "cfg5_pat": 0xDAAC0005, "cfg5_dense": 0xDAAC0025,
}

ALPHA_ABCD = b"abcd"
ALPHA_PRINTABLE = bytes(range(0x20, 0x7F))           # cfg2: "random-ASCII"
ALPHA_LOWER = bytes(range(ord("a"), ord("z") + 1))
ALPHA_LOWER_SPACE = ALPHA_LOWER + b" "               # cfg3 (i)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def zstream(seed, j):
    """z(j) = mix64(seed + (j + 1) * GOLDEN) for an array of indices j"""
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed) + (np.asarray(j, dtype=np.uint64) + np.uint64(1)) * GOLDEN)


def uniform_haystack(n, seed, alphabet, offset=0):
    """bytes offset .. offset+n of the uniform stream (numpy uint8)"""
    al = np.frombuffer(bytes(alphabet), dtype=np.uint8)
    g0, g1 = offset >> 3, (offset + n + 7) >> 3
    z = zstream(seed, np.arange(g0, g1, dtype=np.uint64))
    b = (z[:, None] >> (np.arange(8, dtype=np.uint64) * np.uint64(8))[None, :]) & np.uint64(0xFF)
    sym = al[((b * np.uint64(len(al))) >> np.uint64(8)).astype(np.int64)].reshape(-1)
    lo = offset - (g0 << 3)
    return np.ascontiguousarray(sym[lo:lo + n])


def wordsoup_haystack(n, seed, words, slot_bytes, pad=b" ", noise_256=77, alphabet=ALPHA_LOWER, offset=0):
    """bytes offset .. offset+n of the word-soup stream (see include/daac_synth.h)"""
    al = np.frombuffer(bytes(alphabet), dtype=np.uint8)
    s0, s1 = offset // slot_bytes, (offset + n + slot_bytes - 1) // slot_bytes
    z = zstream(seed, np.arange(s0, s1, dtype=np.uint64))
    out = np.full((s1 - s0, slot_bytes), pad[0], dtype=np.uint8)
    is_noise = (z & np.uint64(0xFF)) < np.uint64(noise_256)
    widx = ((z >> np.uint64(8)) % np.uint64(len(words))).astype(np.int64)
    maxw = max(len(w) for w in words)
    wmat = np.full((len(words), maxw), pad[0], dtype=np.uint8)
    wlen = np.zeros(len(words), dtype=np.int64)
    for i, w in enumerate(words):
        wmat[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        wlen[i] = len(w)
    assert maxw < slot_bytes
    rows = np.nonzero(~is_noise)[0]
    out[rows, :maxw] = wmat[widx[rows]]
    nrows = np.nonzero(is_noise)[0]
    if len(nrows):
        j = np.arange(slot_bytes - 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            zz = mix64(z[nrows][:, None] + (j >> np.uint64(3))[None, :] + np.uint64(1))
        r = (zz >> ((j & np.uint64(7)) * np.uint64(8))[None, :]) & np.uint64(0xFF)
        out[nrows, :slot_bytes - 1] = al[((r * np.uint64(len(al))) >> np.uint64(8)).astype(np.int64)]
    flat = out.reshape(-1)
    lo = offset - s0 * slot_bytes
    return np.ascontiguousarray(flat[lo:lo + n])


# ------------------------------------------------------------------------------------- pattern sets
def patterns_cfg1():
    return [b"bcd", b"ab", b"a"]


def patterns_cfg2(n=1000, seed=SEEDS["cfg2_pat"]):
    """n distinct patterns, length uniform 4..12, bytes uniform a-z"""
    seen, out, j = set(), [], 0
    while len(out) < n:
        z = zstream(seed, np.arange(j, j + 4, dtype=np.uint64))
        j += 4
        length = 4 + int(z[0] % np.uint64(9))
        raw = np.concatenate([(z[1:, None] >> (np.arange(8, dtype=np.uint64) * np.uint64(8))[None, :]) & np.uint64(0xFF)]).reshape(-1)
        w = bytes((ord("a") + (raw[:length] * np.uint64(26) >> np.uint64(8))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            out.append(w)
    return out


# English unigram frequencies (per mille, a..z) and the word-length distribution of SURVEY §8d
_UNIGRAM = [82, 15, 28, 43, 127, 22, 20, 61, 70, 2, 8, 40, 24, 67, 75, 19, 1, 60, 63, 91, 28, 10, 24, 2, 20, 1]
_LENGTHS = [(2, 1), (3, 3), (4, 7), (5, 11), (6, 14), (7, 15), (8, 14), (9, 12), (10, 9), (11, 6), (12, 4),
            (13, 1), (14, 1), (15, 1), (16, 1)]


def patterns_cfg3(n=100_000, seed=SEEDS["cfg3_pat"]):
    """n distinct lowercase 'words' (words_100000-style): lengths from _LENGTHS (percent), letters
    i.i.d. from English unigram frequencies; de-duplicated in generation order; value = index."""
    cum_len = np.cumsum([w for _, w in _LENGTHS])
    len_of = np.array([l for l, _ in _LENGTHS])
    cum_let = np.cumsum(_UNIGRAM)
    seen, out, j = set(), [], 0
    batch = 1 << 16
    while len(out) < n:
        idx = np.arange(j, j + batch, dtype=np.uint64)
        j += batch
        zl = zstream(seed, idx * np.uint64(5))
        lengths = len_of[np.searchsorted(cum_len, (zl % np.uint64(cum_len[-1])).astype(np.int64), side="right")]
        # 16 letter draws per word, 16 bits each (4 stream words), scaled to the per-mille total
        sh = (np.arange(4, dtype=np.uint64) * np.uint64(16))[None, :]
        raw = np.concatenate([(zstream(seed, idx * np.uint64(5) + np.uint64(t))[:, None] >> sh) & np.uint64(0xFFFF)
                              for t in (1, 2, 3, 4)], axis=1)
        draw = (raw * np.uint64(cum_let[-1]) >> np.uint64(16)).astype(np.int64)
        letters = (ord("a") + np.searchsorted(cum_let, draw, side="right")).astype(np.uint8)
        for row, length in zip(letters, lengths):
            w = bytes(row[:length])
            if w not in seen:
                seen.add(w)
                out.append(w)
                if len(out) == n:
                    break
    return out


# cfg5: a CJK-like vocabulary.  Code points: hiragana, katakana, then CJK unified ideographs; every one is
# 3 bytes in UTF-8.  Characters are drawn with Zipf-like weights (kana and the first ideographs dominate).
_CFG5_CODEPOINTS = list(range(0x3041, 0x3097)) + list(range(0x30A1, 0x30F7)) + list(range(0x4E00, 0x4E00 + 2228))
_CFG5_LENGTHS = [(1, 2), (2, 25), (3, 30), (4, 22), (5, 12), (6, 6), (7, 2), (8, 1)]  # characters, percent


def patterns_cfg5(n=50_000, seed=SEEDS["cfg5_pat"]):
    """n distinct UTF-8 'words' (bytes), 1..8 characters of 3 bytes each, value = index"""
    cps = np.array(_CFG5_CODEPOINTS, dtype=np.int64)
    weights = (1_000_000 / (np.arange(len(cps)) + 12.0)).astype(np.int64)
    cum_chr = np.cumsum(weights)
    cum_len = np.cumsum([w for _, w in _CFG5_LENGTHS])
    len_of = np.array([l for l, _ in _CFG5_LENGTHS])
    enc = [chr(int(c)).encode("utf-8") for c in cps]
    seen, out, j = set(), [], 0
    batch = 1 << 15
    while len(out) < n:
        idx = np.arange(j, j + batch, dtype=np.uint64)
        j += batch
        zl = zstream(seed, idx * np.uint64(5))
        lengths = len_of[np.searchsorted(cum_len, (zl % np.uint64(cum_len[-1])).astype(np.int64), side="right")]
        # 8 character draws per word, 32 bits each (4 stream words)
        sh = (np.arange(2, dtype=np.uint64) * np.uint64(32))[None, :]
        raw = np.concatenate([(zstream(seed, idx * np.uint64(5) + np.uint64(t))[:, None] >> sh) & np.uint64(0xFFFFFFFF)
                              for t in (1, 2, 3, 4)], axis=1)
        draw = ((raw * np.uint64(cum_chr[-1])) >> np.uint64(32)).astype(np.int64)
        chars = np.searchsorted(cum_chr, draw, side="right")
        for row, length in zip(chars, lengths):
            w = b"".join(enc[c] for c in row[:length])
            if w not in seen:
                seen.add(w)
                out.append(w)
                if len(out) == n:
                    break
    return out


def cfg5_text_block(words, nbytes, seed=SEEDS["cfg5_dense"], word_share=0.8):
    """about nbytes of UTF-8 text (numpy uint8, whole characters): dictionary words (word_share of the draws) run
    together with stray characters of the same vocabulary, no separators — every byte belongs to a 3-byte character"""
    rng = np.random.default_rng(seed)
    cps = [chr(c).encode("utf-8") for c in _CFG5_CODEPOINTS]
    avg = word_share * float(np.mean([len(w) for w in words])) + (1 - word_share) * 3
    n_items = int(nbytes / avg) + 1
    pick = rng.integers(0, len(words), size=n_items)
    stray = rng.integers(0, len(cps), size=n_items)
    is_word = rng.random(n_items) < word_share
    text = b"".join(words[i] if w else cps[c] for i, c, w in zip(pick.tolist(), stray.tolist(), is_word.tolist()))
    text = text[:nbytes - nbytes % 3]
    return np.frombuffer(text, dtype=np.uint8)


# --------------------------------------------------------------------------------- device generators
def _alpha(alphabet):
    a = np.frombuffer(bytes(alphabet), dtype=np.uint8)
    return a, a.ctypes.data, len(a)


def device_uniform(tensor, seed, alphabet, offset=0, stream=None):
    """fill a torch CUDA uint8 tensor with bytes offset.. of the uniform stream"""
    a, ap, an = _alpha(alphabet)
    _ffi.check(_ffi.lib().daac_synth_uniform(tensor.data_ptr(), tensor.numel(), C.c_uint64(seed), ap, an, C.c_uint64(offset), stream))
    return tensor


def device_wordsoup(tensor, seed, words, slot_bytes, pad=b" ", noise_256=77, alphabet=ALPHA_LOWER, offset=0, stream=None):
    a, ap, an = _alpha(alphabet)
    offs = np.zeros(len(words) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(w) for w in words], dtype=np.uint64)
    blob = np.frombuffer(b"".join(words), dtype=np.uint8)
    _ffi.check(_ffi.lib().daac_synth_wordsoup(tensor.data_ptr(), tensor.numel(), C.c_uint64(seed), blob.ctypes.data, offs.ctypes.data,
                                              len(words), slot_bytes, pad[0], noise_256, ap, an, C.c_uint64(offset), stream))
    return tensor
