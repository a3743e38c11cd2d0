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


# A 60-class variant of the cfg3 dictionary (dictionaries beyond 31 byte classes): the same words in lower,
# Capitalised or UPPER case (by a hash of the index), a quarter of them followed by a digit 0-7.  26 + 26 + 8 = 60 pattern bytes.
ALPHA_BYTES = bytes(range(256))                       # binary256: every byte value
SEEDS["bin_pat"], SEEDS["bin_hay"] = 0xDAAC0006, 0xDAAC0016


def patterns_binary256(n=100_000, seed=SEEDS["bin_pat"]):
    """n distinct binary patterns: length uniform 3..12, bytes uniform over all 256 values (a dictionary no byte-class engine serves)"""
    seen, out, j = set(), [], 0
    batch = 1 << 15
    while len(out) < n:
        idx = np.arange(j, j + batch, dtype=np.uint64)
        j += batch
        z0 = zstream(seed, idx * np.uint64(3))
        lengths = (3 + (z0 % np.uint64(10))).astype(np.int64)
        sh = (np.arange(8, dtype=np.uint64) * np.uint64(8))[None, :]
        raw = np.concatenate([(zstream(seed, idx * np.uint64(3) + np.uint64(t))[:, None] >> sh) & np.uint64(0xFF) for t in (1, 2)],
                             axis=1).astype(np.uint8)
        for row, length in zip(raw, lengths):
            w = row[:length].tobytes()
            if w not in seen:
                seen.add(w)
                out.append(w)
                if len(out) == n:
                    break
    return out


ALPHA_WIDE = ALPHA_LOWER + bytes(range(ord("A"), ord("Z") + 1)) + b"01234567"
ALPHA_WIDE_SPACE = ALPHA_WIDE + b" "


def patterns_cfg3_wide(n=100_000, seed=SEEDS["cfg3_pat"]):
    base = patterns_cfg3(n, seed)
    z = zstream(seed ^ 0x5A5A, np.arange(len(base), dtype=np.uint64))
    out = []
    for w, zi in zip(base, z.tolist()):
        style = zi % 3
        v = w if style == 0 else (w[:1].upper() + w[1:] if style == 1 else w.upper())
        if (zi >> 8) % 4 == 0:
            v += bytes([ord("0") + ((zi >> 16) % 8)])
        out.append(v)
    return out


# cfg5: 50 000 distinct UTF-8 patterns of 2-8 scalars drawn with Zipf(1.0) ranks from a 6 000-symbol
# alphabet (hiragana, katakana, then CJK unified ideographs from U+4E00; all 3 bytes in UTF-8), no empty pattern;
# the haystack is i.i.d. scalars of the same distribution plus 10 % ASCII (1-byte) characters, cut at a character boundary.
CFG5_CODEPOINTS = np.array(list(range(0x3041, 0x3097)) + list(range(0x30A1, 0x30F7)) + list(range(0x4E00, 0x4E00 + 6000 - 172)),
                           dtype=np.uint32)
CFG5_WEIGHTS = ((1 << 20) // (np.arange(len(CFG5_CODEPOINTS), dtype=np.int64) + 1)).astype(np.uint32)  # Zipf(1.0): weight 1 / rank
CFG5_CUM = np.cumsum(CFG5_WEIGHTS.astype(np.uint64)).astype(np.uint32)
CFG5_SLOT = 48           # bytes per slot of the text stream: every multiple is a character boundary
CFG5_ASCII_256 = 26      # 26 / 256 = 10 % of the scalars are ASCII
CFG5_ASCII = (0x20, 95)  # printable ASCII 0x20 .. 0x7E
_CFG5_LENGTHS = [(2, 27), (3, 30), (4, 22), (5, 12), (6, 6), (7, 2), (8, 1)]  # scalars, percent
SEEDS["cfg5_hay"] = 0xDAAC0015


def patterns_cfg5(n=50_000, seed=SEEDS["cfg5_pat"], lengths=None):
    """n distinct UTF-8 patterns (bytes) of 2..8 three-byte scalars, Zipf(1.0) over CFG5_CODEPOINTS, value = index"""
    assert len(CFG5_CODEPOINTS) == 6000
    lengths = _CFG5_LENGTHS if lengths is None else lengths
    total = np.uint64(CFG5_CUM[-1])
    cum_len = np.cumsum([w for _, w in lengths])
    len_of = np.array([l for l, _ in lengths])
    enc = [chr(int(c)).encode("utf-8") for c in CFG5_CODEPOINTS]
    seen, out, j = set(), [], 0
    batch = 1 << 15
    while len(out) < n:
        idx = np.arange(j, j + batch, dtype=np.uint64)
        j += batch
        zl = zstream(seed, idx * np.uint64(5))
        lengths = len_of[np.searchsorted(cum_len, (zl % np.uint64(cum_len[-1])).astype(np.int64), side="right")]
        # 8 scalar draws per pattern, 32 bits each (4 stream words)
        sh = (np.arange(2, dtype=np.uint64) * np.uint64(32))[None, :]
        raw = np.concatenate([(zstream(seed, idx * np.uint64(5) + np.uint64(t))[:, None] >> sh) & np.uint64(0xFFFFFFFF)
                              for t in (1, 2, 3, 4)], axis=1)
        chars = np.searchsorted(CFG5_CUM, ((raw * total) >> np.uint64(32)).astype(np.uint32), side="right")
        for row, length in zip(chars, lengths):
            w = b"".join(enc[c] for c in row[:length])
            if w not in seen:
                seen.add(w)
                out.append(w)
                if len(out) == n:
                    break
    return out


# Look-alikes of the two WIDE dictionaries the crate publishes numbers for (figures/overlapping.txt:1-5, figures/memory.txt:4) — the real
# ones are not in the repository (and there is no network): same number of patterns, same kind of bytes, same kind of automaton.
#   unidic_like: 675 000 UTF-8 patterns of 1-8 three-byte scalars (single characters included), Zipf(1.0) over the cfg5 alphabet; scanned
#                bytewise the automaton has ~250 distinct bytes on its edges; text: zipf_text / device_zipf_text (cfg5's).
#   o200k_like:  200 000 byte-level tokens, ALL 256 one-byte patterns among them: words in four spellings ("word", " word", "Word", " Word",
#                "WORD"), numbers, punctuation runs, UTF-8 pieces (lead + continuation bytes, whole CJK characters and pairs), short random
#                ASCII; text: word soup of the same base words (wordsoup_haystack / device_wordsoup with o200k_soup_words()).
SEEDS["unidic_pat"], SEEDS["o200k_pat"], SEEDS["o200k_hay"] = 0xDAAC0007, 0xDAAC0008, 0xDAAC0018
_UNIDIC_LENGTHS = [(1, 6), (2, 26), (3, 28), (4, 20), (5, 11), (6, 6), (7, 2), (8, 1)]  # scalars, percent


def patterns_unidic_like(n=675_000, seed=SEEDS["unidic_pat"]):
    return patterns_cfg5(n, seed, _UNIDIC_LENGTHS)


def o200k_soup_words(n=60_000):
    return patterns_cfg3(n)


def patterns_o200k_like(n=200_000, seed=SEEDS["o200k_pat"]):
    out, seen = [], set()

    def add(w):
        if w and w not in seen and len(w) <= 24:
            seen.add(w)
            out.append(w)

    for b in range(256):
        add(bytes([b]))
    words = o200k_soup_words()
    z = zstream(seed, np.arange(len(words), dtype=np.uint64)).tolist()
    for w, zi in zip(words, z):          # every word plain and with its leading space, a third also capitalised / upper
        add(w)
        add(b" " + w)
        if zi % 3 == 0:
            add(w[:1].upper() + w[1:])
            add(b" " + w[:1].upper() + w[1:])
        if zi % 11 == 0:
            add(w.upper())
        if len(out) >= n * 3 // 4:
            break
    for i in range(1000):
        add(str(i).encode())
        add(b" " + str(i).encode())
    for c in b".,;:!?-_=*#/\\()[]{}<>\"'\n\t":
        for k in range(2, 9):
            add(bytes([c]) * k)
    for lead in range(0xC2, 0xE0):       # two-byte UTF-8 characters
        for cont in range(0x80, 0xC0, 2):
            add(bytes([lead, cont]))
    enc = [chr(int(c)).encode("utf-8") for c in CFG5_CODEPOINTS]
    for e in enc:                        # three-byte characters and their two-byte heads
        add(e)
        add(e[:2])
    total = np.uint64(CFG5_CUM[-1])
    j = 0
    while len(out) < n:                  # the rest: CJK pairs (Zipf) and short random printable ASCII, alternating
        zz = zstream(seed ^ 0x77, np.arange(j, j + 4096, dtype=np.uint64))
        j += 4096
        a = np.searchsorted(CFG5_CUM, (((zz & np.uint64(0xFFFFFFFF)) * total) >> np.uint64(32)).astype(np.uint32), side="right")
        b = np.searchsorted(CFG5_CUM, (((zz >> np.uint64(32)) * total) >> np.uint64(32)).astype(np.uint32), side="right")
        for k, (x, y, zi) in enumerate(zip(a.tolist(), b.tolist(), zz.tolist())):
            if k & 1:
                add(enc[x] + enc[y])
            else:
                ln = 2 + zi % 3
                add(bytes(0x20 + ((zi >> (8 * (t + 1))) & 0xFF) % 95 for t in range(ln)))
            if len(out) >= n:
                break
    return out[:n]


def zipf_text(n, seed=SEEDS["cfg5_hay"], offset=0, codepoints=CFG5_CODEPOINTS, cum=CFG5_CUM, ascii_256=CFG5_ASCII_256,
              ascii=CFG5_ASCII, slot=CFG5_SLOT):
    """bytes offset .. offset+n of the Zipf text stream (see include/daac_synth.h), numpy uint8"""
    s0, s1 = offset // slot, (offset + n + slot - 1) // slot
    z = zstream(seed, np.arange(s0, s1, dtype=np.uint64))
    ns = len(z)
    out = np.zeros((ns, slot + 3), dtype=np.uint8)
    pos = np.zeros(ns, dtype=np.int64)
    total = np.uint64(cum[-1])
    rows = np.arange(ns)
    k = 0
    while True:
        live = pos < slot
        if not live.any():
            break
        with np.errstate(over="ignore"):
            zz = mix64(z + np.uint64(k + 1))
        k += 1
        is_ascii = ((slot - pos) < 3) | ((zz & np.uint64(0xFF)) < np.uint64(ascii_256))
        a = (ascii[0] + ((((zz >> np.uint64(8)) & np.uint64(0xFFFFFF)) * np.uint64(ascii[1])) >> np.uint64(24))).astype(np.uint8)
        u = (((zz >> np.uint64(32)) * total) >> np.uint64(32)).astype(np.uint32)
        cp = codepoints[np.searchsorted(cum, u, side="right")].astype(np.uint32)
        b0 = np.where(is_ascii, a, (0xE0 | (cp >> 12)).astype(np.uint8))
        b1 = (0x80 | ((cp >> 6) & 0x3F)).astype(np.uint8)
        b2 = (0x80 | (cp & 0x3F)).astype(np.uint8)
        r = rows[live]
        out[r, pos[live]] = b0[live]
        wide = live & ~is_ascii
        r = rows[wide]
        out[r, pos[wide] + 1] = b1[wide]
        out[r, pos[wide] + 2] = b2[wide]
        pos = np.where(live, pos + np.where(is_ascii, 1, 3), pos)
    flat = np.ascontiguousarray(out[:, :slot]).reshape(-1)
    lo = offset - s0 * slot
    return np.ascontiguousarray(flat[lo:lo + n])


def cfg5_haystack_bytes(nominal=1 << 30):
    """the configured size cut down to a character boundary (a whole number of slots)"""
    return nominal - nominal % CFG5_SLOT


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


def device_zipf_text(tensor, seed=SEEDS["cfg5_hay"], offset=0, codepoints=CFG5_CODEPOINTS, cum=CFG5_CUM, ascii_256=CFG5_ASCII_256,
                     ascii=CFG5_ASCII, slot=CFG5_SLOT, stream=None):
    cps = np.ascontiguousarray(codepoints, dtype=np.uint32)
    cw = np.ascontiguousarray(cum, dtype=np.uint32)
    _ffi.check(_ffi.lib().daac_synth_zipf_text(tensor.data_ptr(), tensor.numel(), C.c_uint64(seed), cps.ctypes.data, cw.ctypes.data, len(cps),
                                               ascii_256, ascii[0], ascii[1], slot, C.c_uint64(offset), stream))
    return tensor
