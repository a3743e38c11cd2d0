"""Host-side mirror of the crate's charwise API over the C ABI.

Same names, argument meaning and error behaviour as reference src/charwise.rs /
src/charwise/builder.rs:

    pma = CharwiseDoubleArrayAhoCorasick.new(["全世界", "世界", "に"])
    [(m.start(), m.end(), m.value()) for m in pma.find_iter("全世界中に")]   # byte offsets, as in the crate

Patterns and haystacks are `str` (UTF-8 encoded at the boundary) or already-encoded bytes / device
tensors holding well-formed UTF-8.  Every scan runs on the MI355X; nothing here computes matches.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import DaachorseError
from .bytewise import DoubleArrayAhoCorasick, MatchKind, _as_bytes


class CharwiseDoubleArrayAhoCorasick(DoubleArrayAhoCorasick):
    """CharwiseDoubleArrayAhoCorasick<u32> (reference src/charwise.rs:59-65).  The iterators
    (find_iter, find_overlapping_iter, find_overlapping_no_suffix_iter, leftmost_find_iter: charwise.rs:
    101-157, 160-221, 224-303, 306-400 of charwise/iter.rs), serialize(), info(), scan() and
    scan_count() are inherited: the handle behind them is polymorphic."""

    # ---- construction (charwise.rs:87-94, 139-146) ----------------------------------------------------
    @classmethod
    def new(cls, patterns):
        return CharwiseDoubleArrayAhoCorasickBuilder().build(patterns)

    @classmethod
    def with_values(cls, patvals):
        return CharwiseDoubleArrayAhoCorasickBuilder().build_with_values(patvals)

    # ---- deserialisation (charwise.rs:896-952) ----------------------------------------------------------
    @classmethod
    def deserialize(cls, source):
        """-> (pma, remaining bytes), as the reference returns (Self, &[u8])"""
        source = bytes(source)
        h, consumed = C.c_void_p(), C.c_size_t()
        _ffi.check(_ffi.lib().daac_charwise_from_serialized(source, len(source), C.byref(h), C.byref(consumed)))
        return cls(h), source[consumed.value:]

    @classmethod
    def from_parts(cls, *a, **k):
        raise NotImplementedError("charwise automata are handed over as serialize() blobs")

    def alphabet_size(self):
        return self.info().alphabet_size


class CharwiseDoubleArrayAhoCorasickBuilder:
    """reference src/charwise/builder.rs:20-239"""

    def __init__(self):
        self._kind = MatchKind.Standard
        self._num_free_blocks = 16

    def match_kind(self, kind):
        self._kind = MatchKind(kind)
        return self

    def num_free_blocks(self, n):
        assert n >= 1  # builder.rs:131
        self._num_free_blocks = int(n)
        return self

    def build(self, patterns):
        return self._build(list(patterns), None)

    def build_with_values(self, patvals):
        patvals = list(patvals)
        return self._build([p for p, _ in patvals], [v for _, v in patvals])

    def _build(self, patterns, values):
        pats = [_as_bytes(p) for p in patterns]
        offs = np.zeros(len(pats) + 1, dtype=np.uint64)
        if pats:
            offs[1:] = np.cumsum([len(p) for p in pats], dtype=np.uint64)
        blob = np.frombuffer(b"".join(pats) or b"\0", dtype=np.uint8)
        vals = None
        if values is not None:
            if any(not (0 <= int(v) <= 0xFFFFFFFF) for v in values):
                raise DaachorseError(3, "value does not fit u32")
            vals = np.ascontiguousarray(values, dtype=np.uint32)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().daac_charwise_build(blob.ctypes.data, offs.ctypes.data,
                                                  vals.ctypes.data if vals is not None and vals.size else None,
                                                  len(pats), int(self._kind), self._num_free_blocks, C.byref(h)))
        return CharwiseDoubleArrayAhoCorasick(h)
