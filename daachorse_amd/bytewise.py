"""Host-side mirror of the crate's bytewise API over the C ABI.

Same names, argument meaning and error behaviour as reference src/bytewise.rs /
src/bytewise/builder.rs, so the parity tests read like the reference's own tests:

    pma = DoubleArrayAhoCorasick.new(["bcd", "ab", "a"])
    [(m.start(), m.end(), m.value()) for m in pma.find_overlapping_iter("abcd")]

Every scan runs on the MI355X through libdaachorse_amd.so; nothing here computes matches.
"""
import ctypes as C
import enum

import numpy as np

from . import _ffi
from ._ffi import DaachorseError

MATCH_DTYPE = np.dtype([("start", "<u8"), ("end", "<u8"), ("value", "<u4"), ("_pad", "<u4")])


class MatchKind(enum.IntEnum):
    """src/lib.rs:324-346"""
    Standard = 0
    LeftmostLongest = 1
    LeftmostFirst = 2


class ScanMode(enum.IntEnum):
    FindOverlapping = 0
    Find = 1
    LeftmostFind = 2
    FindOverlappingNoSuffix = 3


class Engine(enum.IntEnum):
    Auto = 0
    Tiered = 1
    DArray = 2
    Gram = 3
    Pfx = 4


class Match:
    """src/lib.rs:286-320"""
    __slots__ = ("_s", "_e", "_v")

    def __init__(self, start, end, value):
        self._s, self._e, self._v = int(start), int(end), int(value)

    def start(self):
        return self._s

    def end(self):
        return self._e

    def value(self):
        return self._v

    def __eq__(self, o):
        return isinstance(o, Match) and (self._s, self._e, self._v) == (o._s, o._e, o._v)

    def __repr__(self):
        return f"Match(start={self._s}, end={self._e}, value={self._v})"


def _as_bytes(x):
    if isinstance(x, str):
        return x.encode("utf-8")
    return bytes(x)


class _Haystack:
    """A haystack argument: str/bytes/numpy (host) or a torch CUDA uint8 tensor (device)."""

    def __init__(self, h):
        self.keep = h
        self.is_device = 0
        if hasattr(h, "data_ptr") and hasattr(h, "is_cuda"):  # torch tensor
            if h.dtype.itemsize != 1 or not h.is_contiguous():
                raise DaachorseError(1, "haystack tensor must be contiguous uint8")
            self.ptr, self.len, self.is_device = h.data_ptr(), h.numel(), int(h.is_cuda)
            if self.len == 0:
                self.ptr = None
            return
        if isinstance(h, np.ndarray):
            if h.dtype.itemsize != 1:  # a value cast would scan something else than the caller's bytes
                raise DaachorseError(1, "haystack array must have a 1-byte dtype (uint8)")
            a = np.ascontiguousarray(h).view(np.uint8)
        else:
            a = np.frombuffer(_as_bytes(h), dtype=np.uint8)
        self.keep = a
        self.ptr = a.ctypes.data if a.size else None
        self.len = a.size


class _MatchList:
    """Owns a daac_matches handle and exposes its (page-locked) tuples to numpy without a copy."""

    def __init__(self, handle, n):
        self._h = handle
        ptr = _ffi.lib().daac_matches_data(handle)
        self.__array_interface__ = {"data": (ptr, True), "shape": (n,), "typestr": "|V%d" % MATCH_DTYPE.itemsize,
                                    "descr": MATCH_DTYPE.descr, "version": 3}

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().daac_matches_free(self._h)
                self._h = None
        except Exception:
            pass


MATCH16_DTYPE = np.dtype([("end", "<u8"), ("length", "<u4"), ("value", "<u4")])  # daac_match16 = the crate's own Match fields
MATCH8_DTYPE = np.dtype([("value", "<u4"), ("end_len", "<u4")])  # daac_match8: end relative to the run's base | length << end_bits


class DeviceMatches:
    """`count` tuples at device address `ptr`, in the reference's order: daac_match (24 bytes, MATCH_DTYPE) or, from
    scan_device(fmt16=True), daac_match16 (MATCH16_DTYPE)"""

    def __init__(self, ptr, count, dtype=None):
        self.ptr, self.count, self.dtype = ptr, count, (MATCH_DTYPE if dtype is None else dtype)

    def to_numpy(self, first=0, n=None):
        if self.count and self.ptr is None:
            raise DaachorseError(1, "the device match list has been freed")
        n = self.count - first if n is None else n
        if not (0 <= first <= self.count and 0 <= n <= self.count - first):
            raise DaachorseError(1, f"to_numpy({first}, {n}) outside a list of {self.count} tuples")
        out = np.zeros(n, dtype=self.dtype)
        if n:
            _ffi.check(_ffi.lib().daac_device_to_host(out.ctypes.data, self.ptr + first * self.dtype.itemsize, n * self.dtype.itemsize))
        return out

    def free(self):
        if self.ptr:
            _ffi.lib().daac_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _LazyIter:
    """Iterator<Item = Match<u32>> over daac_iter_* (bytewise/iter.rs next())."""

    def __init__(self, pma, mode, haystack, engine, stream, compact=False):
        self._pma = pma
        self._h = _Haystack(haystack)
        self._it = C.c_void_p()
        self._compact = bool(compact)
        opener = _ffi.lib().daac_iter_open_compact if compact else _ffi.lib().daac_iter_open
        _ffi.check(opener(pma._h, int(mode), int(engine), self._h.ptr, self._h.len, self._h.is_device, stream, C.byref(self._it)))

    def __iter__(self):
        return self

    def __next__(self):
        m = _ffi.Match()
        r = _ffi.lib().daac_iter_next(self._it, C.byref(m))
        if r == 1:
            return Match(m.start, m.end, m.value)
        if r == 0:
            raise StopIteration
        _ffi.check(-r)

    def next_batch(self):
        """daac_iter_next_batch: the next run of matches as a structured numpy VIEW {end u64, length u32, value u32} of the iterator's
        own window buffer (valid until the next call), or None when the iterator is exhausted."""
        import numpy as np
        p, n = C.c_void_p(), C.c_size_t()
        r = _ffi.lib().daac_iter_next_batch(self._it, C.byref(p), C.byref(n))
        if r == 0:
            return None
        if r < 0:
            _ffi.check(-r)
        buf = (C.c_char * (n.value * 16)).from_address(p.value)
        return np.frombuffer(buf, dtype=MATCH16_DTYPE)

    def next_batch8(self):
        """daac_iter_next_batch8 (iterators opened with compact=True): the next run as (view {value u32, end_len u32}, end_base, end_bits) —
        end = end_base + (end_len & ((1 << end_bits) - 1)), length = end_len >> end_bits — or None when the iterator is exhausted."""
        import numpy as np
        p, n, base, eb = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint32()
        r = _ffi.lib().daac_iter_next_batch8(self._it, C.byref(p), C.byref(n), C.byref(base), C.byref(eb))
        if r == 0:
            return None
        if r < 0:
            _ffi.check(-r)
        buf = (C.c_char * (n.value * 8)).from_address(p.value)
        return np.frombuffer(buf, dtype=MATCH8_DTYPE), base.value, eb.value

    def close(self):
        if self._it:
            _ffi.lib().daac_iter_close(self._it)
            self._it = None

    def __del__(self):
        try:
            if self._it:
                _ffi.lib().daac_iter_close(self._it)
                self._it = None
        except Exception:
            pass


class _Stepper:
    """FindStepper / FindOverlappingStepper (bytewise/iter.rs:344-475, charwise/iter.rs:403-534) fed chunk by chunk:
    `feed(chunk)` returns the matches decided so far, positions counted from the first byte ever fed."""

    def __init__(self, pma, mode, engine, stream):
        self._pma = pma
        self._stream = stream
        self._s = C.c_void_p()
        _ffi.check(_ffi.lib().daac_stream_open(pma._h, int(mode), int(engine), stream, C.byref(self._s)))

    def feed(self, chunk):
        h = _Haystack(chunk)
        out = C.c_void_p()
        _ffi.check(_ffi.lib().daac_stream_feed(self._s, h.ptr, h.len, h.is_device, C.byref(out)))
        n = _ffi.lib().daac_matches_count(out)
        if n == 0:
            _ffi.lib().daac_matches_free(out)
            return np.zeros(0, dtype=MATCH_DTYPE)
        return np.asarray(_MatchList(out, n))

    def feed_compact(self, chunk):
        """daac_stream_feed_compact: the same feed with the chunk's matches as 8-byte tuples -> (view {value u32, end_len u32}, end_base, end_bits),
        end = end_base + (end_len & ((1 << end_bits) - 1)), length = end_len >> end_bits; the view is valid until the next feed"""
        h = _Haystack(chunk)
        p, n, base, eb = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint32()
        _ffi.check(_ffi.lib().daac_stream_feed_compact(self._s, h.ptr, h.len, h.is_device, C.byref(p), C.byref(n), C.byref(base), C.byref(eb)))
        if n.value == 0:
            return np.zeros(0, dtype=MATCH8_DTYPE), base.value, eb.value
        buf = (C.c_char * (n.value * 8)).from_address(p.value)
        return np.frombuffer(buf, dtype=MATCH8_DTYPE), base.value, eb.value

    @staticmethod
    def decode8(run, end_base, end_bits):
        """8-byte tuples -> MATCH_DTYPE {start, end, value}"""
        out = np.zeros(len(run), dtype=MATCH_DTYPE)
        el = run["end_len"].astype(np.uint64)
        out["end"] = np.uint64(end_base) + (el & np.uint64((1 << end_bits) - 1))
        out["start"] = out["end"] - (el >> np.uint64(end_bits))
        out["value"] = run["value"]
        return out

    def __del__(self):
        try:
            if self._s:
                _ffi.lib().daac_stream_close(self._s)
                self._s = None
        except Exception:
            pass


class DoubleArrayAhoCorasick:
    """DoubleArrayAhoCorasick<u32> (reference src/bytewise.rs:54-68)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().daac_pma_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- construction (bytewise.rs:103-110, 154-161) ------------------------------------------------
    @classmethod
    def new(cls, patterns):
        return DoubleArrayAhoCorasickBuilder().build(patterns)

    @classmethod
    def with_values(cls, patvals):
        return DoubleArrayAhoCorasickBuilder().build_with_values(patvals)

    # ---- (de)serialisation (bytewise.rs:801-820, 868-964) --------------------------------------------
    def serialize(self):
        buf, n = C.c_void_p(), C.c_size_t()
        _ffi.check(_ffi.lib().daac_pma_serialize(self._h, C.byref(buf), C.byref(n)))
        data = C.string_at(buf, n.value)
        _ffi.lib().daac_free(buf)
        return data

    @classmethod
    def deserialize(cls, source):
        """-> (pma, remaining bytes), as the reference returns (Self, &[u8])"""
        source = bytes(source)
        h, consumed = C.c_void_p(), C.c_size_t()
        _ffi.check(_ffi.lib().daac_bytewise_from_serialized(source, len(source), C.byref(h), C.byref(consumed)))
        return cls(h), source[consumed.value:]

    @classmethod
    def from_parts(cls, kind, num_states, outputs, states=None, leftmost_states=None, fails=None):
        def arr(x, cols):
            if x is None:
                return None, 0
            a = np.ascontiguousarray(x, dtype=np.uint32).reshape(-1, cols) if cols > 1 else np.ascontiguousarray(x, dtype=np.uint32)
            return a, len(a)
        st, n_st = arr(states, 3)
        ls, n_ls = arr(leftmost_states, 2)
        fl, n_fl = arr(fails, 1)
        if n_fl != n_ls:  # the C ABI copies n_lstates entries from `fails` (bytewise.rs:61-63: the two arrays are parallel)
            raise DaachorseError(1, "`fails` must have one entry per leftmost state")
        ou, n_ou = arr(outputs, 3)
        h = C.c_void_p()
        p = lambda a: a.ctypes.data if a is not None and a.size else None
        _ffi.check(_ffi.lib().daac_bytewise_from_parts(p(st), n_st, p(ls), p(fl), n_ls, p(ou), n_ou, int(kind), int(num_states),
                                                       C.byref(h)))
        return cls(h)

    # ---- introspection -----------------------------------------------------------------------------------
    def info(self):
        i = _ffi.Info()
        i.struct_size = C.sizeof(_ffi.Info)
        _ffi.check(_ffi.lib().daac_pma_info(self._h, C.byref(i)))
        return i

    def explain(self):
        """the engine plan as text: which engine / kernel family serves each kind of request, and why not the fastest one"""
        n = _ffi.lib().daac_pma_explain(self._h, None, 0)
        buf = C.create_string_buffer(n)
        _ffi.lib().daac_pma_explain(self._h, buf, n)
        return buf.value.decode()

    def match_kind(self):
        return MatchKind(self.info().match_kind)

    def num_states(self):
        return self.info().num_states

    def heap_bytes(self):
        return self.info().heap_bytes

    def upload(self, device=0):
        _ffi.check(_ffi.lib().daac_pma_upload(self._h, device))
        return self

    def set_option(self, name, value=None):
        """daac_pma_set_option: an option for THIS handle (overrides the process-wide daac_set_option value); value=None removes the override"""
        _ffi.check(_ffi.lib().daac_pma_set_option(self._h, name.encode(), 0 if value is None else int(value), 1 if value is None else 0))
        return self

    def trim(self):
        """daac_pma_trim: gives back the scratch the handle keeps between calls (tables stay)"""
        _ffi.check(_ffi.lib().daac_pma_trim(self._h))
        return self

    # ---- lazy iterators, crate names (bytewise.rs:190-203, 292-314, 410-428, 547-566) --------------------
    # (compact=True: daac_iter_open_compact — 8-byte tuples over PCIe, read with next_batch8(); iterating match by match works on either)
    def find_iter(self, haystack, engine=Engine.Auto, stream=None, compact=False):
        return _LazyIter(self, ScanMode.Find, haystack, engine, stream, compact)

    def find_overlapping_iter(self, haystack, engine=Engine.Auto, stream=None, compact=False):
        return _LazyIter(self, ScanMode.FindOverlapping, haystack, engine, stream, compact)

    def find_overlapping_no_suffix_iter(self, haystack, engine=Engine.Auto, stream=None, compact=False):
        return _LazyIter(self, ScanMode.FindOverlappingNoSuffix, haystack, engine, stream, compact)

    def leftmost_find_iter(self, haystack, engine=Engine.Auto, stream=None, compact=False):
        return _LazyIter(self, ScanMode.LeftmostFind, haystack, engine, stream, compact)

    # ---- steppers for haystacks that arrive in pieces (bytewise.rs:238-251, 353-375; iter.rs:344-475) ------------------
    def find_stepper(self, engine=Engine.Auto, stream=None):
        return _Stepper(self, ScanMode.Find, engine, stream)

    def find_overlapping_stepper(self, engine=Engine.Auto, stream=None):
        return _Stepper(self, ScanMode.FindOverlapping, engine, stream)

    def find_overlapping_no_suffix_stepper(self, engine=Engine.Auto, stream=None):
        return _Stepper(self, ScanMode.FindOverlappingNoSuffix, engine, stream)

    # ---- eager forms (`.collect()` / `.count()` on the iterators) ---------------------------------------------
    def scan(self, mode, haystack, engine=Engine.Auto, stream=None):
        """-> numpy structured array (start, end, value) in the reference's order"""
        h = _Haystack(haystack)
        out = C.c_void_p()
        _ffi.check(_ffi.lib().daac_scan(self._h, int(mode), int(engine), h.ptr, h.len, h.is_device, stream, C.byref(out)))
        n = _ffi.lib().daac_matches_count(out)
        if n == 0:
            _ffi.lib().daac_matches_free(out)
            return np.zeros(0, dtype=MATCH_DTYPE)
        return np.asarray(_MatchList(out, n))  # read-only view of the library's buffer, freed with the array

    def scan_device(self, mode, haystack, engine=Engine.Auto, stream=None, fmt16=False):
        """-> DeviceMatches: the match list left in device memory (daac_scan_device; fmt16: daac_scan_device16, 16-byte tuples)"""
        h = _Haystack(haystack)
        ptr, n = C.c_void_p(), C.c_uint64()
        fn = _ffi.lib().daac_scan_device16 if fmt16 else _ffi.lib().daac_scan_device
        _ffi.check(fn(self._h, int(mode), int(engine), h.ptr, h.len, h.is_device, stream, C.byref(ptr), C.byref(n)))
        return DeviceMatches(ptr.value, n.value, MATCH16_DTYPE if fmt16 else MATCH_DTYPE)

    def count(self, mode, haystack, engine=Engine.Auto, stream=None, result_dev=None, begin=0):
        """`.count()` of the iterator: the number of matches with end in (begin, len], no checksum; with `result_dev`
        (device pointer to 3 x u64, the count goes to [0]) the call is asynchronous and returns None."""
        h = _Haystack(haystack)
        if result_dev is not None:
            _ffi.check(_ffi.lib().daac_scan_count_only_range(self._h, int(mode), int(engine), h.ptr, h.len, begin, h.is_device, stream,
                                                             None, result_dev))
            return None
        cnt = C.c_uint64()
        _ffi.check(_ffi.lib().daac_scan_count_only_range(self._h, int(mode), int(engine), h.ptr, h.len, begin, h.is_device, stream,
                                                         C.byref(cnt), None))
        return cnt.value

    def scan_count(self, mode, haystack, engine=Engine.Auto, stream=None, result_dev=None, begin=0):
        """-> (count, checksum) of the matches with end in (begin, len]; with `result_dev` (device
        pointer to 3 x u64 = count, S1, S2) the call is asynchronous and returns None."""
        h = _Haystack(haystack)
        if result_dev is not None:
            _ffi.check(_ffi.lib().daac_scan_count_range(self._h, int(mode), int(engine), h.ptr, h.len, begin, h.is_device, stream,
                                                        None, None, result_dev))
            return None
        cnt, cs = C.c_uint64(), C.c_uint64()
        _ffi.check(_ffi.lib().daac_scan_count_range(self._h, int(mode), int(engine), h.ptr, h.len, begin, h.is_device, stream,
                                                    C.byref(cnt), C.byref(cs), None))
        return cnt.value, cs.value


def scan_count_multi(pma, mode, shards, engine=Engine.Auto, checksum=True):
    """daac_scan_count_multi: one haystack sharded across the GPUs of a node.  `shards` = [(device, buffer, halo, base)]: `buffer` holds
    `halo` bytes of the haystack in front of the shard and then the shard (a CUDA uint8 tensor on that device, or host bytes for all
    shards), `base` = haystack position of the shard's first byte.  -> (count, checksum) or count; equal to scan_count / count of the
    whole haystack."""
    hs = [_Haystack(b) for _, b, _, _ in shards]
    if len({h.is_device for h in hs}) > 1:
        raise DaachorseError(1, "shards must all be device tensors or all host buffers")
    arr = (_ffi.Shard * len(shards))()
    for i, ((dev, _, halo, base), h) in enumerate(zip(shards, hs)):
        if halo > h.len:
            raise DaachorseError(1, "halo longer than the shard's buffer")
        arr[i] = _ffi.Shard(int(dev), h.ptr, int(halo), h.len - int(halo), int(base))
    cnt, cs = C.c_uint64(), C.c_uint64()
    _ffi.check(_ffi.lib().daac_scan_count_multi(pma._h, int(mode), int(engine), arr, len(shards), hs[0].is_device if hs else 0,
                                                C.byref(cnt), C.byref(cs) if checksum else None))
    return (cnt.value, cs.value) if checksum else cnt.value


class DoubleArrayAhoCorasickBuilder:
    """reference src/bytewise/builder.rs:21-244"""

    def __init__(self):
        self._kind = MatchKind.Standard
        self._num_free_blocks = 16

    def match_kind(self, kind):
        self._kind = MatchKind(kind)
        return self

    def num_free_blocks(self, n):
        assert n >= 1  # builder.rs:113
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
        _ffi.check(_ffi.lib().daac_bytewise_build(blob.ctypes.data, offs.ctypes.data,
                                                  vals.ctypes.data if vals is not None and vals.size else None,
                                                  len(pats), int(self._kind), self._num_free_blocks, C.byref(h)))
        return DoubleArrayAhoCorasick(h)
