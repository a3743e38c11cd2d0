"""ctypes binding for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package `daachorse_amd` never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

STANDARD, LEFTMOST_LONGEST, LEFTMOST_FIRST = 0, 1, 2
KIND = {"Standard": 0, "LeftmostLongest": 1, "LeftmostFirst": 2}

MATCH_DTYPE = np.dtype([("start", "<u8"), ("end", "<u8"), ("value", "<u4"), ("_pad", "<u4")])


class OracleError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"oracle {what} failed with status {code}")
        self.code = code


def build_lib(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("daac_oracle.c", "daac_oracle_charwise.c", "daac_oracle.h", "oracle_internal.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def build_native_lib():
    """The same sources with -O3 -march=native, compiled on the machine that runs it (never shipped: a binary tuned to
    one host's CPU may not start on another).  Used by bench.py's cpu_baseline leg (DAAC_ORACLE_NATIVE=1)."""
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getuid()}.so")
    subprocess.check_call(["make", "-C", _HERE, "-B", "native", f"NATIVE_OUT={out}"], stdout=subprocess.DEVNULL)
    return out


class _Pma(C.Structure):
    _fields_ = [
        ("states", C.c_void_p), ("n_states", C.c_size_t),
        ("root_table", C.c_void_p), ("n_root", C.c_size_t),
        ("lstates", C.c_void_p), ("n_lstates", C.c_size_t),
        ("fails", C.c_void_p), ("n_fails", C.c_size_t),
        ("outputs", C.c_void_p), ("n_outputs", C.c_size_t),
        ("match_kind", C.c_uint8), ("num_states", C.c_uint32),
    ]


class _CPma(C.Structure):
    _fields_ = [
        ("states", C.c_void_p), ("n_states", C.c_size_t),
        ("table", C.c_void_p), ("n_table", C.c_size_t),
        ("alphabet_size", C.c_uint32),
        ("outputs", C.c_void_p), ("n_outputs", C.c_size_t),
        ("match_kind", C.c_uint8), ("num_states", C.c_uint32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_native_lib() if os.environ.get("DAAC_ORACLE_NATIVE") == "1" else build_lib())
        P = C.POINTER
        L.orc_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, P(P(_Pma))]
        L.orc_build.restype = C.c_int
        L.orc_free_pma.argtypes = [P(_Pma)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_heap_bytes.argtypes = [P(_Pma)]
        L.orc_heap_bytes.restype = C.c_size_t
        L.orc_max_pattern_len.argtypes = [P(_Pma)]
        L.orc_max_pattern_len.restype = C.c_uint32
        L.orc_serialize.argtypes = [P(_Pma), P(C.c_void_p), P(C.c_size_t)]
        L.orc_serialize.restype = C.c_int
        L.orc_deserialize.argtypes = [C.c_char_p, C.c_size_t, P(P(_Pma)), P(C.c_size_t)]
        L.orc_deserialize.restype = C.c_int
        for name in ("orc_find_iter", "orc_find_overlapping_iter", "orc_find_overlapping_no_suffix_iter",
                     "orc_leftmost_find_iter", "orc_find_stepper", "orc_find_overlapping_stepper"):
            f = getattr(L, name)
            f.argtypes = [P(_Pma), C.c_void_p, C.c_size_t, P(C.c_void_p), P(C.c_size_t)]
            f.restype = C.c_int
        L.orc_overlapping_count.argtypes = [P(_Pma), C.c_void_p, C.c_size_t, C.c_int, P(C.c_uint64), P(C.c_uint64)]
        L.orc_overlapping_count.restype = C.c_int
        L.orc_cbuild.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, P(P(_CPma))]
        L.orc_cbuild.restype = C.c_int
        L.orc_cfree_pma.argtypes = [P(_CPma)]
        L.orc_cheap_bytes.argtypes = [P(_CPma)]
        L.orc_cheap_bytes.restype = C.c_size_t
        L.orc_cserialize.argtypes = [P(_CPma), P(C.c_void_p), P(C.c_size_t)]
        L.orc_cserialize.restype = C.c_int
        L.orc_cdeserialize.argtypes = [C.c_char_p, C.c_size_t, P(P(_CPma)), P(C.c_size_t)]
        L.orc_cdeserialize.restype = C.c_int
        for name in ("orc_cfind_iter", "orc_cfind_overlapping_iter", "orc_cfind_overlapping_no_suffix_iter",
                     "orc_cleftmost_find_iter", "orc_cfind_stepper", "orc_cfind_overlapping_stepper"):
            f = getattr(L, name)
            f.argtypes = [P(_CPma), C.c_void_p, C.c_size_t, P(C.c_void_p), P(C.c_size_t)]
            f.restype = C.c_int
        L.orc_matches_checksum.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_matches_checksum.restype = C.c_uint64
        _lib = L
    return _lib


def _as_bytes(x):
    if isinstance(x, str):
        return x.encode("utf-8")
    return bytes(x)


def pack_patterns(patterns):
    """-> (blob uint8 array, offsets uint64 array[n+1])"""
    pats = [_as_bytes(p) for p in patterns]
    offs = np.zeros(len(pats) + 1, dtype=np.uint64)
    if pats:
        offs[1:] = np.cumsum([len(p) for p in pats], dtype=np.uint64)
    blob = np.frombuffer(b"".join(pats) or b"\0", dtype=np.uint8)
    return blob, offs


def _hay(haystack):
    if isinstance(haystack, np.ndarray):
        a = np.ascontiguousarray(haystack, dtype=np.uint8)
    else:
        a = np.frombuffer(_as_bytes(haystack) or b"", dtype=np.uint8)
    return a


class OraclePma:
    """Mirror of DoubleArrayAhoCorasick<u32> (src/bytewise.rs:54-68) backed by the C oracle."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                lib().orc_free_pma(self._h)
                self._h = None
        except Exception:
            pass

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def build(cls, patterns, values=None, kind=STANDARD, num_free_blocks=16):
        kind = KIND.get(kind, kind)
        blob, offs = pack_patterns(patterns)
        vals = None
        if values is not None:
            vals = np.ascontiguousarray(values, dtype=np.uint32)
        out = C.POINTER(_Pma)()
        rc = lib().orc_build(blob.ctypes.data, offs.ctypes.data, vals.ctypes.data if vals is not None else None,
                             len(patterns), kind, num_free_blocks, C.byref(out))
        if rc:
            raise OracleError(rc, "build")
        return cls(out)

    @classmethod
    def deserialize(cls, data):
        out = C.POINTER(_Pma)()
        consumed = C.c_size_t()
        rc = lib().orc_deserialize(bytes(data), len(data), C.byref(out), C.byref(consumed))
        if rc:
            raise OracleError(rc, "deserialize")
        p = cls(out)
        p.consumed = consumed.value
        return p

    def serialize(self):
        buf = C.c_void_p()
        n = C.c_size_t()
        rc = lib().orc_serialize(self._h, C.byref(buf), C.byref(n))
        if rc:
            raise OracleError(rc, "serialize")
        data = C.string_at(buf, n.value)
        lib().orc_free(buf)
        return data

    # -- introspection --------------------------------------------------------------------
    @property
    def match_kind(self):
        return self._h.contents.match_kind

    @property
    def num_states(self):
        return self._h.contents.num_states

    def heap_bytes(self):
        return lib().orc_heap_bytes(self._h)

    def max_pattern_len(self):
        return lib().orc_max_pattern_len(self._h)

    def _arr(self, field, nfield, dtype, cols):
        n = getattr(self._h.contents, nfield)
        ptr = getattr(self._h.contents, field)
        if n == 0:
            return np.zeros((0, cols), dtype=dtype)
        buf = (C.c_uint32 * (n * cols)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(n, cols).copy()

    def states(self):
        """(n,3) u32 [base, fail, opos_ch]"""
        return self._arr("states", "n_states", np.uint32, 3)

    def leftmost_states(self):
        return self._arr("lstates", "n_lstates", np.uint32, 2)

    def fails(self):
        return self._arr("fails", "n_fails", np.uint32, 1)[:, 0]

    def outputs(self):
        """(n,3) u32 [value, length, parent]"""
        return self._arr("outputs", "n_outputs", np.uint32, 3)

    # -- scans: return numpy structured arrays of (start, end, value) ------------------------
    def _scan(self, fname, haystack):
        a = _hay(haystack)
        out = C.c_void_p()
        n = C.c_size_t()
        rc = getattr(lib(), fname)(self._h, a.ctypes.data if a.size else None, a.size, C.byref(out), C.byref(n))
        if rc:
            raise OracleError(rc, fname)
        if n.value:
            buf = (C.c_char * (n.value * MATCH_DTYPE.itemsize)).from_address(out.value)
            res = np.frombuffer(buf, dtype=MATCH_DTYPE).copy()
        else:
            res = np.zeros(0, dtype=MATCH_DTYPE)
        lib().orc_free(out)
        return res

    def find_iter(self, h):
        return self._scan("orc_find_iter", h)

    def find_overlapping_iter(self, h):
        return self._scan("orc_find_overlapping_iter", h)

    def find_overlapping_no_suffix_iter(self, h):
        return self._scan("orc_find_overlapping_no_suffix_iter", h)

    def leftmost_find_iter(self, h):
        return self._scan("orc_leftmost_find_iter", h)

    def find_stepper(self, h):
        return self._scan("orc_find_stepper", h)

    def find_overlapping_stepper(self, h):
        return self._scan("orc_find_overlapping_stepper", h)

    def overlapping_count(self, haystack, threads=1):
        a = _hay(haystack)
        cnt = C.c_uint64()
        cs = C.c_uint64()
        rc = lib().orc_overlapping_count(self._h, a.ctypes.data if a.size else None, a.size, threads,
                                         C.byref(cnt), C.byref(cs))
        if rc:
            raise OracleError(rc, "overlapping_count")
        return cnt.value, cs.value


class OracleCharwisePma:
    """Mirror of CharwiseDoubleArrayAhoCorasick<u32> (src/charwise.rs:59-65) backed by the C oracle."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                lib().orc_cfree_pma(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def build(cls, patterns, values=None, kind=STANDARD, num_free_blocks=16):
        kind = KIND.get(kind, kind)
        blob, offs = pack_patterns(patterns)
        vals = np.ascontiguousarray(values, dtype=np.uint32) if values is not None else None
        out = C.POINTER(_CPma)()
        rc = lib().orc_cbuild(blob.ctypes.data, offs.ctypes.data, vals.ctypes.data if vals is not None else None,
                              len(patterns), kind, num_free_blocks, C.byref(out))
        if rc:
            raise OracleError(rc, "cbuild")
        return cls(out)

    @classmethod
    def deserialize(cls, data):
        out = C.POINTER(_CPma)()
        consumed = C.c_size_t()
        rc = lib().orc_cdeserialize(bytes(data), len(data), C.byref(out), C.byref(consumed))
        if rc:
            raise OracleError(rc, "cdeserialize")
        p = cls(out)
        p.consumed = consumed.value
        return p

    def serialize(self):
        buf, n = C.c_void_p(), C.c_size_t()
        rc = lib().orc_cserialize(self._h, C.byref(buf), C.byref(n))
        if rc:
            raise OracleError(rc, "cserialize")
        data = C.string_at(buf, n.value)
        lib().orc_free(buf)
        return data

    @property
    def match_kind(self):
        return self._h.contents.match_kind

    @property
    def num_states(self):
        return self._h.contents.num_states

    @property
    def alphabet_size(self):
        return self._h.contents.alphabet_size

    def num_elements(self):
        return self._h.contents.n_states

    def heap_bytes(self):
        return lib().orc_cheap_bytes(self._h)

    def states(self):
        """(n,4) u32 [base, check, fail, output_pos]"""
        n = self._h.contents.n_states
        buf = (C.c_uint32 * (n * 4)).from_address(self._h.contents.states)
        return np.frombuffer(buf, dtype=np.uint32).reshape(n, 4).copy()

    def table(self):
        n = self._h.contents.n_table
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        buf = (C.c_uint32 * n).from_address(self._h.contents.table)
        return np.frombuffer(buf, dtype=np.uint32).copy()

    def outputs(self):
        n = self._h.contents.n_outputs
        if n == 0:
            return np.zeros((0, 3), dtype=np.uint32)
        buf = (C.c_uint32 * (n * 3)).from_address(self._h.contents.outputs)
        return np.frombuffer(buf, dtype=np.uint32).reshape(n, 3).copy()

    def _scan(self, fname, haystack):
        a = _hay(haystack)
        out, n = C.c_void_p(), C.c_size_t()
        rc = getattr(lib(), fname)(self._h, a.ctypes.data if a.size else None, a.size, C.byref(out), C.byref(n))
        if rc:
            raise OracleError(rc, fname)
        if n.value:
            buf = (C.c_char * (n.value * MATCH_DTYPE.itemsize)).from_address(out.value)
            res = np.frombuffer(buf, dtype=MATCH_DTYPE).copy()
        else:
            res = np.zeros(0, dtype=MATCH_DTYPE)
        lib().orc_free(out)
        return res

    def find_iter(self, h):
        return self._scan("orc_cfind_iter", h)

    def find_overlapping_iter(self, h):
        return self._scan("orc_cfind_overlapping_iter", h)

    def find_overlapping_no_suffix_iter(self, h):
        return self._scan("orc_cfind_overlapping_no_suffix_iter", h)

    def leftmost_find_iter(self, h):
        return self._scan("orc_cleftmost_find_iter", h)

    def find_stepper(self, h):
        return self._scan("orc_cfind_stepper", h)

    def find_overlapping_stepper(self, h):
        return self._scan("orc_cfind_overlapping_stepper", h)


def matches_checksum(m):
    m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
    return lib().orc_matches_checksum(m.ctypes.data if m.size else None, m.size)


def triples_sev(m):
    """structured match array -> list of (start, end, value)"""
    return [(int(x["start"]), int(x["end"]), int(x["value"])) for x in m]
