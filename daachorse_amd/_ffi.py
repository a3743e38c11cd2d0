"""ctypes loader for the C ABI (include/daachorse_amd.h).  Fails loudly: there is no fallback."""
import ctypes as C
import os

from . import _build

_lib = None


class DaachorseError(RuntimeError):
    """Mirror of daachorse::errors::DaachorseError (reference src/errors.rs:10-22) plus the
    boundary's extra statuses; `.code` is the daac_status."""

    NAMES = {1: "InvalidArgument", 2: "AutomatonScale", 3: "InvalidConversion", 4: "InvalidAutomaton",
             5: "MatchKindMismatch", 6: "Unsupported", 7: "Device"}

    def __init__(self, code, msg):
        super().__init__(f"{self.NAMES.get(code, code)}: {msg}")
        self.code = code


class Match(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("value", C.c_uint32), ("_pad", C.c_uint32)]


class Shard(C.Structure):
    """daac_shard: one device's part of a haystack for daac_scan_count_multi"""
    _fields_ = [("device", C.c_int), ("hay", C.c_void_p), ("halo", C.c_size_t), ("len", C.c_size_t), ("base", C.c_uint64)]


class Info(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("match_kind", C.c_uint8), ("num_states", C.c_uint32), ("states_len", C.c_uint64),
                ("outputs_len", C.c_uint64), ("heap_bytes", C.c_uint64), ("max_pattern_len", C.c_uint32),
                ("num_classes", C.c_uint32), ("tier_dense_states", C.c_uint32), ("tier_lds_states", C.c_uint32),
                ("tier_lds_bytes", C.c_uint32), ("tiered_available", C.c_uint8), ("gram_available", C.c_uint8),
                ("gram_k", C.c_uint32), ("gram_lds_bytes", C.c_uint32),
                ("charwise", C.c_uint8), ("alphabet_size", C.c_uint32),
                ("gram2_available", C.c_uint8), ("gram2_exact", C.c_uint8), ("gram2_k", C.c_uint32),
                ("gram2_lds_count", C.c_uint32), ("gram2_lds_exact", C.c_uint32), ("gram_wide", C.c_uint8),
                ("pfx_available", C.c_uint8), ("pfx_key_bytes", C.c_uint32), ("pfx_lds_bytes", C.c_uint32),
                ("plan_engine", C.c_uint8 * 8), ("plan_kernel", C.c_uint8 * 8), ("plan_reason", C.c_uint8 * 8)]


ABI_VERSION = 6  # DAAC_ABI_VERSION of include/daachorse_amd.h this mirror was written against


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'`).  daachorse_amd has no CPU fallback.")
    # PyTorch-ROCm ships its own libamdhip64; two HIP runtimes in one process cannot both own the
    # GPU.  torch is this package's plumbing for device memory and streams, so its runtime is loaded
    # first and libdaachorse_amd.so binds to the same one.
    import torch  # noqa: F401
    L = C.CDLL(path)
    if not hasattr(L, "daac_abi_version"):
        raise ImportError(f"{path} predates ABI version {ABI_VERSION} (rebuild: __graft_entry__.build())")
    L.daac_abi_version.restype = C.c_uint32
    if L.daac_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {L.daac_abi_version()}, this package mirrors version {ABI_VERSION} (rebuild: __graft_entry__.build())")
    P, vp, sz, u8p = C.POINTER, C.c_void_p, C.c_size_t, C.c_void_p
    L.daac_last_error.restype = C.c_char_p
    L.daac_free.argtypes = [vp]
    L.daac_bytewise_from_serialized.argtypes = [C.c_char_p, sz, P(vp), P(sz)]
    L.daac_bytewise_from_parts.argtypes = [vp, sz, vp, vp, sz, vp, sz, C.c_uint8, C.c_uint32, P(vp)]
    L.daac_bytewise_build.argtypes = [vp, vp, vp, sz, C.c_uint8, C.c_uint32, P(vp)]
    L.daac_charwise_from_serialized.argtypes = [C.c_char_p, sz, P(vp), P(sz)]
    L.daac_charwise_build.argtypes = [vp, vp, vp, sz, C.c_uint8, C.c_uint32, P(vp)]
    L.daac_pma_serialize.argtypes = [vp, P(vp), P(sz)]
    L.daac_pma_info.argtypes = [vp, P(Info)]
    L.daac_pma_explain.argtypes = [vp, C.c_char_p, sz]
    L.daac_pma_explain.restype = sz
    L.daac_pma_free.argtypes = [vp]
    L.daac_pma_upload.argtypes = [vp, C.c_int]
    L.daac_pma_trim.argtypes = [vp]
    L.daac_pma_trim.restype = C.c_int
    L.daac_scan.argtypes = [vp, C.c_int, C.c_int, u8p, sz, C.c_int, vp, P(vp)]
    L.daac_matches_count.argtypes = [vp]
    L.daac_matches_count.restype = sz
    L.daac_matches_data.argtypes = [vp]
    L.daac_matches_data.restype = vp
    L.daac_matches_free.argtypes = [vp]
    L.daac_scan_count.argtypes = [vp, C.c_int, C.c_int, u8p, sz, C.c_int, vp, P(C.c_uint64), P(C.c_uint64), vp]
    L.daac_scan_count_range.argtypes = [vp, C.c_int, C.c_int, u8p, sz, sz, C.c_int, vp, P(C.c_uint64), P(C.c_uint64), vp]
    L.daac_iter_open.argtypes = [vp, C.c_int, C.c_int, u8p, sz, C.c_int, vp, P(vp)]
    L.daac_iter_next.argtypes = [vp, P(Match)]
    L.daac_iter_next.restype = C.c_int
    L.daac_iter_next_batch.argtypes = [vp, P(vp), P(sz)]
    L.daac_iter_next_batch.restype = C.c_int
    L.daac_iter_open_compact.argtypes = [vp, C.c_int, C.c_int, u8p, sz, C.c_int, vp, P(vp)]
    L.daac_iter_next_batch8.argtypes = [vp, P(vp), P(sz), P(C.c_uint64), P(C.c_uint32)]
    L.daac_iter_next_batch8.restype = C.c_int
    L.daac_iter_close.argtypes = [vp]
    L.daac_stream_open.argtypes = [vp, C.c_int, C.c_int, vp, P(vp)]
    L.daac_stream_feed.argtypes = [vp, u8p, sz, C.c_int, P(vp)]
    L.daac_stream_feed_compact.argtypes = [vp, u8p, sz, C.c_int, P(vp), P(sz), P(C.c_uint64), P(C.c_uint32)]
    L.daac_stream_feed_compact.restype = C.c_int
    L.daac_stream_close.argtypes = [vp]
    L.daac_scan_count_only_range.argtypes = [vp, C.c_int, C.c_int, u8p, sz, sz, C.c_int, vp, P(C.c_uint64), vp]
    L.daac_scan_count_only_range.restype = C.c_int
    L.daac_scan_count_multi.argtypes = [vp, C.c_int, C.c_int, P(Shard), sz, C.c_int, P(C.c_uint64), P(C.c_uint64)]
    L.daac_scan_count_multi.restype = C.c_int
    L.daac_scan_device.argtypes = [vp, C.c_int, C.c_int, u8p, sz, C.c_int, vp, P(vp), P(C.c_uint64)]
    L.daac_scan_device.restype = C.c_int
    L.daac_scan_device16.argtypes = [vp, C.c_int, C.c_int, u8p, sz, C.c_int, vp, P(vp), P(C.c_uint64)]
    L.daac_scan_device16.restype = C.c_int
    L.daac_device_free.argtypes = [vp]
    L.daac_device_to_host.argtypes = [vp, vp, sz]
    L.daac_device_to_host.restype = C.c_int
    L.daac_set_option.argtypes = [C.c_char_p, C.c_int64]
    L.daac_pma_set_option.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int]
    L.daac_pma_set_option.restype = C.c_int
    L.daac_last_engine.argtypes = []
    L.daac_last_engine.restype = C.c_int
    L.daac_last_kernel.argtypes = []
    L.daac_last_kernel.restype = C.c_char_p
    L.daac_synth_uniform.argtypes = [vp, sz, C.c_uint64, vp, C.c_uint32, C.c_uint64, vp]
    L.daac_synth_wordsoup.argtypes = [vp, sz, C.c_uint64, vp, vp, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32,
                                      vp, C.c_uint32, C.c_uint64, vp]
    L.daac_synth_zipf_text.argtypes = [vp, sz, C.c_uint64, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint64, vp]
    for name in ("daac_bytewise_from_serialized", "daac_bytewise_from_parts", "daac_bytewise_build", "daac_charwise_from_serialized",
                 "daac_charwise_build", "daac_pma_serialize",
                 "daac_pma_info", "daac_pma_upload", "daac_scan", "daac_scan_count", "daac_scan_count_range", "daac_iter_open", "daac_iter_open_compact", "daac_stream_open", "daac_stream_feed", "daac_set_option",
                 "daac_synth_uniform", "daac_synth_wordsoup", "daac_synth_zipf_text"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(status):
    if status != 0:
        msg = lib().daac_last_error()
        raise DaachorseError(status, msg.decode("utf-8", "replace") if msg else "")


def last_engine():
    """daac_engine that served this thread's most recent scan"""
    return lib().daac_last_engine()


def last_kernel():
    """daac_last_kernel: kernel family + launch shape of this thread's most recent count (a diagnostic string)"""
    return lib().daac_last_kernel().decode()


def set_option(name, value):
    check(lib().daac_set_option(name.encode(), int(value)))
