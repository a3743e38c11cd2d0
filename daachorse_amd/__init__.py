"""daachorse_amd — MI355X-native scan path for the daachorse double-array Aho-Corasick automaton.

Only the scan path (find_overlapping_iter / find_iter / leftmost_find_iter over
DoubleArrayAhoCorasick<u32> and CharwiseDoubleArrayAhoCorasick<u32>) lives here, behind a C ABI (include/daachorse_amd.h) implemented
with hand-written HIP kernels for gfx950.  Importing the package loads the HIP library and
raises ImportError if it has not been built: there is no CPU fallback.
"""
from . import _ffi
from ._ffi import DaachorseError, last_engine, last_kernel, set_option
from .bytewise import (DoubleArrayAhoCorasick, DoubleArrayAhoCorasickBuilder, Engine, Match, MatchKind, ScanMode,
                       MATCH_DTYPE, MATCH16_DTYPE, scan_count_multi)
from .charwise import CharwiseDoubleArrayAhoCorasick, CharwiseDoubleArrayAhoCorasickBuilder

_ffi.lib()  # fail loudly at import time if the extension is missing

__all__ = ["DoubleArrayAhoCorasick", "DoubleArrayAhoCorasickBuilder", "CharwiseDoubleArrayAhoCorasick",
           "CharwiseDoubleArrayAhoCorasickBuilder", "Match", "MatchKind", "ScanMode", "Engine",
           "DaachorseError", "set_option", "last_engine", "last_kernel", "MATCH_DTYPE", "MATCH16_DTYPE", "scan_count_multi"]
