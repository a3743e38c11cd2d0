//! `extern "C"` mirror of include/daachorse_amd.h (ABI version 4).  Plain pointers and sizes only.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_void};

pub const DAAC_ABI_VERSION: u32 = 6;

/// daac_status
pub const DAAC_OK: i32 = 0;
pub const DAAC_ERR_MATCH_KIND: i32 = 5; // the crate's assert! panics (bytewise.rs:194-197, 299-302, 415-418, 551-554)
pub const DAAC_ERR_UNSUPPORTED: i32 = 6;
/// daac_scan_mode
pub const DAAC_FIND_OVERLAPPING: i32 = 0; // FindOverlappingIterator          bytewise/iter.rs:117-177
pub const DAAC_FIND: i32 = 1; //              FindIterator                     bytewise/iter.rs:44-114
pub const DAAC_LEFTMOST_FIND: i32 = 2; //     LeftmostFindIterator             bytewise/iter.rs:247-341
pub const DAAC_FIND_OVERLAPPING_NO_SUFFIX: i32 = 3; // FindOverlappingNoSuffixIterator bytewise/iter.rs:180-244
pub const DAAC_ENGINE_AUTO: i32 = 0;

/// Match<u32> as the library reports it (src/lib.rs:286-320: start() = end - length)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct daac_match {
    pub start: u64,
    pub end: u64,
    pub value: u32,
    pub _pad: u32,
}
/// the 16-byte device tuple of daac_scan_device16: the crate's own `Match` fields (src/lib.rs:287-291)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct daac_match16 {
    pub end: u64,
    pub length: u32,
    pub value: u32,
}
/// the 8-byte tuple of the compact lazy iterator (daac_iter_open_compact): end = run base + (end_len & ((1 << end_bits) - 1)),
/// length = end_len >> end_bits
#[repr(C)]
#[derive(Clone, Copy)]
pub struct daac_match8 {
    pub value: u32,
    pub end_len: u32,
}
/// one device's part of a haystack for daac_scan_count_multi (include/daachorse_amd.h: daac_shard)
#[repr(C)]
pub struct daac_shard {
    pub device: i32,
    pub hay: *const u8,
    pub halo: usize,
    pub len: usize,
    pub base: u64,
}
#[repr(C)]
pub struct daac_pma {
    _p: [u8; 0],
}
#[repr(C)]
pub struct daac_iter {
    _p: [u8; 0],
}
#[repr(C)]
pub struct daac_stream {
    _p: [u8; 0],
}
#[repr(C)]
pub struct daac_matches {
    _p: [u8; 0],
}

#[link(name = "daachorse_amd")]
extern "C" {
    pub fn daac_abi_version() -> u32;
    pub fn daac_last_error() -> *const c_char;
    pub fn daac_bytewise_from_serialized(blob: *const u8, len: usize, out: *mut *mut daac_pma, consumed: *mut usize) -> i32;
    pub fn daac_bytewise_from_parts(states: *const u32, n_states: usize, lstates: *const u32, fails: *const u32, n_lstates: usize,
                                    outputs: *const u32, n_outputs: usize, match_kind: u8, num_states: u32, out: *mut *mut daac_pma) -> i32;
    pub fn daac_charwise_from_serialized(blob: *const u8, len: usize, out: *mut *mut daac_pma, consumed: *mut usize) -> i32;
    pub fn daac_pma_upload(pma: *mut daac_pma, device: i32) -> i32;
    pub fn daac_pma_free(pma: *mut daac_pma);
    pub fn daac_iter_open(pma: *mut daac_pma, mode: i32, engine: i32, hay: *const u8, len: usize, hay_is_device: i32,
                          stream: *mut c_void, out: *mut *mut daac_iter) -> i32;
    /// the same iterator with 8 bytes per tuple over PCIe (what bounds `next()` on match-dense text); read with daac_iter_next_batch8.
    /// DAAC_ERR_UNSUPPORTED for dictionaries with patterns of several KB (no room for a window in the packed word).
    pub fn daac_iter_open_compact(pma: *mut daac_pma, mode: i32, engine: i32, hay: *const u8, len: usize, hay_is_device: i32,
                                  stream: *mut c_void, out: *mut *mut daac_iter) -> i32;
    pub fn daac_iter_next_batch8(it: *mut daac_iter, batch: *mut *const daac_match8, n: *mut usize, end_base: *mut u64, end_bits: *mut u32) -> i32;
    pub fn daac_iter_next(it: *mut daac_iter, m: *mut daac_match) -> i32; // 1 = Some, 0 = None, < 0 = -daac_status
    /// the next run of matches as a view of the iterator's own window buffer (valid until the next call): 1 = a run, 0 = exhausted
    pub fn daac_iter_next_batch(it: *mut daac_iter, batch: *mut *const daac_match16, n: *mut usize) -> i32;
    pub fn daac_iter_close(it: *mut daac_iter);
    pub fn daac_scan_count_only_range(pma: *mut daac_pma, mode: i32, engine: i32, hay: *const u8, len: usize, begin: usize,
                                      hay_is_device: i32, stream: *mut c_void, count: *mut u64, result_dev: *mut u64) -> i32;
    /// one haystack sharded across the devices of a node: counts (and checksum sums) of the shards added on the host
    pub fn daac_scan_count_multi(pma: *mut daac_pma, mode: i32, engine: i32, shards: *const daac_shard, n: usize, hay_is_device: i32,
                                 count: *mut u64, checksum: *mut u64) -> i32;
    pub fn daac_scan_count(pma: *mut daac_pma, mode: i32, engine: i32, hay: *const u8, len: usize, hay_is_device: i32,
                           stream: *mut c_void, count: *mut u64, checksum: *mut u64, result_dev: *mut u64) -> i32;
    pub fn daac_scan_device(pma: *mut daac_pma, mode: i32, engine: i32, hay: *const u8, len: usize, hay_is_device: i32,
                            stream: *mut c_void, dev_out: *mut *mut daac_match, count: *mut u64) -> i32;
    pub fn daac_scan_device16(pma: *mut daac_pma, mode: i32, engine: i32, hay: *const u8, len: usize, hay_is_device: i32,
                              stream: *mut c_void, dev_out: *mut *mut daac_match16, count: *mut u64) -> i32;
    pub fn daac_device_free(p: *mut c_void);
    /// an option for one handle (overrides the process-wide daac_set_option value; unset != 0 removes the override)
    pub fn daac_pma_set_option(pma: *mut daac_pma, name: *const c_char, value: i64, unset: i32) -> i32;
    /// releases the scratch a handle keeps between calls (tables stay)
    pub fn daac_pma_trim(pma: *mut daac_pma) -> i32;
    // the chunk-fed steppers (stepper_hip.rs): FindStepper / FindOverlappingStepper / the *_from_iter entry points
    pub fn daac_stream_open(pma: *mut daac_pma, mode: i32, engine: i32, stream: *mut c_void, out: *mut *mut daac_stream) -> i32;
    pub fn daac_stream_feed(s: *mut daac_stream, chunk: *const u8, len: usize, chunk_is_device: i32, out: *mut *mut daac_matches) -> i32;
    pub fn daac_stream_feed_compact(s: *mut daac_stream, chunk: *const u8, len: usize, chunk_is_device: i32, batch: *mut *const daac_match8, n: *mut usize,
                                    end_base: *mut u64, end_bits: *mut u32) -> i32; // ABI 6
    pub fn daac_stream_close(s: *mut daac_stream);
    pub fn daac_matches_count(m: *const daac_matches) -> usize;
    pub fn daac_matches_data(m: *const daac_matches) -> *const daac_match;
    pub fn daac_matches_free(m: *mut daac_matches);
}

/// Owner of a `daac_pma*`: the device twin of one automaton, immutable after upload.
pub struct HipPma(pub(crate) *mut daac_pma);
unsafe impl Send for HipPma {}
unsafe impl Sync for HipPma {}
impl Drop for HipPma {
    fn drop(&mut self) {
        unsafe { daac_pma_free(self.0) }
    }
}

/// What every iterator of the crate holds under the `hip` feature instead of (state_id, pos, output_pos):
/// the open `daac_iter`, the run of matches it handed out last, and what `count()` needs to run as one device pass.
///
/// SAFETY contract with the iterator structs: `hay_ptr` points into the haystack the iterator owns, and `daac_iter` reads it lazily,
/// window by window, for as long as the cursor lives.  `P: AsRef<[u8]>` admits haystacks that keep their bytes INLINE (`[u8; N]`,
/// ArrayVec, SmallVec): moving such a value moves the bytes.  The iterators therefore keep the haystack in a `Box` (made BEFORE the
/// cursor is opened, see bytewise_hip.rs) — the bytes a boxed `P` hands out do not move when the iterator itself is moved — and declare
/// the cursor field before the haystack field, so that the cursor (and the worker thread behind it) is gone before the bytes are.
pub struct HipCursor<'a> {
    pub(crate) it: *mut daac_iter,
    pub(crate) pma: &'a HipPma,
    pub(crate) mode: i32,
    pub(crate) hay_ptr: *const u8,
    pub(crate) hay_len: usize,
    pub(crate) compact: bool,            // 8-byte runs (else: the dictionary's patterns are too long for them, 16-byte runs)
    pub(crate) run: *const daac_match8,  // what is left of the last daac_iter_next_batch8 ...
    pub(crate) run16: *const daac_match16, // ... or daac_iter_next_batch
    pub(crate) run_len: usize,
    pub(crate) run_base: u64, // ends of the run count from here
    pub(crate) end_bits: u32,
    pub(crate) consumed: bool, // next() has been called: count() must not restart
}
impl<'a> HipCursor<'a> {
    /// Opens the lazy façade over `hay`, which must stay where it is until the cursor is dropped (see the struct's contract).
    /// Panics exactly where the crate panics (wrong MatchKind), with the crate's messages.
    pub(crate) fn open(pma: &'a HipPma, mode: i32, hay: &[u8]) -> Self {
        let mut it = core::ptr::null_mut();
        let mut compact = true;
        let mut st = unsafe { daac_iter_open_compact(pma.0, mode, DAAC_ENGINE_AUTO, hay.as_ptr(), hay.len(), 0, core::ptr::null_mut(), &mut it) };
        if st == DAAC_ERR_UNSUPPORTED {
            compact = false;
            st = unsafe { daac_iter_open(pma.0, mode, DAAC_ENGINE_AUTO, hay.as_ptr(), hay.len(), 0, core::ptr::null_mut(), &mut it) };
        }
        assert!(!(st == DAAC_ERR_MATCH_KIND && mode != DAAC_LEFTMOST_FIND), "Error: match_kind must be standard.");
        assert!(!(st == DAAC_ERR_MATCH_KIND && mode == DAAC_LEFTMOST_FIND), "Error: match_kind must be leftmost.");
        assert!(st == DAAC_OK, "daachorse_amd: device scan failed (status {st})");
        Self { it, pma, mode, hay_ptr: hay.as_ptr(), hay_len: hay.len(), compact, run: core::ptr::null(), run16: core::ptr::null(), run_len: 0,
               run_base: 0, end_bits: 32, consumed: false }
    }
    /// `Iterator::next`: one 8-byte read from the run at hand; a library call per WINDOW (tens of millions of matches), not per match.
    #[inline]
    pub(crate) fn next(&mut self) -> Option<crate::Match<u32>> {
        self.consumed = true;
        if self.run_len == 0 {
            let (mut n, mut base, mut eb) = (0usize, 0u64, 32u32);
            let r = if self.compact {
                let mut p = core::ptr::null();
                let r = unsafe { daac_iter_next_batch8(self.it, &mut p, &mut n, &mut base, &mut eb) };
                self.run = p;
                r
            } else {
                let mut p = core::ptr::null();
                let r = unsafe { daac_iter_next_batch(self.it, &mut p, &mut n) };
                self.run16 = p;
                r
            };
            match r {
                1 => {
                    self.run_len = n;
                    self.run_base = base;
                    self.end_bits = eb;
                }
                0 => return None,
                e => panic!("daachorse_amd: device scan failed (status {})", -e),
            }
        }
        self.run_len -= 1;
        if !self.compact {
            let t = unsafe { *self.run16 }; // the crate's own Match fields (src/lib.rs:287-291)
            self.run16 = unsafe { self.run16.add(1) };
            return Some(crate::Match { length: t.length as usize, end: t.end as usize, value: t.value });
        }
        let t = unsafe { *self.run };
        self.run = unsafe { self.run.add(1) };
        let end = self.run_base + (t.end_len & ((1u32 << self.end_bits) - 1)) as u64;
        Some(crate::Match { length: (t.end_len >> self.end_bits) as usize, end: end as usize, value: t.value })
    }
    /// `Iterator::count()` as ONE device pass (`daac_scan_count_only_range`) when nothing has been pulled yet.
    pub(crate) fn count(mut self) -> usize {
        if self.consumed {
            let mut n = 0;
            while self.next().is_some() {
                n += 1;
            }
            return n;
        }
        let mut n = 0u64;
        let st = unsafe {
            daac_scan_count_only_range(self.pma.0, self.mode, DAAC_ENGINE_AUTO, self.hay_ptr, self.hay_len, 0, 0,
                                       core::ptr::null_mut(), &mut n, core::ptr::null_mut())
        };
        assert!(st == DAAC_OK, "daachorse_amd: device scan failed (status {st})");
        n as usize
    }
}
impl Drop for HipCursor<'_> {
    fn drop(&mut self) {
        unsafe { daac_iter_close(self.it) }
    }
}
