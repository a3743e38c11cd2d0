//! src/bytewise/stepper_hip.rs — the chunk-fed forms of the crate's steppers and `*_from_iter` entry points on the device
//! (feature `hip`), over `daac_stream_*` of include/daachorse_amd.h.
//!
//! The crate's steppers take ONE byte per call (`FindStepper::consume`, `FindOverlappingStepper::consume`: src/bytewise/iter.rs:357-395,
//! 449-474; made by `find_stepper` / `find_overlapping_stepper`, src/bytewise.rs:627-645, 719-729): a device round trip per byte would be
//! absurd, so those two methods keep the crate's CPU bodies under the feature as well.  What moves to the device is their bulk form:
//!
//!   * `find_iter_from_iter` / `find_overlapping_iter_from_iter` (src/bytewise.rs:238-251, 353-375), same names, same return types
//!     (`FindIterator<'_, P, u32>` / `FindOverlappingIterator<'_, P, u32>` with `P: Iterator<Item = u8>`): the iterator drains `P` a
//!     chunk at a time into a page-locked buffer, feeds it to a `daac_stream`, and hands out that chunk's matches — the stream carries
//!     what the reference carries in `state_id` (a halo of max_pattern_len - 1 bytes, or FindIterator's restart point) from chunk to chunk;
//!   * `FindStepper::consume_slice` / `FindOverlappingStepper::consume_slice` (additions): the matches a run of `consume` + `matches`
//!     calls over the slice would have reported, in that order.
//!
//! The structs of src/bytewise/iter.rs get one more private field under the feature: `chunked: Option<HipChunked<'a>>`
//! (None for iterators made by the slice entry points, which go through `HipCursor`).
#![cfg(feature = "hip")]

use std::collections::VecDeque;

use crate::bytewise::iter::{FindIterator, FindOverlappingIterator, FindOverlappingStepper, FindStepper};
use crate::hip::ffi::*;
use crate::{DoubleArrayAhoCorasick, Match};

/// bytes drained from the source iterator per feed (one device scan + one copy of the chunk's tuples back)
pub const HIP_CHUNK_BYTES: usize = 64 << 20;

/// Owner of a `daac_stream*`: the device side of one stepper / `*_from_iter` iterator.
pub struct HipChunked<'a> {
    pub(crate) s: *mut daac_stream,
    pub(crate) _pma: &'a HipPma,
    pub(crate) buf: Vec<u8>,
    pub(crate) out: VecDeque<Match<u32>>,
    pub(crate) done: bool,
}
impl<'a> HipChunked<'a> {
    pub(crate) fn open(pma: &'a HipPma, mode: i32) -> Self {
        let mut s = core::ptr::null_mut();
        let st = unsafe { daac_stream_open(pma.0, mode, DAAC_ENGINE_AUTO, core::ptr::null_mut(), &mut s) };
        assert!(st != DAAC_ERR_MATCH_KIND, "Error: match_kind must be standard."); // src/bytewise.rs:242-245, 360-363
        assert!(st == DAAC_OK, "daachorse_amd: stream open failed (status {st})");
        Self { s, _pma: pma, buf: Vec::new(), out: VecDeque::new(), done: false }
    }
    /// One feed: the matches the chunk decides (FindIterator: up to its last restart point), appended to `out` in the iterator's order.
    /// The tuples come back as 8-byte `daac_match8` in the stream object's page-locked block (`daac_stream_feed_compact`, ABI 6: a third of
    /// `daac_match`'s bytes over PCIe, no list to allocate and free per feed); a dictionary whose longest pattern leaves no room for a
    /// 64 MiB chunk in the packed word (status 6) takes the 24-byte form.
    pub(crate) fn feed(&mut self, chunk: &[u8]) {
        let (mut run, mut n, mut base, mut bits) = (core::ptr::null(), 0usize, 0u64, 0u32);
        let st = unsafe { daac_stream_feed_compact(self.s, chunk.as_ptr(), chunk.len(), 0, &mut run, &mut n, &mut base, &mut bits) };
        if st == DAAC_OK {
            self.out.reserve(n);
            let mask = (1u32 << bits) - 1;
            for i in 0..n {
                let t = unsafe { *run.add(i) }; // daac_match8 {value, end - base | length << bits}
                self.out.push_back(Match { length: (t.end_len >> bits) as usize, end: (base + (t.end_len & mask) as u64) as usize, value: t.value });
            }
            return;
        }
        assert!(st == DAAC_ERR_UNSUPPORTED, "daachorse_amd: stream feed failed (status {st})");
        let mut m = core::ptr::null_mut();
        let st = unsafe { daac_stream_feed(self.s, chunk.as_ptr(), chunk.len(), 0, &mut m) };
        assert!(st == DAAC_OK, "daachorse_amd: stream feed failed (status {st})");
        let (n, p) = unsafe { (daac_matches_count(m), daac_matches_data(m)) };
        self.out.reserve(n);
        for i in 0..n {
            let t = unsafe { *p.add(i) }; // daac_match {start, end, value, pad}
            self.out.push_back(Match { length: (t.end - t.start) as usize, end: t.end as usize, value: t.value });
        }
        unsafe { daac_matches_free(m) };
    }
    /// `Iterator::next` of the `*_from_iter` iterators: refills from `src` when the matches at hand are used up.
    pub(crate) fn next_from<I: Iterator<Item = u8>>(&mut self, src: &mut I) -> Option<Match<u32>> {
        loop {
            if let Some(m) = self.out.pop_front() {
                return Some(m);
            }
            if self.done {
                return None;
            }
            self.buf.clear();
            self.buf.extend(src.by_ref().take(HIP_CHUNK_BYTES));
            if self.buf.len() < HIP_CHUNK_BYTES {
                self.done = true; // the source is exhausted: the last feed, then an empty one flushes what FindIterator still holds back
            }
            let chunk = core::mem::take(&mut self.buf);
            self.feed(&chunk);
            self.buf = chunk;
            if self.done {
                self.feed(&[]);
            }
        }
    }
}
impl Drop for HipChunked<'_> {
    fn drop(&mut self) {
        unsafe { daac_stream_close(self.s) }
    }
}

impl DoubleArrayAhoCorasick<u32> {
    /// src/bytewise.rs:238-251
    pub fn find_iter_from_iter<P>(&self, haystack: P) -> FindIterator<'_, P, u32>
    where
        P: Iterator<Item = u8>,
    {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard.");
        FindIterator::from_chunked(HipChunked::open(self.hip(), DAAC_FIND), haystack)
    }
    /// src/bytewise.rs:353-375
    pub fn find_overlapping_iter_from_iter<P>(&self, haystack: P) -> FindOverlappingIterator<'_, P, u32>
    where
        P: Iterator<Item = u8>,
    {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard.");
        FindOverlappingIterator::from_chunked(HipChunked::open(self.hip(), DAAC_FIND_OVERLAPPING), haystack)
    }
}

impl<'a> FindOverlappingStepper<'a, u32> {
    /// Every match a run of `consume(b)` + `matches()` over `chunk` would have reported (src/bytewise/iter.rs:449-474), in that order,
    /// from one device scan.  Mixed use with `consume` is not supported on one stepper (the CPU state and the device stream do not know
    /// of each other): a stepper is byte-fed or slice-fed.
    pub fn consume_slice(&mut self, chunk: &[u8]) -> Vec<Match<u32>> {
        let c = self.hip_stream.get_or_insert_with(|| HipChunked::open(self.pma.hip(), DAAC_FIND_OVERLAPPING));
        c.feed(chunk);
        self.pos += chunk.len();
        c.out.drain(..).collect()
    }
}
impl<'a> FindStepper<'a, u32> {
    /// The same for `FindStepper` (src/bytewise/iter.rs:357-395).  FindIterator's chain may leave the chunk's last bytes undecided;
    /// they are reported with the next slice (or by `consume_slice(&[])` at the end of the input).
    pub fn consume_slice(&mut self, chunk: &[u8]) -> Vec<Match<u32>> {
        let c = self.hip_stream.get_or_insert_with(|| HipChunked::open(self.pma.hip(), DAAC_FIND));
        c.feed(chunk);
        self.pos += chunk.len();
        c.out.drain(..).collect()
    }
}
