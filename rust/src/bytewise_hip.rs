//! src/bytewise/hip.rs — `DoubleArrayAhoCorasick<u32>` on the MI355X behind the crate's own API (feature `hip`).
//!
//! Signatures are the crate's, verbatim:
//!   find_iter                        src/bytewise.rs:190-193  -> FindIterator<'_, U8SliceIterator<P>, V>
//!   find_overlapping_iter            src/bytewise.rs:292-297  -> FindOverlappingIterator<'_, U8SliceIterator<P>, V>
//!   find_overlapping_no_suffix_iter  src/bytewise.rs:410-413  -> FindOverlappingNoSuffixIterator<'_, U8SliceIterator<P>, V>
//!   leftmost_find_iter               src/bytewise.rs:547-550  -> LeftmostFindIterator<'_, P, V>
//! with V = u32.  The four iterator structs of src/bytewise/iter.rs keep name and type parameters; under the feature their
//! private fields are `{ cur: HipCursor<'a>, haystack: Box<I> }` (the haystack stays owned/borrowed by the iterator exactly as
//! `U8SliceIterator<P>` keeps `P`, iter.rs:14-41 — boxed, because the device side reads the bytes lazily through a raw pointer and a `P`
//! with inline storage would carry them along when the iterator is moved: ffi.rs, HipCursor), and the CPU bodies of these four methods
//! are `#[cfg(not(feature = "hip"))]`.
#![cfg(feature = "hip")]

use std::sync::OnceLock;

use crate::bytewise::iter::{FindIterator, FindOverlappingIterator, FindOverlappingNoSuffixIterator, LeftmostFindIterator, U8SliceIterator};
use crate::hip::ffi::*;
use crate::{DoubleArrayAhoCorasick, Match, MatchKind};

// ---- iterator structs under the feature (replace the field lists in src/bytewise/iter.rs:44-57, 117-131, 180-193, 247-258) ----
// field order = drop order: the cursor (worker thread, raw pointer into the haystack) goes before the haystack
pub struct FindIteratorHipFields<'a, I> { pub(crate) cur: HipCursor<'a>, pub(crate) haystack: Box<I> }
// pub struct FindIterator<'a, I, V>                    { f: FindIteratorHipFields<'a, I>, _v: PhantomData<V> }
// pub struct FindOverlappingIterator<'a, I, V>         { f: FindIteratorHipFields<'a, I>, _v: PhantomData<V> }
// pub struct FindOverlappingNoSuffixIterator<'a, I, V> { f: FindIteratorHipFields<'a, I>, _v: PhantomData<V> }
// pub struct LeftmostFindIterator<'a, P, V>            { f: FindIteratorHipFields<'a, P>, _v: PhantomData<V> }

macro_rules! hip_iterator {
    ($name:ident, $hay:ident) => {
        impl<'a, $hay> Iterator for $name<'a, $hay, u32> {
            type Item = Match<u32>;
            #[inline]
            fn next(&mut self) -> Option<Match<u32>> {
                self.f.cur.next()
            }
            /// `.count()` is one device pass, not a walk through `next` (1.3 TB/s for the 100 k-word dictionary)
            fn count(self) -> usize {
                self.f.cur.count()
            }
        }
    };
}
hip_iterator!(FindIterator, I);
hip_iterator!(FindOverlappingIterator, I);
hip_iterator!(FindOverlappingNoSuffixIterator, I);
hip_iterator!(LeftmostFindIterator, P);

impl DoubleArrayAhoCorasick<u32> {
    /// The device twin, made on first use (serialize() -> daac_bytewise_from_serialized -> daac_pma_upload); a new field
    /// `hip: OnceLock<HipPma>` of the struct (src/bytewise.rs:55-66).  The blob route needs no `repr` on the crate's records;
    /// `to_hip_from_parts` skips the copy once `State<F>` / `Output<V>` are `#[repr(C)]`.
    fn hip(&self) -> &HipPma {
        self.hip.get_or_init(|| {
            assert_eq!(unsafe { daac_abi_version() }, DAAC_ABI_VERSION, "libdaachorse_amd.so: ABI version mismatch");
            let blob = self.serialize(); // src/bytewise.rs:801-820
            let (mut h, mut used) = (core::ptr::null_mut(), 0usize);
            let st = unsafe { daac_bytewise_from_serialized(blob.as_ptr(), blob.len(), &mut h, &mut used) };
            assert!(st == DAAC_OK && used == blob.len(), "daachorse_amd: automaton rejected (status {st})");
            let pma = HipPma(h);
            let device = std::env::var("DAACHORSE_AMD_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            assert_eq!(unsafe { daac_pma_upload(pma.0, device) }, DAAC_OK, "daachorse_amd: upload failed");
            pma
        })
    }

    /// src/bytewise.rs:190-193
    pub fn find_iter<P>(&self, haystack: P) -> FindIterator<'_, U8SliceIterator<P>, u32>
    where
        P: AsRef<[u8]>,
    {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard."); // src/bytewise.rs:194-197
        // the haystack first, in its final place; THEN the cursor, over the bytes as the box hands them out (`inner`: iter.rs:15)
        let haystack = Box::new(U8SliceIterator::new(haystack));
        let cur = HipCursor::open(self.hip(), DAAC_FIND, haystack.inner.as_ref());
        FindIterator { f: FindIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }

    /// src/bytewise.rs:292-297
    pub fn find_overlapping_iter<P>(&self, haystack: P) -> FindOverlappingIterator<'_, U8SliceIterator<P>, u32>
    where
        P: AsRef<[u8]>,
    {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard."); // src/bytewise.rs:299-302
        // the haystack first, in its final place; THEN the cursor, over the bytes as the box hands them out (`inner`: iter.rs:15)
        let haystack = Box::new(U8SliceIterator::new(haystack));
        let cur = HipCursor::open(self.hip(), DAAC_FIND_OVERLAPPING, haystack.inner.as_ref());
        FindOverlappingIterator { f: FindIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }

    /// src/bytewise.rs:410-413
    pub fn find_overlapping_no_suffix_iter<P>(&self, haystack: P) -> FindOverlappingNoSuffixIterator<'_, U8SliceIterator<P>, u32>
    where
        P: AsRef<[u8]>,
    {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard."); // src/bytewise.rs:415-418
        // the haystack first, in its final place; THEN the cursor, over the bytes as the box hands them out (`inner`: iter.rs:15)
        let haystack = Box::new(U8SliceIterator::new(haystack));
        let cur = HipCursor::open(self.hip(), DAAC_FIND_OVERLAPPING_NO_SUFFIX, haystack.inner.as_ref());
        FindOverlappingNoSuffixIterator { f: FindIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }

    /// src/bytewise.rs:547-550
    pub fn leftmost_find_iter<P>(&self, haystack: P) -> LeftmostFindIterator<'_, P, u32>
    where
        P: AsRef<[u8]>,
    {
        assert!(self.match_kind.is_leftmost(), "Error: match_kind must be leftmost."); // src/bytewise.rs:551-554
        let haystack = Box::new(haystack);
        let cur = HipCursor::open(self.hip(), DAAC_LEFTMOST_FIND, (*haystack).as_ref());
        LeftmostFindIterator { f: FindIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }

    /// Beyond the crate's API: the whole match list left in device memory, in the iterator's order, as the crate's own
    /// `Match` fields {end: u64, length: u32, value: u32} (16 bytes each) — for consumers that run on the GPU.
    pub fn find_overlapping_device<P: AsRef<[u8]>>(&self, haystack: P) -> DeviceMatches {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard.");
        let h = haystack.as_ref();
        let (mut p, mut n) = (core::ptr::null_mut(), 0u64);
        let st = unsafe { daac_scan_device16(self.hip().0, DAAC_FIND_OVERLAPPING, DAAC_ENGINE_AUTO, h.as_ptr(), h.len(), 0, core::ptr::null_mut(), &mut p, &mut n) };
        assert!(st == DAAC_OK, "daachorse_amd: device scan failed (status {st})");
        DeviceMatches { ptr: p, len: n as usize }
    }
}

/// A match list in HBM (daac_scan_device16); freed with the library's allocator.
pub struct DeviceMatches {
    pub ptr: *mut daac_match16,
    pub len: usize,
}
impl Drop for DeviceMatches {
    fn drop(&mut self) {
        unsafe { daac_device_free(self.ptr.cast()) }
    }
}

#[allow(dead_code)]
fn _kind_is_what_the_library_expects(k: MatchKind) -> u8 {
    k as u8 // src/lib.rs:324-346: repr(u8), Standard = 0, LeftmostLongest = 1, LeftmostFirst = 2 = daac_match_kind
}
