//! src/charwise/hip.rs — `CharwiseDoubleArrayAhoCorasick<u32>` behind the crate's API (feature `hip`).
//! Signatures verbatim: find_iter src/charwise.rs:184-187, find_overlapping_iter :290-293,
//! find_overlapping_no_suffix_iter :412-415, leftmost_find_iter :553-556 (all `P: AsRef<str>`); positions stay BYTE offsets
//! (src/charwise/iter.rs).  The automaton goes over as its serialize() blob (src/charwise.rs:831-848): 16-byte State records,
//! CodeMapper table and outputs are taken as they are.
#![cfg(feature = "hip")]

use crate::charwise::iter::{FindIterator, FindOverlappingIterator, FindOverlappingNoSuffixIterator, LeftmostFindIterator, StrIterator};
use crate::hip::ffi::*;
use crate::{CharwiseDoubleArrayAhoCorasick, Match};

// boxed for the reason given at HipCursor (ffi.rs): the device side reads the bytes lazily through a raw pointer, and a haystack that
// keeps its bytes inline must not carry them along when the iterator is moved; field order = drop order (cursor first)
pub struct CharIteratorHipFields<'a, P> { pub(crate) cur: HipCursor<'a>, pub(crate) haystack: Box<P> }
// under the feature: pub struct FindIterator<'a, I, V> { f: CharIteratorHipFields<'a, I>, _v: PhantomData<V> }   (and the other three)

macro_rules! hip_iterator {
    ($name:ident) => {
        impl<'a, P> Iterator for $name<'a, P, u32> {
            type Item = Match<u32>;
            #[inline]
            fn next(&mut self) -> Option<Match<u32>> {
                self.f.cur.next()
            }
            fn count(self) -> usize {
                self.f.cur.count()
            }
        }
    };
}
hip_iterator!(FindIterator);
hip_iterator!(FindOverlappingIterator);
hip_iterator!(FindOverlappingNoSuffixIterator);
hip_iterator!(LeftmostFindIterator);

impl CharwiseDoubleArrayAhoCorasick<u32> {
    fn hip(&self) -> &HipPma {
        self.hip.get_or_init(|| {
            let blob = self.serialize();
            let (mut h, mut used) = (core::ptr::null_mut(), 0usize);
            let st = unsafe { daac_charwise_from_serialized(blob.as_ptr(), blob.len(), &mut h, &mut used) };
            assert!(st == DAAC_OK && used == blob.len(), "daachorse_amd: automaton rejected (status {st})");
            let pma = HipPma(h);
            assert_eq!(unsafe { daac_pma_upload(pma.0, 0) }, DAAC_OK, "daachorse_amd: upload failed");
            pma
        })
    }
    /// src/charwise.rs:184-187
    pub fn find_iter<P: AsRef<str>>(&self, haystack: P) -> FindIterator<'_, StrIterator<P>, u32> {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard."); // src/charwise.rs:104-107
        let haystack = Box::new(StrIterator::new(haystack));   // in its final place before the cursor takes a pointer into it
        let cur = HipCursor::open(self.hip(), DAAC_FIND, haystack.inner.as_ref().as_bytes());
        FindIterator { f: CharIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }
    /// src/charwise.rs:290-293
    pub fn find_overlapping_iter<P: AsRef<str>>(&self, haystack: P) -> FindOverlappingIterator<'_, StrIterator<P>, u32> {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard."); // src/charwise.rs:163-166
        let haystack = Box::new(StrIterator::new(haystack));   // in its final place before the cursor takes a pointer into it
        let cur = HipCursor::open(self.hip(), DAAC_FIND_OVERLAPPING, haystack.inner.as_ref().as_bytes());
        FindOverlappingIterator { f: CharIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }
    /// src/charwise.rs:412-415
    pub fn find_overlapping_no_suffix_iter<P: AsRef<str>>(&self, haystack: P) -> FindOverlappingNoSuffixIterator<'_, StrIterator<P>, u32> {
        assert!(self.match_kind.is_standard(), "Error: match_kind must be standard."); // src/charwise.rs:227-230
        let haystack = Box::new(StrIterator::new(haystack));   // in its final place before the cursor takes a pointer into it
        let cur = HipCursor::open(self.hip(), DAAC_FIND_OVERLAPPING_NO_SUFFIX, haystack.inner.as_ref().as_bytes());
        FindOverlappingNoSuffixIterator { f: CharIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }
    /// src/charwise.rs:553-556
    pub fn leftmost_find_iter<P: AsRef<str>>(&self, haystack: P) -> LeftmostFindIterator<'_, P, u32> {
        assert!(self.match_kind.is_leftmost(), "Error: match_kind must be leftmost."); // src/charwise.rs:309-312
        let haystack = Box::new(haystack);
        let cur = HipCursor::open(self.hip(), DAAC_LEFTMOST_FIND, (*haystack).as_ref().as_bytes());
        LeftmostFindIterator { f: CharIteratorHipFields { cur, haystack }, _v: core::marker::PhantomData }
    }
}
