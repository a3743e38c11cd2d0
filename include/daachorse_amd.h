/*
 * daachorse_amd.h — C ABI of the MI355X-native daachorse scan path.
 *
 * This is the drop-in boundary for ONE path of the daachorse crate (v4.0.0): the bytewise
 * double-array Aho-Corasick scan (find_overlapping_iter / find_iter / leftmost_find_iter and
 * friends over DoubleArrayAhoCorasick<u32>).  The reference has no FFI of its own; every entry
 * point below names the Rust item (file:line under the reference tree) whose work it takes
 * over, and INTEGRATION.md shows the `extern "C"` block a crate maintainer would add.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary, no torch/HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream);
 *   - every call returns a daac_status; daac_last_error() gives a thread-local message;
 *   - automata are immutable after creation: any number of host threads may scan the same
 *     handle concurrently (each call brings its own stream);
 *   - V = u32 only (pattern values are 32-bit), positions are 64-bit.
 *   - there is NO CPU scan backend in this library: a scan without a usable gfx950 device
 *     returns DAAC_ERR_DEVICE.
 */
#ifndef DAACHORSE_AMD_H
#define DAACHORSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header's ABI: struct layouts and the meaning of enum values.  daac_abi_version() returns the version the
 * loaded library was built with; a binding checks it once (the ctypes and C++ mirrors here do).
 *   3 (round 3): daac_info starts with struct_size and carries the per-request engine plan; DAAC_ENGINE_PFX; daac_match16 /
 *                daac_scan_device16.
 *   4 (round 4): daac_iter_next_batch; the lazy iterator runs its windows ahead of the consumer on a worker thread; DAAC_ENGINE_JUMP /
 *                DAAC_KERNEL_JUMP are gone (the experiment is in the history: commit c51e1c3 and before, tools/experiments/jump).
 *   5 (round 5): daac_scan_count_multi (one haystack sharded across the devices of a node); daac_pma_set_option (options per handle);
 *                daac_pma_trim; options gram4_arith, gram_tail.
 *   6 (round 6): daac_stream_feed_compact; daac_scan_count_multi runs its shards on one persistent worker thread + stream per device;
 *                daac_pma_set_option answers 6 for an option that is read at upload once the handle has tables on a device;
 *                gram_version 3 is an alias of 4; the options emit_stagger, emit_v3_lds and the four names of engines that left the
 *                library (restart_tier, emit_tiles, emit_rec_cap, emit_version) are gone. */
#define DAAC_ABI_VERSION 6
uint32_t daac_abi_version(void);

/* src/errors.rs:10-22 (first four), plus the panics / extras of this boundary */
typedef enum {
    DAAC_OK = 0,
    DAAC_ERR_INVALID_ARGUMENT = 1,  /* DaachorseError::InvalidArgument   */
    DAAC_ERR_AUTOMATON_SCALE = 2,   /* DaachorseError::AutomatonScale    */
    DAAC_ERR_INVALID_CONVERSION = 3,/* DaachorseError::InvalidConversion */
    DAAC_ERR_INVALID_AUTOMATON = 4, /* DaachorseError::InvalidAutomaton  */
    DAAC_ERR_MATCH_KIND = 5,        /* the reference PANICS here: bytewise.rs:194-197, 299-302, 551-554 */
    DAAC_ERR_UNSUPPORTED = 6,       /* an engine that cannot serve the request; or the one input on which the reference itself never terminates (leftmost kind, "" in the set, haystack ends inside a longer pattern: SURVEY §8a note D) */
    DAAC_ERR_DEVICE = 7             /* HIP error / no gfx950 device */
} daac_status;

/* src/lib.rs:324-346 (repr(u8)) */
typedef enum { DAAC_STANDARD = 0, DAAC_LEFTMOST_LONGEST = 1, DAAC_LEFTMOST_FIRST = 2 } daac_match_kind;

/* Which iterator of src/bytewise/iter.rs a scan reproduces */
typedef enum {
    DAAC_FIND_OVERLAPPING = 0,           /* FindOverlappingIterator          iter.rs:117-177 */
    DAAC_FIND = 1,                       /* FindIterator                     iter.rs:44-114  */
    DAAC_LEFTMOST_FIND = 2,              /* LeftmostFindIterator             iter.rs:247-341 */
    DAAC_FIND_OVERLAPPING_NO_SUFFIX = 3  /* FindOverlappingNoSuffixIterator  iter.rs:180-244 */
} daac_scan_mode;

/* Which device engine to use.  AUTO picks GRAM for daac_scan_count[_range](FIND_OVERLAPPING) and TIERED
 * for everything else when the automaton qualifies, DARRAY otherwise. */
typedef enum {
    DAAC_ENGINE_AUTO = 0,
    DAAC_ENGINE_TIERED = 1, /* re-packed bitmap-rank trie, top levels dense in LDS */
    DAAC_ENGINE_DARRAY = 2, /* the reference's own double array, hot/cold split   */
    DAAC_ENGINE_GRAM = 3,   /* k-gram context tables in LDS, no state chain: count / checksum, and tuples of FIND_OVERLAPPING */
    DAAC_ENGINE_PFX = 4     /* hashed prefix filter in LDS + goto-only walks from the start of an occurrence: `.count()`, count + checksum
                             * and (round 4, dictionaries without duplicate patterns) the tuples of FIND_OVERLAPPING for dictionaries over
                             * any byte alphabet (what AUTO takes where GRAM's byte classes run out) */
} daac_engine;

/* Match<u32> (src/lib.rs:286-320): start() = end - length, end(), value() */
typedef struct {
    uint64_t start;
    uint64_t end;
    uint32_t value;
    uint32_t _pad;
} daac_match;

/* The same match as the crate keeps it (Match<u32>, src/lib.rs:287-291: length, end, value; start() = end - length): 16 bytes, what
 * daac_scan_device16 leaves in device memory — one 16-byte store per tuple and a third less write traffic than daac_match. */
typedef struct {
    uint64_t end;
    uint32_t length;
    uint32_t value;
} daac_match16;

/* What a request is served by (daac_info.plan_*): the engine AUTO resolves to, the kernel family behind it, and — where that is not
 * the fastest family — why.  The families differ by an order of magnitude: this is the performance contract of a handle,
 * visible before the first scan. */
typedef enum {
    DAAC_REQ_OVERLAPPING_COUNT = 0,     /* find_overlapping_iter(h).count()        daac_scan_count_only_range           */
    DAAC_REQ_OVERLAPPING_CHECKSUM = 1,  /* count + checksum of that stream         daac_scan_count / _range             */
    DAAC_REQ_OVERLAPPING_TUPLES = 2,    /* the tuples                              daac_scan / daac_scan_device[16] / daac_iter_* */
    DAAC_REQ_FIND = 3,                  /* find_iter, any form                                                            */
    DAAC_REQ_LEFTMOST_FIND = 4,         /* leftmost_find_iter, any form                                                   */
    DAAC_REQ_NO_SUFFIX = 5,             /* find_overlapping_no_suffix_iter, any form                                      */
    DAAC_REQ_N = 6
} daac_request;
typedef enum {
    DAAC_KERNEL_NONE = 0,        /* the request does not apply to this automaton's MatchKind (the crate panics) */
    DAAC_KERNEL_GRAM_COUNT = 1,  /* gram4_kernels.hip: one LDS lookup per byte, lane-local hit masks           (cfg3: 1.4 TB/s) */
    DAAC_KERNEL_GRAM_EXACT = 2,  /* gram_kernels.hip / gram2_kernels.hip with the checksum                      (cfg3: 1.0 TB/s) */
    DAAC_KERNEL_GRAM_WIDE = 3,   /* gram2w_kernels.hip: 31 .. 62 byte classes                                   (1.0 / 0.9 TB/s) */
    DAAC_KERNEL_GRAM_EMIT = 4,   /* emit3_kernels.hip (gram2_emit_kernels.hip behind it): tuples in reference order (cfg3: 0.22 TB/s of haystack) */
    DAAC_KERNEL_PFX = 5,         /* pfx_kernels.hip: any byte alphabet; `.count()` 0.6 - 1.3 TB/s on sparse dictionaries, 0.05 - 0.1 on the
                                  * wide look-alikes; tuples through pfx_emit_kernel + EXPAND: 0.02 - 0.36 TB/s of haystack */
    DAAC_KERNEL_SEGMENT = 6,     /* scan_kernels.hip: one lane per segment, TIERED or DARRAY tables             (0.03 - 0.4 TB/s) */
    DAAC_KERNEL_MICRO = 7,       /* chain_scan.hpp overlap_count_body: micro-step walker over the double array  (0.08 - 0.4 TB/s) */
    DAAC_KERNEL_CHAIN = 8,       /* chain_scan.hpp: speculate / reconcile / emit for the restart iterators      (0.1 - 0.3 TB/s) */
    DAAC_KERNEL_SELECT = 9       /* find3_kernels.hip / left3_kernels.hip: the restart iterators' count (+ checksum) as a selection over the tuple
                                    emitter's detection, no state chain (cfg3, random text: 0.36 / 0.31 TB/s); text made of dictionary words goes
                                    back to DAAC_KERNEL_CHAIN and the plan then says so */
} daac_kernel_family;
typedef enum {
    DAAC_WHY_FASTEST = 0,        /* nothing faster exists for this request */
    DAAC_WHY_NOT_UPLOADED = 1,   /* daac_pma_upload has not run: the plan is not known yet */
    DAAC_WHY_ALPHABET = 2,       /* more distinct pattern bytes than the byte-class tables take (30 / 62) */
    DAAC_WHY_LDS = 3,            /* the tables do not fit the 160 KB of LDS */
    DAAC_WHY_EMPTY_PATTERN = 4,  /* "" is a pattern: every position matches */
    DAAC_WHY_DUPLICATES = 5,     /* duplicate patterns beyond what the tables encode (more than 3 ending after one context; among patterns of at most K bytes for tuples) */
    DAAC_WHY_CHAIN = 6,          /* the iterator is a chain through its own matches: no position-parallel form */
    DAAC_WHY_CHARWISE = 7,       /* a charwise automaton: scanned over its own double array */
    DAAC_WHY_TRIE_SHAPE = 8      /* not a tree-shaped trie / other table limits */
} daac_plan_reason;

typedef struct {
    uint32_t struct_size;     /* IN: sizeof(daac_info) as the caller was compiled (0 = the round-2 layout is NOT assumed: the call fails);
                               * the library fills at most this many bytes */
    uint8_t match_kind;       /* DoubleArrayAhoCorasick::match_kind()  bytewise.rs:735-737 */
    uint32_t num_states;      /* ::num_states()                        bytewise.rs:785-787 */
    uint64_t states_len;      /* double-array elements (multiple of 256) */
    uint64_t outputs_len;
    uint64_t heap_bytes;      /* ::heap_bytes()                        bytewise.rs:764-770 */
    uint32_t max_pattern_len; /* max Output::length — the halo is this minus one */
    /* device-side re-pack (valid after upload) */
    uint32_t num_classes;     /* byte classes incl. class 0 = "byte occurs in no pattern" */
    uint32_t tier_dense_states;  /* states with a dense, fail-resolved LDS row */
    uint32_t tier_lds_states;    /* + states whose child bitmap lives in LDS */
    uint32_t tier_lds_bytes;     /* LDS bytes of the automaton tables per workgroup */
    uint8_t tiered_available;    /* 0 if only the DARRAY engine can run this automaton */
    uint8_t gram_available;      /* the GRAM count engine can run this automaton */
    uint32_t gram_k;             /* context length K of the GRAM tables */
    uint32_t gram_lds_bytes;
    uint8_t charwise;            /* 1: a CharwiseDoubleArrayAhoCorasick (src/charwise.rs), 0: bytewise */
    uint32_t alphabet_size;      /* charwise: number of distinct code points in the patterns (mapper.rs:10-13) */
    uint8_t gram2_available;     /* the GRAM engine's second table set (one LDS lookup per position) serves this automaton */
    uint8_t gram2_exact;         /* ... also with the checksum (CID/H fit next to M) */
    uint32_t gram2_k;
    uint32_t gram2_lds_count;    /* LDS bytes per workgroup, count only / with checksum */
    uint32_t gram2_lds_exact;
    uint8_t gram_wide;           /* 31 .. 62 byte classes: the GRAM engine runs on 64-bit words with K = 2 (gram2w.hpp) */
    uint8_t pfx_available;       /* the PFX tables were built (any byte alphabet: `.count()`, count + checksum, tuples) */
    uint32_t pfx_key_bytes;      /* G: bytes of a PFX filter key */
    uint32_t pfx_lds_bytes;
    /* the engine plan, indexed by daac_request; valid after upload, for engine AUTO */
    uint8_t plan_engine[8];      /* daac_engine reported by daac_last_engine() after such a request */
    uint8_t plan_kernel[8];      /* daac_kernel_family */
    uint8_t plan_reason[8];      /* daac_plan_reason: why not the fastest family */
} daac_info;

typedef struct daac_pma daac_pma;         /* an automaton (host copy + per-device re-pack) */
typedef struct daac_matches daac_matches; /* result of an eager scan */
typedef struct daac_iter daac_iter;       /* lazy iterator façade */

const char *daac_last_error(void);
void daac_free(void *p);
/* daac_engine that served this thread's most recent scan (AUTO resolves to GRAM / TIERED / DARRAY per request:
 * e.g. GRAM declines ranges of 32 GiB and more and automata whose tables do not fit LDS). */
int daac_last_engine(void);
/* The kernel family — and, for the `.count()` kernel, the launch shape the options in force gave it ("gram4 ppl=32 dir=0 waves=16 arith=1 filter=1
 * tail=auto") — that served the calling thread's last daac_scan_count* call (daac_scan_count_multi: shard 0's, as run by its device's worker).
 * ABI 6; a diagnostic: the text is not a contract. */
const char *daac_last_kernel(void);

/* ---- construction / (de)serialisation ------------------------------------------------------ */

/* DoubleArrayAhoCorasick::deserialize (bytewise.rs:868-964): parses the crate's serialize()
 * blob for V = u32 with the same validation, rebuilds root_table (bytewise.rs:1040-1056).
 * The caller keeps `blob`. */
daac_status daac_bytewise_from_serialized(const uint8_t *blob, size_t len, daac_pma **out, size_t *consumed);

/* Takes the automaton's arrays directly (what a Rust shim has in hand without serialising):
 * `states` = n_states x {base, fail, opos_ch} (State<u32>, bytewise.rs:1131-1137) or, for leftmost
 * kinds, `lstates` = n x {base, opos_ch} + `fails` (bytewise.rs:61-63); `outputs` = n_outputs x
 * {value, length, parent} (lib.rs:213-218).  Same validation as deserialize.  `fails` must hold n_lstates entries;
 * a NULL array with a non-zero count is DAAC_ERR_INVALID_ARGUMENT. */
daac_status daac_bytewise_from_parts(const uint32_t *states, size_t n_states,
                                     const uint32_t *lstates, const uint32_t *fails, size_t n_lstates,
                                     const uint32_t *outputs, size_t n_outputs,
                                     uint8_t match_kind, uint32_t num_states, daac_pma **out);

/* DoubleArrayAhoCorasickBuilder::build / build_with_values (bytewise/builder.rs:152-244) on the
 * host CPU.  Patterns are one blob + n+1 offsets; values == NULL means value = index. */
daac_status daac_bytewise_build(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values,
                                size_t n, uint8_t match_kind, uint32_t num_free_blocks, daac_pma **out);

/* ---- charwise automata (src/charwise.rs) --------------------------------------
 * A charwise handle is a daac_pma like any other: daac_pma_*, daac_scan*, daac_iter_* serve it with
 * the iterators of src/charwise/iter.rs (FindIterator :101-157, FindOverlappingIterator :160-221,
 * FindOverlappingNoSuffixIterator :224-303, LeftmostFindIterator :306-400).  Haystacks are UTF-8
 * (the reference takes AsRef<str>); start/end stay BYTE offsets, as in the reference.  Engines:
 * AUTO or DARRAY (the charwise double array runs as is). */

/* CharwiseDoubleArrayAhoCorasick::deserialize (charwise.rs:896-952), blob format of serialize()
 * (charwise.rs:831-848: states x {base, check, fail, output_pos}, mapper table + alphabet_size,
 * outputs, match_kind, num_states), same validation. */
daac_status daac_charwise_from_serialized(const uint8_t *blob, size_t len, daac_pma **out, size_t *consumed);

/* CharwiseDoubleArrayAhoCorasickBuilder::build / build_with_values (charwise/builder.rs:178-239) on the
 * host CPU; patterns are UTF-8 (one blob + n+1 byte offsets).  Byte-identical arrays. */
daac_status daac_charwise_build(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values,
                                size_t n, uint8_t match_kind, uint32_t num_free_blocks, daac_pma **out);

/* ::serialize (bytewise.rs:801-820 / charwise.rs:831-848); free the buffer with daac_free. */
daac_status daac_pma_serialize(const daac_pma *pma, uint8_t **buf, size_t *len);
/* `info->struct_size` must be set by the caller (see daac_info). */
daac_status daac_pma_info(const daac_pma *pma, daac_info *info);
/* One line of text per request of the plan ("find_overlapping_iter().count(): engine gram, kernel gram4 ..."); returns the
 * number of bytes the full text needs (incl. the terminating 0); writes at most `cap`. */
size_t daac_pma_explain(const daac_pma *pma, char *buf, size_t cap);
void daac_pma_free(daac_pma *pma);

/* Re-packs the automaton for the GPU and copies it to `device` (idempotent).  Scans upload
 * lazily to the current device if this was not called. */
daac_status daac_pma_upload(daac_pma *pma, int device);
/* Releases the scratch a handle keeps between calls beside its tables (the tuple emitter's and the selection kernels' workspace, up to
 * option workspace_keep bytes per device): for processes that hold many handles.  Tables stay; the next scan allocates again. */
daac_status daac_pma_trim(daac_pma *pma);

/* ---- scans ----------------------------------------------------------------------------------- */

/* Eager scan of one haystack; replaces driving the iterator to exhaustion
 * (`pma.find_overlapping_iter(h).collect()` etc.).  `hay` is a host pointer, or a device pointer
 * if hay_is_device != 0.  Matches come back in the reference's order. */
daac_status daac_scan(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len,
                      int hay_is_device, void *stream, daac_matches **out);
/* The same with the match list left in DEVICE memory (for consumers that run on the GPU): `*dev_out` receives a device
 * buffer of `*count` daac_match tuples in the reference's order (NULL when there are none), owned by the caller and
 * released with daac_device_free.  For DAAC_FIND_OVERLAPPING on a bytewise Standard automaton whose tables fit (engine
 * AUTO or GRAM) the tuples come from the GRAM emitter: per-tile counts, one exclusive scan, then every tuple is written
 * once, straight to its final place; all other requests run the segment scanners (count, scan, write).  The call returns
 * after the stream has finished.  daac_device_to_host copies (part of) such a list to host memory. */
daac_status daac_scan_device(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len,
                             int hay_is_device, void *stream, daac_match **dev_out, uint64_t *count);
/* The same list as 16-byte tuples {end, length, value} (daac_match16).  The GRAM emitter writes them directly; lists from the other
 * engines are repacked on the device. */
daac_status daac_scan_device16(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len,
                               int hay_is_device, void *stream, daac_match16 **dev_out, uint64_t *count);
void daac_device_free(void *p);
daac_status daac_device_to_host(void *dst, const void *dev_src, size_t bytes);
size_t daac_matches_count(const daac_matches *m);
const daac_match *daac_matches_data(const daac_matches *m); /* host memory, owned by `m` */
void daac_matches_free(daac_matches *m);

/* Count + order-independent checksum of the match stream without materialising it
 * (`.count()` on the iterator).  checksum = (S1 << 32) | S2 with, over all matches,
 *   h = low32(mix64(value << 32 | length)),  S1 = sum h,  S2 = sum h * low32(end)   (mod 2^32).
 * Asynchronous on `stream` when `result_dev` != NULL: the 3 x u64 {count, S1, S2} are left in
 * device memory there and count/checksum may be NULL; otherwise the call synchronises.
 * Device haystacks are read in whole aligned 16-byte granules: up to 15 bytes before `hay` and after `hay + len`
 * (inside the same 16-byte granules as the first / last byte) are loaded and masked out, never interpreted.  Any
 * hipMalloc'ed buffer satisfies this (allocations are 256-byte granular); a sub-range of a larger allocation always does. */
daac_status daac_scan_count(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len,
                            int hay_is_device, void *stream, uint64_t *count, uint64_t *checksum,
                            uint64_t *result_dev);

/* `.count()` alone: the number of matches with end in (begin, len], no checksum.  With `result_dev` the count is left in
 * result_dev[0] (3 x u64 as above; [1] and [2] are not meaningful) and the call is asynchronous.  The GRAM engine serves
 * this with one LDS lookup per haystack byte; every other engine runs its count + checksum scan and drops the checksum. */
daac_status daac_scan_count_only_range(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin,
                                       int hay_is_device, void *stream, uint64_t *count, uint64_t *result_dev);

/* One haystack sharded across the GPUs of a node (BASELINE configs[3]: 8 shards, one per MI355X).  Shard k is `len` bytes that begin at
 * haystack position `base`; `hay` points at the `halo` bytes in front of it followed by the shard itself, in the memory of `device`
 * (or in host memory when hay_is_device = 0).  Every shard but the one at position 0 needs halo >= max_pattern_len - 1 (charwise:
 * max_pattern_len, at least 3): a match is counted by the shard its END falls in, wherever it starts; halo <= base.  One
 * persistent worker thread per DEVICE the shards name (ABI 6: created at the handle's first such call, with a stream of its own — a
 * blocking stream: it sees what the caller left on the device's default stream — and the handle's options in scope) queues
 * daac_scan_count[_only]_range for its device's shards back to back (tables are uploaded there on first use) and waits once; the host adds the counts and — when
 * `checksum` is not NULL — the checksum sums with every shard's ends re-based to haystack positions, so the result equals
 * daac_scan_count of the whole haystack.  The find_overlapping modes only (status 6 otherwise: the other two iterators are chains).
 * Several shards may name the same device.  daac_last_engine() / daac_last_kernel() report shard 0's. */
typedef struct {
    int device;           /* HIP device ordinal */
    const uint8_t *hay;   /* `halo` bytes of the haystack in front of the shard, then the shard */
    size_t halo;          /* bytes in front that are there (the shard at position 0: 0) */
    size_t len;           /* bytes of the shard */
    uint64_t base;        /* haystack position of the shard's first byte */
} daac_shard;
daac_status daac_scan_count_multi(daac_pma *pma, int mode, int engine, const daac_shard *shards, size_t n, int hay_is_device,
                                  uint64_t *count, uint64_t *checksum);

/* The same over the tail of a haystack: counts the matches with end in (begin, len] — what one
 * shard of a haystack split across devices contributes.  Bytes before begin - Lmax are never read (they need
 * not be resident), byte 0 of the haystack is still `hay`.  For the overlapping modes any `begin` works (charwise:
 * also inside a character).  DAAC_FIND / DAAC_LEFTMOST_FIND are chains through their own matches: there `begin`
 * must be a position where the iterator restarts (0, or the end of a match it reported), and the call reports
 * what the iterator reports from there on.  These two modes settle the chain before counting and synchronise
 * the stream even when `result_dev` is given. */
daac_status daac_scan_count_range(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len, size_t begin,
                                  int hay_is_device, void *stream, uint64_t *count, uint64_t *checksum,
                                  uint64_t *result_dev);

/* Lazy façade = Iterator::next() (iter.rs:58, 133, 195, 272): scans the haystack window by
 * window on the device and hands tuples out one at a time.  The haystack must stay alive until
 * close (the Rust iterator owns/borrows `P` the same way). */
daac_status daac_iter_open(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len,
                           int hay_is_device, void *stream, daac_iter **out);
int daac_iter_next(daac_iter *it, daac_match *m); /* 1 = Some(m), 0 = None, <0 = -daac_status */
/* The same stream, a run at a time and without a copy: `*batch` points at the next `*n` >= 1 matches in the iterator's own
 * (page-locked) window buffer, as the crate keeps a Match (lib.rs:287-291: end, length, value); valid until the next call on this
 * iterator.  1 = a run, 0 = exhausted, <0 = -daac_status.  May be mixed with daac_iter_next (both advance the same position).
 * Behind both: a worker thread scans the windows (option iter_window, 64 MiB) ahead of the consumer with three stages in flight
 * on their own streams — host-to-device copy of window k + 1, scan of window k, device-to-host copy of window k - 1's 16-byte tuples. */
int daac_iter_next_batch(daac_iter *it, const daac_match16 **batch, size_t *n);
/* The compact form of the same iterator.  On match-dense text `next()` is bound by the tuples' way back over PCIe (cfg3: 10 GB of
 * 16-byte tuples per GiB of haystack at the link's ~50 GB/s), so an iterator opened with daac_iter_open_compact sends 8 bytes per tuple:
 * the value, and one word that holds the end relative to the window's first byte in its low `end_bits` bits and the length above them
 * (end_bits = 32 - the bits of the longest pattern's length; windows are kept below 2^end_bits bytes).  daac_iter_next_batch8 hands out
 * runs of them:  end = *end_base + (t.end_len & ((1 << *end_bits) - 1)),  length = t.end_len >> *end_bits,  start = end - length.
 * daac_iter_next works on either kind; daac_iter_next_batch only on the 16-byte kind and daac_iter_next_batch8 only on the compact one
 * (status 6 otherwise); daac_iter_open_compact itself answers 6 for dictionaries with patterns of several KB (no room for a window). */
typedef struct daac_match8 {
    uint32_t value;
    uint32_t end_len;
} daac_match8;
daac_status daac_iter_open_compact(daac_pma *pma, int mode, int engine, const uint8_t *hay, size_t len,
                                   int hay_is_device, void *stream, daac_iter **out);
int daac_iter_next_batch8(daac_iter *it, const daac_match8 **batch, size_t *n, uint64_t *end_base, uint32_t *end_bits);
void daac_iter_close(daac_iter *it);

/* Chunk-fed steppers = FindOverlappingStepper / FindStepper (bytewise/iter.rs:344-475, charwise/iter.rs:403-534)
 * and the *_from_iter entry points (bytewise.rs:238-251, 353-375) for haystacks that arrive piecewise: feed the
 * text chunk by chunk (any sizes, cuts may fall inside UTF-8 characters); every call returns the matches it can
 * already decide — for the overlapping modes everything ending inside the chunk, for DAAC_FIND the chain up to
 * the end of the data — in stream coordinates, so that the concatenation over all calls is exactly what the
 * iterator reports on the concatenated text.  The library keeps at most max-pattern-length bytes (DAAC_FIND: the
 * undecided tail) of earlier chunks on the device.  Modes: DAAC_FIND_OVERLAPPING, DAAC_FIND,
 * DAAC_FIND_OVERLAPPING_NO_SUFFIX (the reference has no leftmost stepper: a leftmost match needs lookahead). */
typedef struct daac_stream daac_stream;
daac_status daac_stream_open(daac_pma *pma, int mode, int engine, void *stream, daac_stream **out);
daac_status daac_stream_feed(daac_stream *s, const uint8_t *chunk, size_t len, int chunk_is_device, daac_matches **out);
/* The same feed with the matches as 8-byte tuples (ABI 6; the form daac_iter_next_batch8 hands out): `*batch` points at `*n` tuples in a
 * page-locked block of the stream object, valid until the next feed or close;  end = *end_base + (t.end_len & ((1 << *end_bits) - 1)),
 * length = t.end_len >> *end_bits.  A third of daac_match's bytes over PCIe and nothing to allocate or free per feed.  A chunk (plus the
 * bytes kept of earlier ones) must stay below 2^end_bits bytes (end_bits = 32 - the bits of the longest pattern's length): status 6 otherwise. */
daac_status daac_stream_feed_compact(daac_stream *s, const uint8_t *chunk, size_t len, int chunk_is_device, const daac_match8 **batch,
                                     size_t *n, uint64_t *end_base, uint32_t *end_bits);
void daac_stream_close(daac_stream *s);

/* ---- tuning knobs (optional) ----------------------------------------------------------------- */
/* Process-wide name/value pairs for experiments; defaults in parentheses.  Options that shape the device tables
 * ("lds_budget", "dense_depth", "rows_share_pct", "gram_lds_budget", "gram_rank_in_lds", "char_map_lds") are read
 * when an automaton is uploaded, the others at every scan.
 *   seg_bytes (0 = auto)        bytes of haystack per lane-segment of the segment scanners
 *   threads (1024), blocks_per_cu (0 = auto)   launch shape of the overlapping scanners
 *   lds_budget (98304), dense_depth (-1 = auto), rows_share_pct (45)   TIERED re-pack
 *   gram_lds_budget (161792), gram_region (0 = auto: 16384 for the first table set, 65536 for the second and PFX, 262144 from 2 GiB on; rounded down to a power of two >= 2048), gram_slab (4096), gram_dense (-1 = auto), gram_rank_in_lds (-1 = auto),
 *   gram_ppl (0 = auto: 32 positions per lane and step for automata without short patterns, else 16)
 *   gram_version (0 = auto: `.count()` on the gram4 kernel over the renumbered second table set, count + checksum on the first where it applies;
 *                 1 = first table set only, 2 = gram2 kernels only, 4 (3: its name in ABI 4) = gram4 for `.count()` or an error), gram2_dpp (1: DPP wave shifts),
 *   gram2_rfull (1)             rank directory with one entry per M word when LDS allows (0: one per four words)
 *   gram_tail (alias gram3_tail; -1 = every workgroup samples its text and picks; 0 / 1: the plain / the tail-record body of the gram4 kernel),
 *   gram4_arith (1: byte classes by arithmetic where the dictionary's bytes are one range; 0: the class table in LDS)
 *   gram4_filter (1)            the gram4 kernel's LDS filter in front of rank + gather (gram4_filter.hpp, ABI 6): a hit asks the L2 for its record
 *                               only if a Bloom word says the state may end a pattern or go on; workgroups whose text is made of dictionary
 *                               words keep the per-word directory and the tail-record body.  0: round 5's bodies
 *   find3 (1)                   find_iter's count (+ checksum) by selection over the tuple emitter's per-position flags (find3_kernels.hip;
 *                               Standard bytewise dictionaries with K = 3 tables and no pattern beyond 19 bytes) instead of the chain walkers,
 *                               in windows of find3_window (2^30) end positions, each restarting at the last match of the one before; a handle
 *                               whose last such request met text made of dictionary words goes back to the walkers (2: never), 0: off
 *   left3 (1)                   leftmost_find_iter's count (+ checksum) likewise (left3_kernels.hip): at upload a leftmost handle's patterns are
 *                               read back from its trie and built into a Standard automaton whose detection tables the selection — by match
 *                               STARTS — runs on; same conditions, same windows, same fallback as find3 (2: whatever the text, 0: off — set
 *                               before the upload to save the second build)
 *   select_emit (1)             the tuple LIST of find_iter / leftmost_find_iter (daac_scan, daac_scan_device[16], the lazy iterator's windows) from
 *                               those selection kernels' emitting form, where find3 / left3 apply; 0: the chain walkers' speculate / reconcile / emit
 *   workspace_keep (8 GiB)      bytes of scratch (annotated stream, record list: ~2 per haystack byte) a handle keeps between its tuple-emitter
 *                               and find3 calls instead of asking the pool every time (tools/micro/pool_ops.hip); 0: none
 *   pfx_probe (16384)           AUTO, count (+ checksum) of a dictionary PFX serves: a synchronous scan of a device haystack >= 32 MiB samples
 *                               65 536 positions; where more than this many survive PFX's filter the micro-step walker over the double array
 *                               takes the scan (the verdict stays in the handle for the asynchronous calls); 0 = never ask, always PFX
 *   pfx (1)                     PFX tables (any byte alphabet): 1 = built where no GRAM table set applies, 2 = for every automaton they can
 *                               serve (DAAC_ENGINE_PFX then selects them explicitly), 0 = never; read at upload
 *   emit (1)                    materialising overlapping scans through the tuple emitter (emit3_kernels.hip: detection once, then expansion;
 *                               GRAM tables, or PFX's for any byte alphabet) where it applies (0: segment scanners);
 *   emit_rec_per_kib (32)       deep-match records per KiB of haystack the record list is first sized for (a handle remembers what its
 *                               last scan met; a list that proves too short is sized exactly and the detection is rerun once)
 *   restart_chain (1)           find_iter / leftmost_find_iter as speculate-reconcile-emit chains (0: sync-point scanners)
 *   chain_rounds (24)           reconciliation rounds before falling back to the sync-point scanners
 *   restart_bpc (8)             256-lane workgroups per CU of the chain walkers (they are bound by VALU issue at full occupancy)
 *   char_map_lds (1)            charwise walkers: ASCII and the populated stretch of the code mapper staged in LDS when they fit 32 KB
 *   char_row_lds (1)            ... and ROOT's row of children beside it when both fit 80 KB (read at upload)
 *   overlap_micro (1)           count (+ checksum) of overlapping scans the GRAM tables do not serve: 1 = the micro-step walker for
 *                               charwise automata and the double array, 2 = also in place of the TIERED engine, 0 = the segment scanners
 *   pool (1), pool_keep (0)     scratch and result buffers from the device's stream-ordered pool, which keeps up to pool_keep bytes
 *                               between calls (0 = 1/8 of the device memory, at most 32 GiB); read at the first scan of the process
 *   iter_window (64 MiB)        haystack bytes per window of the lazy iterator (the first windows are 16 and 32 MiB: matches arrive early)
 *   max_result_bytes (8 GiB)    largest match list daac_scan may materialise */
daac_status daac_set_option(const char *name, int64_t value);
/* The same option for ONE handle (ABI 5): overrides the process-wide value for every scan, iterator and stream of `pma`, whichever thread runs
 * them (the worker threads of the lazy iterator and of daac_scan_count_multi included).  Two threads scanning two handles with different
 * settings do not share state.  unset != 0 removes the override.  The options read when the tables are laid out (lds_budget, dense_depth,
 * rows_share_pct, gram_lds_budget, gram_rank_in_lds, pfx, char_map_lds, char_row_lds) are set BEFORE the handle's first upload: on a handle
 * that already has tables on a device they could not take effect, and the call says so (status 6, ABI 6) instead of answering OK.
 * `pool` / `pool_keep` belong to the device's allocator, not to a handle (status 1). */
daac_status daac_pma_set_option(daac_pma *pma, const char *name, int64_t value, int unset);

#ifdef __cplusplus
}
#endif
#endif
