// Kernel argument structs shared by the HIP kernels and the host driver.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/daachorse_amd.h"

namespace daac {

// TIERED engine tables (see repack.hpp).  All pointers are device pointers; the first five are
// staged into LDS by every workgroup at kernel start, at the byte offsets given below.
struct TierDev {
    const void *rows;        // NA x C entries, u16 (id | flag << 15) or u32 (id | flag << 31)
    const uint32_t *bcmap;   // NB - NA child bitmaps
    const uint32_t *bfail;   // NB - NA failure links (new ids)
    const uint2 *ssum;       // N x {output count, sum of h32}; [0, NA) also staged in LDS
    const uint8_t *cls;      // 256 byte -> class
    const uint4 *grec;       // N x {cmap, omap, first_child, fail}
    const uint32_t *sopos;   // N output_pos per state (1-based, 0 = none)
    const uint32_t *outputs; // n_outputs x {value, length, parent}
    const uint32_t *ohash;   // n_outputs x h32 of the record's own match (checksum definition)
    uint32_t C, NA, NB, N;
    uint32_t off_bcmap, off_bfail, off_ssum, off_cls, lds_bytes;  // rows are at offset 0
    uint32_t row32, root_flag;
};

// DARRAY engine tables: the reference's double array, hot/cold split.
struct DArrayDev {
    const uint2 *hot;        // {base, opos_ch} per slot
    const uint32_t *fail;    // per slot (the automaton's own links; leftmost kinds: 1 = DEAD)
    const uint32_t *fail_plain;  // classic links of the same trie (== fail for Standard automata)
    const uint4 *rec;            // per slot {base, opos_ch, fail, child filter}: the chain walkers' one-request record (repack.hpp, fmap)
    const uint4 *root_chain;     // 256 x {child, child.base, child.output_pos << 8 | child.fail, child's filter}: ROOT's row for the chain walkers (LDS)
    const uint4 *root;       // 256 x {child, child.base, child.opos_ch, 0}, staged into LDS
    const uint2 *osum;       // per output record {chain count, chain sum of h32}
    const uint32_t *outputs; // n_outputs x {value, length, parent}
    const uint32_t *ohash;   // n_outputs x h32 of the record's own match
    uint32_t n, root_flag, leftmost;
};

// CHARWISE tables: the reference's charwise double array as is (charwise.rs:1096-1101) + the code mapper.
struct CharDev {
    const uint4 *states;         // {base, check = parent slot, fail, output_pos} per slot
    const uint32_t *fail_plain;  // leftmost kinds: classic links of the same trie; Standard: null (states[].z is classic)
    const uint32_t *table;       // code point -> code, 0xffffffff = not in any pattern (mapper.rs:36-42)
    const uint2 *osum;           // per output record {chain count, chain sum of h32}
    const uint32_t *outputs;     // n_outputs x {value, length, parent}
    const uint32_t *ohash;       // n_outputs x h32 of the record's own match
    uint32_t table_len, n, root_flag, leftmost;
    uint32_t map_in_lds;         // the populated stretch of the mapper fits LDS as u16 codes ((table_len - map_lo) * 2 <= 32 KB)
    uint32_t map_lo;             // first code point worth staging (the dense tail of the table starts here)
    const uint2 *root_row;       // per code: {child.base | child.fail << 30 (0 ROOT, 1 DEAD; 2: ROOT has no such child), output_pos | filter << obits}
    uint32_t alphabet;           // number of codes
    uint32_t row_in_lds;         // ROOT's row fits LDS beside the mapper (and every entry can be packed)
    // the chain walkers' copy of `states`: word 3 = output_pos | child filter << obits.  Bit (code & (fbits - 1)) of the filter is set iff
    // the state has a child on a code with those low bits: a probe the filter rules out is a miss without a memory request (and the
    // failure link's record is asked for in its place).  fbits = 16 / 8 with fewer than 2^16 / 2^24 output records, else 0 (no filter;
    // obits = 0 then and the word is the output_pos alone).  root_row[].y is packed the same way, root_row[].x = child.base | fail << 30.
    const uint4 *wstates;
    uint32_t obits, fbits;
};

struct ScanArgs {
    const uint8_t *hay;  // address of haystack byte 0 (only bytes >= begin - halo are read)
    uint64_t begin;      // scan range is [begin, len): matches with end in (begin, len] are reported
    uint64_t len;
    uint64_t seg_bytes;  // bytes per lane-segment
    uint64_t nseg;
    uint32_t halo;       // max pattern length - 1
    unsigned long long *result;      // MODE 0: {count, S1, S2}
    unsigned long long *seg_counts;  // MODE 1 out / MODE 2 in (exclusive offsets)
    daac_match *out;                 // MODE 2
    uint64_t total_len;              // restart scanners: real end of the haystack (a.len is the nominal end of this window)
    unsigned long long *flags;       // optional, 1 x u64: bit 0 = the reference would not terminate on this input
};

struct ScanArgs;
struct ChainArgs;

// Passes of the restart scanners in their speculate / reconcile / emit form (chain_scan.hpp)
struct ChainArgs {
    const unsigned long long *x_spec;  // per segment: exit of the speculative chain
    const unsigned long long *x_prev;  // per segment: exit as of the previous round
    unsigned long long *x_out;         // per segment: exit computed by this pass
    unsigned int *flags;               // [0] = some exit changed this round, [1] = a link overflowed
    uint4 *tally_spec;                 // per segment {count lo, count hi, S1, S2} of the speculative chain
    uint4 *tally_delta;                // per segment: what the true chain reports more (mod 2^64 / 2^32) than the speculative one
    uint64_t cap;                      // bytes a link may run past its segment without match or ROOT
};

// GRAM engine tables (see gram.hpp).  Everything up to `drec` is staged into LDS.
struct GramDev {
    const uint32_t *cls32;    // 256 byte classes, one u32 each (a byte select + shift addresses them in one instruction)
    const uint16_t *cid;      // C^K: K-gram -> combination id (0 = no pattern ends here)
    const uint2 *combo;       // per id {count, hsum}
    const uint32_t *bbits;    // (K+1)-gram trie-prefix bitmap
    const uint8_t *brank;     // per word: set bits before it inside its 8-word superblock
    const uint32_t *bsuper;   // per 8 words: set bits before the superblock
    const uint4 *drec;        // N x {cmap, first_child, own_cnt, own_hsum}  (HBM / L2)
    const uint2 *dhit;        // depth-(K+1) states by rank: {cmap, own_hsum}          (HBM / L2)
    const uint4 *dhit4;       // the same with the first child: {cmap, own_hsum, first_child, 0} (what the kernel reads)
    const uint32_t *cfirst;   // depth-(K+1) states by rank: id of the first child     (HBM / L2)
    uint32_t off_cid, off_combo, off_bbits, off_brank, off_bsuper, off_scratch, lds_bytes;  // cls32 at 0, bbits at 1024
    uint32_t K, C, CC, CCC;
    uint32_t level_start, unused_byte, has_short;
    uint32_t rank_in_lds;     // brank/bsuper staged in LDS (else read from L2 on hits)
    uint32_t n_deep;          // number of depth-(K+1) states = set bits of B
};

struct GramArgs {
    const uint8_t *hay_al;   // haystack address rounded down to 16 bytes
    uint32_t lead;           // bytes between hay_al and haystack byte 0 (0..15)
    uint64_t vlen;           // lead + haystack length ("virtual" positions count from hay_al)
    uint64_t region_bytes;   // contiguous bytes a wave takes at a time (multiple of 1024)
    uint64_t nregions;
    unsigned long long *result;  // {count, S1, S2}
    uint2 *wq;                   // per-wave walker slabs: {position lo, state | class after next << 27}
    uint32_t wq_slab;            // entries per wave
    uint32_t ppl;                // positions per lane and step: 16 or 32 (region_bytes is a multiple of 64 * ppl)
    uint32_t dense;              // B hits are frequent: queue them position by position without testing the group first
    const uint32_t *sel;         // (unused)
    uint32_t sel_want;           // gram3: 0 = plain records, 1 = tail records from the hit record on, 2 = decided by the kernel's density probe
};

hipError_t launch_gram_scan(const GramDev &dev, const GramArgs &a, uint32_t blocks, uint32_t threads, hipStream_t stream);

// GRAM engine, second table set (see gram2.hpp).  LDS layout, count only: [0,256) classes | M at 512 | S | hit rings;
// with checksum: [0,256) classes | H at 512 | M | S | CID | hit rings.
constexpr uint32_t kGram2OffH = 512;
struct Gram2Dev {
    const uint8_t *cls;       // 256 byte classes
    const uint32_t *m;        // C^K words: continuation bits 1..29, short-pattern count in bits 30-31
    const void *sdir;         // per 4 words of m: set continuation bits before the group (u16 or u32 entries)
    const uint16_t *rfull;    // per word of m: set continuation bits before it (only when there are fewer than 65536)
    const uint16_t *cid4;     // C^K: LDS address of the context's H entry (kGram2OffH + 4 * id)
    const uint32_t *hsum;     // per id: sum of h32
    const uint4 *drec;        // N x {cmap, first_child, own_cnt, own_hsum}  (HBM / L2)
    const uint2 *dhit;        // depth-(K+1) states by rank: {cmap, own_hsum}
    const uint4 *dhit4;       // for count + checksum: {cmap, own_hsum, first_child, 0}
    const uint32_t *cfirst;   // depth-(K+1) states by rank: id of the first child
    uint32_t m_bytes, s_bytes, cid_bytes, h_bytes;  // multiples of 16
    uint32_t off_m_count, off_s_count, off_ring_count, lds_count;
    uint32_t rfull_bytes, off_ring_rfull, lds_rfull, rfull_ok;  // count only, with the per-word directory in place of `sdir`
    uint32_t off_m_exact, off_s_exact, off_cid, off_ring_exact, lds_exact;
    uint32_t K, C, s16, unused_byte, n_deep, exact_ok;
    uint32_t xlane_dpp;       // neighbour exchange through DPP wave shifts (else ds_bpermute)
};
uint32_t gram2_lds_bytes(const Gram2Dev &dev, bool exact);

// GRAM engine for 31 .. 62 byte classes (gram2w.hpp): K = 2, 64-bit M words and child bitmaps.  LDS layout as Gram2Dev.
struct Gram2WDev {
    const uint8_t *cls;
    const unsigned long long *m;  // C^2 words: continuation bits 1..61, short-pattern count in bits 62-63
    const uint32_t *sdir;         // per 4 words of m: set continuation bits before the group
    const uint16_t *cid4;         // C^2: LDS address of the context's H entry
    const uint32_t *hsum;
    const uint4 *drec;            // N x {cmap lo, cmap hi, first_child, own_hsum}
    const uint4 *dhit;            // depth-3 states by rank: {cmap lo, cmap hi, own_hsum, first_child}
    uint32_t m_bytes, s_bytes, cid_bytes, h_bytes;
    uint32_t off_m_count, off_s_count, off_ring_count, lds_count;
    uint32_t off_m_exact, off_s_exact, off_cid, off_ring_exact, lds_exact;
    uint32_t C, unused_byte, n_deep, exact_ok;
};
hipError_t launch_gram2w_scan(const Gram2WDev &dev, const GramArgs &a, bool exact, uint32_t blocks, hipStream_t stream);

// GRAM engine, tuple emission (gram2_emit_kernels.hip).  LDS: [0,256) classes | ME at 512 | S | V1 | V2 | hit rings | per-wave areas
constexpr uint32_t kGram2OffM = 512;
struct Gram2EmitDev {
    const uint8_t *cls;
    const uint32_t *me;       // C^K words: continuation bits 1..28, bit 28 + len = a pattern of length len ends after the context
    const void *sdir;         // rank directory (as Gram2Dev::sdir)
    const uint32_t *v1, *v2;  // values of the 1- and 2-gram patterns (staged in LDS)
    const uint32_t *v3;       // values of the 3-gram patterns (L2)
    const uint4 *erec;        // N x {cmap | own (bit 0), first_child, own_value, depth | further copies << 24}
    const uint2 *ehit;        // depth-(K+1) states by rank: {cmap | own, own_value}
    const uint4 *ehit4;       // the same with the first child: {cmap | own, own_value, first_child, further copies << 24} (what the kernel reads)
    const uint32_t *dupo, *dupv;  // per state: where in dupv the values of the further copies of a duplicate pattern start
    uint32_t level_start;     // id of the first depth-(K+1) state
    const uint32_t *cfirst;
    uint32_t m_bytes, s_bytes, v1_bytes, v2_bytes;
    uint32_t off_s, off_v1, off_v2, off_ring, off_wave, lds_bytes;
    uint32_t K, C, s16, unused_byte;
    // emit3 EXPAND: the values of the 3-byte patterns as a rank structure small enough for LDS (a membership bit per 3-gram, a popcount
    // directory per 32 bits, the values in rank order) — the 4 C^3-byte table v3 lives in L2, and a gather per 3-byte match ran at the
    // L2's ~120 G requests/s (profiles/r04_emit3_experiments.txt).  Null when K = 2 or the structure is too large.
    const uint32_t *v3c;      // [bitmap words | directory (u16 per word, padded to 4 bytes) | values]
    uint32_t v3c_bytes, v3c_dir, v3c_val;   // size of the whole (multiple of 16), byte offsets of directory and values
};
hipError_t launch_gram2_scan(const Gram2Dev &dev, const GramArgs &a, bool exact, uint32_t blocks, uint32_t threads, hipStream_t stream);

// LDS plan of the tuple emitter's DETECT kernel (emit3_kernels.hip): per-wave text slots and hit queue behind the tables.
struct Gram3Lds {
    uint32_t off_s;        // rank directory (M sits at 256)
    uint32_t off_wave;     // first wave's area: two text slots, then the hit queue
    uint32_t wave_stride;
    uint32_t lds_bytes, threads, rfull;
};

// `.count()` of round 5 (gram4_kernels.hip); tables of gram4.hpp ("no pattern" is the LAST class, so that byte classes of a
// dictionary over one byte range are min(byte - lo, C - 1): no class table in LDS).
constexpr uint32_t kGram4EndsBitDev = 30;  // hit records: the depth-(K+1) state ends a pattern (= kGram4EndsBit of gram4.hpp)
struct Gram4Dev {
    const uint8_t *cls;       // 256 byte classes (staged only when the classes are not arithmetic)
    const uint32_t *m;        // C^K words: continuation bits 0 .. C-2, short-pattern count in bits 30-31
    const uint16_t *rfull;    // per word of m: set continuation bits before it (null with 65536 depth-(K+1) states or more)
    const void *sdir;         // per 4 words of m (u16 or u32 entries)
    const uint2 *dhit_c;      // depth-(K+1) states by rank: {cmap | ends-a-pattern << 30, first_child}
    const uint4 *dhit_t;      // the same as 16-byte records, single paths below a hit folded into a tail record (bit 31)
    const uint4 *drec_c;      // walk records {cmap, first_child, own_cnt, 0}, tail records from depth K + 3 on
    const uint4 *drec_t;      // ... from depth K + 2 on
    const uint32_t *bloom;    // the filter in front of rank + gather (gram4_filter.hpp), null when it was not built
    uint32_t bloom_words;     // a multiple of 4
    uint32_t m_bytes, rfull_bytes, s_bytes;  // multiples of 16
    uint32_t K, C, s16, arith, lo, unused_byte, n_deep;
};
struct Gram4Lds {
    uint32_t off_wave;     // first wave's area (0): two text slots, then the hit queue
    uint32_t wave_stride;
    uint32_t off_cls;      // 256-byte class table (not staged when `arith`)
    uint32_t off_s;        // rank directory
    uint32_t off_m;        // M
    uint32_t s_bytes;      // bytes of the directory staged
    uint32_t lds_bytes, threads;
    uint32_t arith;        // classes by min(byte - lo, C - 1)
    uint32_t dir;          // 0: u16 per word, 1: u16 per four words, 2: u32 per four words
    uint32_t filter;       // the workgroups whose text is not made of dictionary words stage [coarse directory | Bloom array] at off_s instead and
    uint32_t off_b;        // ... run the body with the filter: off_b = where the Bloom array then lies, s_bytes_f = bytes of the coarse directory
    uint32_t s_bytes_f;
};
// the LDS the filter's Bloom array may take at the preferred launch shape (32 positions per lane, 16 waves, coarse directory): what
// build_gram4_filter is given at upload
uint32_t gram4_filter_room(uint32_t m_bytes, uint32_t sdir_bytes, bool arith, uint32_t lds_limit);
bool gram4_plan(const Gram4Dev &dev, uint32_t ppl, uint32_t waves, bool rfull, bool want_arith, bool want_filter, uint32_t lds_limit, Gram4Lds &L);
hipError_t launch_gram4_scan(const Gram4Dev &dev, const GramArgs &a, const Gram4Lds &L, uint32_t blocks, hipStream_t stream);

// GRAM tuple emission with detection done ONCE (emit3_kernels.hip): DETECT leaves, per haystack byte, one "annotated class"
// byte (class | which short patterns end here << 5), per tile of 1024 positions the number of short tuples, and every deep match
// (longer than K bytes) as a record in a chunked list; the records are binned by tile, the tile counts scanned, and EXPAND turns
// stream + bins into tuples — lanes on CONSECUTIVE positions, so that one store instruction fills neighbouring slots.
constexpr uint32_t kEmit3Tile = 1024;        // positions per tile (one wave-step of EXPAND; half a wave-step of DETECT)
constexpr uint32_t kEmit3Chunk = 1024;       // records per chunk of the list (a wave owns one open chunk at a time)
constexpr uint32_t kEmit3MaxExtras = 64;     // records per tile the per-position length bits cannot carry (as gram2_emit_kernels.hip)
constexpr uint32_t kEmit3Stage = 960;        // EXPAND: staged tuples per wave and pass (a tile of more tuples takes several passes)
// EXPAND, per wave: staged tuples + one dump entry per lane | length bits | first slot per lane | flag bytes per lane | extras | counter
constexpr uint32_t kEmit3ExpandWave = (kEmit3Stage + 64) * 8 + 2048 + 256 + 1024 + kEmit3MaxExtras * 16 + 16;
// ... of the 16-byte format: u16 entries + the tile's stream bytes | length bits | first slot per lane | flag bytes per lane | extras | counter
constexpr uint32_t kEmit3ExpandWave16 = (kEmit3Stage + 64) * 2 + 1040 + 2048 + 256 + 1024 + kEmit3MaxExtras * 16 + 16;
struct Emit3Args {
    const uint8_t *hay_al;        // window address rounded down to 16 bytes ("virtual" positions count from here)
    uint32_t lead;                // bytes between hay_al and the first byte of the window
    uint32_t vlen;                // lead + window length
    uint32_t emit_from;           // matches whose last byte lies at a virtual position >= this are reported
    uint8_t *ann;                 // out: one byte per virtual position (whole wave-steps: nsteps * 2048 bytes)
    uint32_t *tile_short;         // out, per tile of this window: tuples of patterns of at most K bytes
    uint32_t *tile_deep;          // out (atomic adds, zeroed by the caller), per tile of this window: deep matches that end in the tile
    uint32_t tile0;               // global number of this window's tile 0 (goes into the records)
    uint4 *recs;                  // record list: chunk_cap chunks of kEmit3Chunk x {virtual position of the last byte, length | copy << 24, value, global tile}
    uint32_t *chunk_fill;         // per chunk: records in it (zeroed by the caller)
    uint32_t *chunk_next;         // next free chunk (may run past chunk_cap: the caller then knows how many a rerun needs)
    uint32_t chunk_cap;
    uint2 *wq;                    // per-wave walker slabs
    uint32_t wq_slab;
    uint32_t region_bytes;        // contiguous bytes a wave takes at a time (multiple of 2048)
    uint32_t nregions;
    unsigned int *fail;           // bit 1: a wave logged more records between two checkpoints than a chunk holds (-> other engines)
};
struct Expand3Args {
    const uint8_t *ann;
    uint32_t ntiles;
    const unsigned long long *tile_off;  // this window's tiles: exclusive tuple offsets ([ntiles] is valid)
    const unsigned long long *bin_off;   // this window's tiles: exclusive record offsets ([ntiles] is valid)
    const uint4 *binned;                 // the records, grouped by global tile
    void *out;                           // daac_match (24 bytes) or {end u64, length u32, value u32} (16 bytes) tuples
    unsigned long long pos_base;         // end (haystack coordinates) of a match whose last byte is at virtual position v = pos_base + v
    uint32_t off_wave;                   // LDS: V1 at 0, V2 behind it, the per-wave areas from here
    uint32_t has_len1;                   // the dictionary has one-byte patterns
    uint32_t v3_in_lds;                  // the rank structure of the 3-byte patterns' values is staged behind V2 (else they are read from L2)
    unsigned int *fail;                  // bit 2: more extras in one tile than EXPAND places; bit 3: a slot outside its tile (a bug)
    uint32_t vlen, emit_from;            // RAW (PFX) only: `ann` is the haystack; positions in [emit_from, vlen) can end a one-byte match
};
bool emit3_plan(const Gram2EmitDev &dev, uint32_t waves, uint32_t lds_limit, Gram3Lds &L);
hipError_t launch_emit3_detect(const Gram2EmitDev &dev, const Emit3Args &a, const Gram3Lds &L, uint32_t blocks, hipStream_t stream);
hipError_t launch_emit3_combine(const uint32_t *tile_short, const uint32_t *tile_deep, unsigned long long *total, unsigned long long *deep, uint64_t n,
                                hipStream_t stream);
hipError_t launch_emit3_bin(const uint4 *recs, const uint32_t *chunk_fill, const uint32_t *chunk_next, uint32_t chunk_cap, const unsigned long long *bin_off,
                            uint32_t *cursor, uint4 *binned, uint32_t n1k, unsigned long long rec_limit, uint32_t blocks, hipStream_t stream);
hipError_t launch_emit3_expand(const Gram2EmitDev &dev, const Expand3Args &a, bool f16, uint32_t blocks, hipStream_t stream);
hipError_t launch_emit3_expand_raw(const Gram2EmitDev &dev, const Expand3Args &a, bool f16, uint32_t blocks, hipStream_t stream);
uint32_t emit3_expand_lds_bytes(const Gram2EmitDev &dev, uint32_t waves, bool f16, bool v3_in_lds);

// PFX engine (pfx.hpp): `.count()` for bytewise automata over any byte alphabet.  LDS: BLOOM at 0 | DISP | CNT1 | per-wave areas
// find_iter without a state chain (find3_kernels.hip): SELECT over the emitter's annotated stream and binned records.
constexpr uint32_t kFind3Tile = 2048;                  // positions a wave takes at a time (32 per lane)
constexpr uint32_t kFind3Deep = 255;                   // deep selections a tile's list holds
constexpr uint32_t kLeft3Wave = 2080 + 4096 + 1024 + 256 + 288;   // left3_kernels.hip: 16 more stream bytes (the tile after), 65 words of cover bits
constexpr uint32_t kFind3Wave = 2064 + 4096 + 1024 + 256;   // per wave in LDS: 16 + 2048 stream bytes | 2048 x u16 length bits (then the staged positions) | the list of deep selections | per lane: positions with deep matches
struct Find3Dev {
    const uint32_t *h1, *h2;  // h32 of the pattern that IS the 1- / 2-gram of classes (layout of Gram2EmitDev::v1 / v2)
    const uint32_t *h3c;      // h32 of the 3-byte patterns as a rank structure (layout of Gram2EmitDev::v3c)
    uint32_t h1_bytes, h2_bytes, h3c_bytes, h3c_dir, h3c_val, C;
};
struct Find3Args {
    const uint8_t *ann;                  // the annotated stream of DETECT (one window)
    uint32_t ntiles;                     // tiles of kFind3Tile positions
    uint32_t n1k;                        // tiles of 1024 positions (bin_off has n1k + 1 entries)
    const unsigned long long *bin_off;   // exclusive record offsets per tile of 1024 positions
    const uint4 *binned;                 // the deep matches, grouped by that tile
    const uint32_t *entry_in;            // per tile: its last word of the pass before (a tile enters with the word of the tile before it); null: first pass
    uint32_t *exit_out;                  // per tile: its last word of this pass
    uint32_t force_pos;                  // virtual position just before the first byte that counts (the restart point); 0xffffffff: before position 0
    unsigned long long pos_base;         // end (haystack coordinates) of a match whose last byte is at virtual position v = pos_base + v
    uint32_t off_wave;                   // LDS: the h tables at 0 (tallying passes), the per-wave areas from here
    uint32_t count_only;                 // `.count()`: the selections are counted, their h not looked up
    unsigned long long *result;          // {count, S1, S2} (tallying passes; zeroed by the caller)
    unsigned long long *tile_cnt;        // tallying passes, optional: per tile the number of selected matches
    const unsigned long long *tile_off;  // emitting pass: per tile the index of its first tuple (exclusive scan of tile_cnt)
    void *out;                           // emitting pass: the list
    uint32_t f16;                        // emitting pass: daac_match16 (else daac_match)
    uint32_t first_start;                // left3: virtual position of the first START that counts (the restart point)
    uint32_t last_start;                 // left3: virtual position of the first start that no longer counts (the next window's)
    uint32_t *last_sel;                  // max over the last two tiles of (virtual position of a selection, the restart point included) + 1
    unsigned int *flag;                  // bit 0: some tile's last word differs from the pass before (one more pass); bit 1: a match this engine
                                         // cannot place (longer than 19 bytes, a duplicate's copy); bit 2: a tile that would not settle
    // launched behind DETECT without the host having looked: nothing is done when DETECT's record list overflowed (ctl[0] > chunk_cap),
    // its queue failed (ctl[1] != 0) or the records are more than rec_limit (text for the chain walkers)
    const uint32_t *ctl;
    uint32_t chunk_cap;
    unsigned long long rec_limit;
};
__device__ __forceinline__ bool find3_detect_usable(const Find3Args &a) {
    return a.ctl[0] <= a.chunk_cap && a.ctl[1] == 0u && a.bin_off[a.n1k] <= a.rec_limit;
}
uint32_t find3_lds_bytes(const Find3Dev &dev, bool tally);
hipError_t launch_find3_select(const Find3Dev &dev, const Find3Args &a, bool has_len1, bool tally, uint32_t blocks, hipStream_t stream);
hipError_t launch_find3_emit(const Find3Dev &dev, const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream);
hipError_t launch_left3_emit(const Find3Dev &dev, const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream);
hipError_t launch_find3_tail(const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream);
// leftmost_find_iter likewise (left3_kernels.hip): exit_out / entry_in hold, per tile, how many positions of the NEXT tile lie under its last match
uint32_t left3_lds_bytes(const Find3Dev &dev, bool tally);
hipError_t launch_left3_tail(const Find3Args &a, bool has_len1, uint32_t blocks, hipStream_t stream);
hipError_t launch_left3_select(const Find3Dev &dev, const Find3Args &a, bool has_len1, bool tally, uint32_t blocks, hipStream_t stream);

struct PfxDev {
    const uint32_t *bloom;   // bloom_words words
    const uint16_t *cnt1;    // 256: patterns that are this one byte
    const uint16_t *disp;    // one displacement per bucket of the perfect hash
    const uint4 *slots;      // n_slots x {key bytes 0-3, key bytes 4-5 | flags, BASE of the depth-G state, patterns that are the key} or a tail record (pfx.hpp)
    const uint2 *wrec;       // per double-array slot {BASE, CHECK | patterns that end in this state << 8}
    const uint4 *slots_x;    // count + checksum: {key 0-3, key 4-5 | own << 16, BASE, sum of h32 of the patterns that are the key}
    const uint4 *wrec_x;     // count + checksum: {BASE, CHECK | own << 8, sum of h32 of the patterns that end there, 0}
    const uint32_t *hs1;     // 256: sum of h32 of the one-byte patterns (staged behind CNT1)
    const uint4 *slots_e;    // tuple emission: {key 0-3, key 4-5 | own << 16, BASE, VALUE of the pattern that is the key}; wrec_x[.].w = the value of a state's own pattern
    uint32_t G, has_len1, bloom_words, buckets, n_slots, seed;
    uint32_t bloom_bytes, disp_bytes;                       // multiples of 16
    uint32_t off_disp, off_cnt1, off_wave, wave_stride, lds_bytes, threads;   // pfx_plan
    uint32_t n_keys;
};
bool pfx_plan(PfxDev &d, uint32_t lds_limit);
hipError_t launch_pfx_scan(const PfxDev &dev, const GramArgs &a, bool exact, uint32_t blocks, hipStream_t stream);
hipError_t launch_pfx_probe(const PfxDev &dev, const uint8_t *hay, uint64_t len, unsigned int *out, hipStream_t stream);
struct Emit3Args;
hipError_t launch_pfx_emit_detect(const PfxDev &dev, const Emit3Args &a, uint32_t blocks, hipStream_t stream);

hipError_t launch_overlap_count(const DArrayDev &dev, const ScanArgs &a, bool heads, uint32_t blocks, hipStream_t stream);
hipError_t launch_char_overlap_count(const CharDev &dev, const ScanArgs &a, bool heads, uint32_t blocks, hipStream_t stream);
hipError_t launch_tier_scan(const TierDev &dev, const ScanArgs &a, int mode, bool heads, uint32_t blocks, uint32_t threads,
                            hipStream_t stream);
hipError_t launch_darray_scan(const DArrayDev &dev, const ScanArgs &a, int mode, bool heads, uint32_t blocks, uint32_t threads,
                              hipStream_t stream);
// Restart scanners (find_iter / leftmost_find_iter): kmode 0 = totals, 1 = per-segment counts, 2 = write.
// `next_begin` (device, 1 x u64) receives the first sync point >= a.len's nominal end.
hipError_t launch_restart_scan(const DArrayDev &dev, const ScanArgs &a, int kmode, bool leftmost, unsigned long long *next_begin,
                               uint32_t blocks, uint32_t threads, hipStream_t stream);
// Charwise scans: lanes decode UTF-8 on the fly; positions and lengths stay byte offsets (charwise/iter.rs).
hipError_t launch_char_scan(const CharDev &dev, const ScanArgs &a, int mode, bool heads, uint32_t blocks, uint32_t threads,
                            hipStream_t stream);
hipError_t launch_char_restart_scan(const CharDev &dev, const ScanArgs &a, int kmode, bool leftmost, unsigned long long *next_begin,
                                    uint32_t blocks, uint32_t threads, hipStream_t stream);
// pass 0 = speculate, 1 = reconcile (one round), 2 = emit in `kmode` (2 = write), 3 = sum up the tallies in `kmode`
// (0 totals, 1 per-segment counts)
hipError_t launch_chain(const DArrayDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, bool leftmost,
                        unsigned long long *next_begin, uint32_t blocks, hipStream_t stream);
hipError_t launch_char_chain(const CharDev &dev, const ScanArgs &a, const ChainArgs &c, int pass, int kmode, bool leftmost,
                             unsigned long long *next_begin, uint32_t blocks, hipStream_t stream);
hipError_t launch_exclusive_scan(unsigned long long *v, uint64_t n, unsigned long long *total, unsigned long long *scratch, hipStream_t stream);
inline uint64_t exclusive_scan_scratch(uint64_t n) { return n / 2048 + 2; }  // values of scratch the scan of n counts wants

}  // namespace daac
