/*
 * daac_oracle.h — CPU ORACLE for the daachorse bytewise scan path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load this library; the product library
 * (daachorse_amd/csrc) never links, includes or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference crate (daachorse 4.0.0, /root/reference),
 * written by reading the Rust source; every function cites the file:line it follows.
 * The Rust crate itself cannot be compiled here (no rustc/cargo in the image), so the
 * oracle is pinned against the reference's own golden vectors instead
 * (tests/golden/, transcribed from tests/aho_corasick_crate_test.rs:63-382 and the
 * in-module layout pins of src/bytewise.rs:1243-1508) — see tests/test_oracle_*.py.
 */
#ifndef DAAC_ORACLE_H
#define DAAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:324-346 */
enum { ORC_STANDARD = 0, ORC_LEFTMOST_LONGEST = 1, ORC_LEFTMOST_FIRST = 2 };

/* status codes (mirror src/errors.rs:10-22, plus the panic cases) */
enum {
    ORC_OK = 0,
    ORC_ERR_INVALID_ARGUMENT = 1,
    ORC_ERR_AUTOMATON_SCALE = 2,
    ORC_ERR_INVALID_CONVERSION = 3,
    ORC_ERR_INVALID_AUTOMATON = 4,
    ORC_ERR_MATCH_KIND = 5, /* the reference panics: bytewise.rs:194-197, 299-302, 551-554 */
    ORC_ERR_DIVERGED = 6    /* reference iterator does not terminate (SURVEY §8a note D) */
};

/* State<u32>: src/bytewise.rs:1131-1137, serialised base,fail,opos_ch (1195-1199) */
typedef struct { uint32_t base, fail, opos_ch; } orc_state;
/* State<Empty>: 8 bytes, no fail (serializer.rs:133-146) */
typedef struct { uint32_t base, opos_ch; } orc_lstate;
/* Output<u32>: src/lib.rs:213-218, serialised value,length,parent (258-262) */
typedef struct { uint32_t value, length, parent; } orc_output;
/* Match<u32>: src/lib.rs:286-320; start = end - length */
typedef struct { uint64_t start, end; uint32_t value, _pad; } orc_match;

/* DoubleArrayAhoCorasick<u32>: src/bytewise.rs:54-68 */
typedef struct {
    orc_state *states;      size_t n_states;
    uint32_t *root_table;   size_t n_root;     /* 256 for Standard, 0 for leftmost */
    orc_lstate *lstates;    size_t n_lstates;
    uint32_t *fails;        size_t n_fails;
    orc_output *outputs;    size_t n_outputs;
    uint8_t match_kind;
    uint32_t num_states;
} orc_pma;

/* DoubleArrayAhoCorasickBuilder::build / build_with_values (bytewise/builder.rs:152-244).
 * patterns are passed as one blob + n+1 offsets; values == NULL => value = index. */
int orc_build(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n,
              uint8_t match_kind, uint32_t num_free_blocks, orc_pma **out);
void orc_free_pma(orc_pma *p);
void orc_free(void *p);

size_t orc_heap_bytes(const orc_pma *p);             /* bytewise.rs:764-770 */
/* bytewise.rs:801-820 / 868-964 */
int orc_serialize(const orc_pma *p, uint8_t **buf, size_t *len);
int orc_deserialize(const uint8_t *src, size_t len, orc_pma **out, size_t *consumed);

/* eager collectors over the literal iterator restatements (bytewise/iter.rs) */
int orc_find_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_find_overlapping_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_find_overlapping_no_suffix_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_leftmost_find_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
/* steppers driven exactly as tests/aho_corasick_crate_test.rs:422-445, 477-499 drive them */
int orc_find_stepper(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_find_overlapping_stepper(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);

/* Count + order-independent checksum of the find_overlapping_iter stream
 * (checksum = (S1 << 32) | S2, S1 = sum h, S2 = sum h * low32(end) mod 2^32, h = low32(mix64(value << 32 | length));
 * the HIP path implements the same definition).
 * `threads` > 1 splits the haystack into contiguous shards with an (Lmax-1)-byte halo. */
int orc_overlapping_count(const orc_pma *p, const uint8_t *hay, size_t len, int threads,
                          uint64_t *count, uint64_t *checksum);
/* checksum of an explicit match list (same definition) */
uint64_t orc_matches_checksum(const orc_match *m, size_t n);
uint32_t orc_max_pattern_len(const orc_pma *p);

/* ---------------------------------------------------------------- charwise engine (daac_oracle_charwise.c)
 * State: src/charwise.rs:1096-1101, serialised base,check,fail,output_pos (1162-1167); check = PARENT index */
typedef struct { uint32_t base, check, fail, output_pos; } orc_cstate;
/* CharwiseDoubleArrayAhoCorasick<u32>: src/charwise.rs:59-65 (mapper: src/charwise/mapper.rs:10-13) */
typedef struct {
    orc_cstate *states;   size_t n_states;
    uint32_t *table;      size_t n_table;   /* code point -> code, 0xffffffff = unmapped */
    uint32_t alphabet_size;
    orc_output *outputs;  size_t n_outputs;
    uint8_t match_kind;
    uint32_t num_states;
} orc_cpma;

/* patterns are UTF-8 (valid), one blob + n+1 offsets; values == NULL => value = index */
int orc_cbuild(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n,
               uint8_t match_kind, uint32_t num_free_blocks, orc_cpma **out);
void orc_cfree_pma(orc_cpma *p);
size_t orc_cheap_bytes(const orc_cpma *p);
int orc_cserialize(const orc_cpma *p, uint8_t **buf, size_t *len);
int orc_cdeserialize(const uint8_t *src, size_t len, orc_cpma **out, size_t *consumed);
/* haystacks are valid UTF-8; positions are BYTE offsets (charwise/iter.rs) */
int orc_cfind_iter(const orc_cpma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_cfind_overlapping_iter(const orc_cpma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_cfind_overlapping_no_suffix_iter(const orc_cpma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_cleftmost_find_iter(const orc_cpma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_cfind_stepper(const orc_cpma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);
int orc_cfind_overlapping_stepper(const orc_cpma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n);

#ifdef __cplusplus
}
#endif
#endif
