/*
 * oracle_internal.h — pieces shared by the bytewise and charwise oracle files (TEST INFRASTRUCTURE).
 * Restatements of reference src/build_helper.rs, the Match vector helper and the checksum.
 */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H

#include <stdlib.h>
#include <string.h>

#include "daac_oracle.h"

/* ================================================================ BuildHelper
 * build_helper.rs:16-227: circular doubly linked vacant list over the active blocks */
typedef struct { uint32_t next, prev; uint8_t used_base, used_index; } list_item;
typedef struct {
    list_item *items;
    uint32_t capacity; /* block_len * num_free_blocks */
    uint32_t block_len, num_free_blocks, num_blocks;
    int has_head;
    uint32_t head_idx;
} helper_t;

static inline uint32_t h_num_elements(const helper_t *h) { return h->num_blocks * h->block_len; }
static inline uint32_t h_active_block_start(const helper_t *h) { /* :61-63 */
    return h->num_blocks > h->num_free_blocks ? h->num_blocks - h->num_free_blocks : 0;
}
static inline list_item *h_at(const helper_t *h, uint32_t idx) { return &h->items[idx % h->capacity]; } /* :204-207 */

/* build_helper.rs:118-130 */
static inline void h_use_index(helper_t *h, uint32_t idx) {
    h_at(h, idx)->used_index = 1;
    uint32_t next = h_at(h, idx)->next;
    uint32_t prev = h_at(h, idx)->prev;
    h_at(h, prev)->next = next;
    h_at(h, next)->prev = prev;
    if (h->head_idx == idx) {
        if (next != idx) h->head_idx = next; else h->has_head = 0;
    }
}

/* build_helper.rs:177-179 */
static inline int h_dropped_block(const helper_t *h, uint32_t *blk) {
    if (h->capacity <= h_num_elements(h)) { *blk = h_active_block_start(h); return 1; }
    return 0;
}

/* build_helper.rs:133-173 */
static inline int h_push_block(helper_t *h) {
    if (h_num_elements(h) > 0xffffffffu - h->block_len) return ORC_ERR_AUTOMATON_SCALE;
    uint32_t closed;
    if (h_dropped_block(h, &closed)) {
        uint32_t end_idx = (closed + 1) * h->block_len;
        while (h->has_head) {
            if (end_idx <= h->head_idx) break;
            h_use_index(h, h->head_idx);
        }
    }
    uint32_t old_len = h_num_elements(h);
    uint32_t new_len = old_len + h->block_len;
    h->num_blocks += 1;
    for (uint32_t idx = old_len; idx < new_len; idx++) {
        list_item *it = h_at(h, idx);
        it->used_base = 0;
        it->used_index = 0;
        it->next = idx + 1;
        it->prev = idx - 1; /* wrapping_sub */
    }
    if (h->has_head) {
        uint32_t head = h->head_idx;
        uint32_t tail = h_at(h, head)->prev;
        h_at(h, old_len)->prev = tail;
        h_at(h, tail)->next = old_len;
        h_at(h, new_len - 1)->next = head;
        h_at(h, head)->prev = new_len - 1;
    } else {
        h_at(h, old_len)->prev = new_len - 1;
        h_at(h, new_len - 1)->next = old_len;
        h->has_head = 1;
        h->head_idx = old_len;
    }
    return ORC_OK;
}

/* build_helper.rs:76-80 */
static inline int h_unused_base_in_block(const helper_t *h, uint32_t block_idx, uint32_t *base) {
    uint32_t start = block_idx * h->block_len, end = start + h->block_len;
    for (uint32_t b = start; b < end; b++) {
        if (!h_at(h, b)->used_base) { *base = b; return 1; }
    }
    return 0;
}

/* ================================================================ match vector */
typedef struct { orc_match *m; size_t n, cap; } mvec;
static inline int mv_push(mvec *v, uint64_t length, uint64_t end, uint32_t value) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 64;
        orc_match *nm = (orc_match *)realloc(v->m, nc * sizeof(orc_match));
        if (!nm) return -1;
        v->m = nm; v->cap = nc;
    }
    v->m[v->n].start = end - length; /* lib.rs:301-303 */
    v->m[v->n].end = end;
    v->m[v->n].value = value;
    v->m[v->n]._pad = 0;
    v->n++;
    return 0;
}
static inline int mv_finish(mvec *v, orc_match **out, size_t *n) {
    if (!v->m) v->m = (orc_match *)malloc(sizeof(orc_match));
    *out = v->m; *n = v->n;
    return ORC_OK;
}

/* ================================================================ checksum pieces */
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
typedef struct { uint32_t s1, s2; } cksum;
static inline void ck_add(cksum *k, uint32_t value, uint64_t length, uint64_t end) {
    uint32_t h = (uint32_t)mix64(((uint64_t)value << 32) | (uint64_t)(uint32_t)length);
    k->s1 += h;
    k->s2 += h * (uint32_t)end;
}
static inline uint64_t ck_fin(cksum k) { return ((uint64_t)k.s1 << 32) | k.s2; }

#endif
