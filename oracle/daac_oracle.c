/*
 * daac_oracle.c — CPU ORACLE (test infrastructure; see daac_oracle.h).
 *
 * Plain-C restatement of daachorse 4.0.0's bytewise engine: sparse NFA construction,
 * double-array construction, (de)serialisation and every bytewise iterator/stepper.
 * Citations are file:line under /root/reference.  Nothing here is used by the product.
 */
#include "daac_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_internal.h"

#define ROOT_STATE_ID 0u   /* nfa_builder.rs:12 */
#define DEAD_STATE_ID 1u   /* nfa_builder.rs:14 */
#define ROOT_STATE_IDX 0u  /* bytewise.rs:25 */
#define DEAD_STATE_IDX 1u  /* bytewise.rs:27 */
#define BLOCK_LEN 256u     /* bytewise/builder.rs:15 */
#define U24_MAX 0x00ffffffu /* intpack.rs:15 */

/* ---------------------------------------------------------------- intpack.rs:31-54 */
static inline uint32_t opos_a(uint32_t x) { return x >> 8; }          /* output_pos (U24) */
static inline uint8_t opos_b(uint32_t x) { return (uint8_t)(x & 0xff); } /* check */
static inline uint32_t set_a(uint32_t x, uint32_t a) { return (a << 8) | (x & 0xff); }
static inline uint32_t set_b(uint32_t x, uint8_t b) { return ((x >> 8) << 8) | b; }

/* ================================================================ sparse NFA
 * nfa_builder.rs:33-58 (state), edge_map.rs:4-76 (label-sorted edge list) */
typedef struct {
    uint8_t *labels;
    uint32_t *children;
    uint32_t n_edges, cap_edges;
    uint32_t fail;
    uint32_t *out_vals, *out_lens; /* output: Vec<(V, u32)> */
    uint32_t n_out, cap_out;
    uint32_t output_pos; /* Option<NonZeroU32>, 0 = None */
} nfa_state;

typedef struct {
    nfa_state *states;
    size_t n_states, cap_states;
    orc_output *outputs;
    size_t n_outputs, cap_outputs;
    size_t len;
    uint8_t match_kind;
} nfa_t;

static int nfa_push_state(nfa_t *nfa) {
    if (nfa->n_states == nfa->cap_states) {
        size_t nc = nfa->cap_states ? nfa->cap_states * 2 : 1024;
        nfa_state *ns = (nfa_state *)realloc(nfa->states, nc * sizeof(nfa_state));
        if (!ns) return -1;
        nfa->states = ns;
        nfa->cap_states = nc;
    }
    memset(&nfa->states[nfa->n_states], 0, sizeof(nfa_state)); /* fail = ROOT, no output */
    nfa->n_states++;
    return 0;
}

/* edge_map.rs:46-55 */
static int edge_get(const nfa_state *s, uint8_t c, uint32_t *child) {
    uint32_t lo = 0, hi = s->n_edges;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (s->labels[mid] < c) lo = mid + 1; else hi = mid;
    }
    if (lo < s->n_edges && s->labels[lo] == c) { *child = s->children[lo]; return 1; }
    return 0;
}

/* edge_map.rs:15-44 (only called for absent keys) */
static int edge_insert(nfa_state *s, uint8_t c, uint32_t child) {
    if (s->n_edges == s->cap_edges) {
        uint32_t nc = s->cap_edges ? s->cap_edges * 2 : 2;
        uint8_t *nl = (uint8_t *)realloc(s->labels, nc);
        if (!nl) return -1;
        s->labels = nl;
        uint32_t *nch = (uint32_t *)realloc(s->children, nc * sizeof(uint32_t));
        if (!nch) return -1;
        s->children = nch;
        s->cap_edges = nc;
    }
    uint32_t pos = 0;
    while (pos < s->n_edges && s->labels[pos] < c) pos++;
    memmove(s->labels + pos + 1, s->labels + pos, s->n_edges - pos);
    memmove(s->children + pos + 1, s->children + pos, (s->n_edges - pos) * sizeof(uint32_t));
    s->labels[pos] = c;
    s->children[pos] = child;
    s->n_edges++;
    return 0;
}

/* nfa_builder.rs:78-113 */
static int nfa_add(nfa_t *nfa, const uint8_t *pat, size_t plen, uint32_t value) {
    if (plen > 0xffffffffull) return ORC_ERR_INVALID_ARGUMENT;
    uint32_t state_id = ROOT_STATE_ID;
    for (size_t i = 0; i < plen; i++) {
        uint8_t c = pat[i];
        if (nfa->match_kind == ORC_LEFTMOST_FIRST) {
            /* If state_id has an output, the descendants will never be searched. (87-92) */
            if (nfa->states[state_id].n_out != 0) return ORC_OK;
        }
        uint32_t next;
        if (edge_get(&nfa->states[state_id], c, &next)) {
            state_id = next;
        } else {
            if (nfa->n_states > 0xffffffffull) return ORC_ERR_AUTOMATON_SCALE;
            next = (uint32_t)nfa->n_states;
            if (nfa_push_state(nfa)) return ORC_ERR_AUTOMATON_SCALE;
            if (edge_insert(&nfa->states[state_id], c, next)) return ORC_ERR_AUTOMATON_SCALE;
            state_id = next;
        }
    }
    nfa_state *s = &nfa->states[state_id];
    if (s->n_out == s->cap_out) {
        uint32_t nc = s->cap_out ? s->cap_out * 2 : 1;
        s->out_vals = (uint32_t *)realloc(s->out_vals, nc * sizeof(uint32_t));
        s->out_lens = (uint32_t *)realloc(s->out_lens, nc * sizeof(uint32_t));
        if (!s->out_vals || !s->out_lens) return ORC_ERR_AUTOMATON_SCALE;
        s->cap_out = nc;
    }
    s->out_vals[s->n_out] = value;
    s->out_lens[s->n_out] = (uint32_t)plen;
    s->n_out++;
    nfa->len++;
    return ORC_OK;
}

/* nfa_builder.rs:115-144; returns the BFS queue (malloc'd), *qn = its length */
static uint32_t *nfa_build_fails(nfa_t *nfa, size_t *qn) {
    uint32_t *q = (uint32_t *)malloc((nfa->n_states + 1) * sizeof(uint32_t));
    size_t ql = 0;
    const nfa_state *root = &nfa->states[ROOT_STATE_ID];
    for (uint32_t i = 0; i < root->n_edges; i++) q[ql++] = root->children[i];
    size_t qi = 0;
    while (qi < ql) {
        uint32_t state_id = q[qi++];
        const nfa_state *s = &nfa->states[state_id];
        for (uint32_t e = 0; e < s->n_edges; e++) {
            uint8_t c = s->labels[e];
            uint32_t child_id = s->children[e];
            uint32_t fail_id = s->fail;
            uint32_t new_fail_id;
            for (;;) {
                uint32_t cf;
                if (edge_get(&nfa->states[fail_id], c, &cf)) { new_fail_id = cf; break; }
                uint32_t next_fail_id = nfa->states[fail_id].fail;
                if (fail_id == ROOT_STATE_ID && next_fail_id == ROOT_STATE_ID) { new_fail_id = ROOT_STATE_ID; break; }
                fail_id = next_fail_id;
            }
            nfa->states[child_id].fail = new_fail_id;
            q[ql++] = child_id;
        }
    }
    *qn = ql;
    return q;
}

/* nfa_builder.rs:146-201 */
static uint32_t *nfa_build_fails_leftmost(nfa_t *nfa, size_t *qn) {
    uint32_t *q = (uint32_t *)malloc((nfa->n_states + 1) * sizeof(uint32_t));
    size_t ql = 0;
    const nfa_state *root = &nfa->states[ROOT_STATE_ID];
    for (uint32_t i = 0; i < root->n_edges; i++) q[ql++] = root->children[i];
    if (root->n_out != 0) {
        for (uint32_t i = 0; i < root->n_edges; i++) nfa->states[root->children[i]].fail = DEAD_STATE_ID;
    }
    size_t qi = 0;
    while (qi < ql) {
        uint32_t state_id = q[qi++];
        nfa_state *s = &nfa->states[state_id];
        /* Sets the output state to the dead fail. (169-172) */
        if (s->n_out != 0) s->fail = DEAD_STATE_ID;
        for (uint32_t e = 0; e < s->n_edges; e++) {
            uint8_t c = s->labels[e];
            uint32_t child_id = s->children[e];
            uint32_t fail_id = s->fail;
            uint32_t new_fail_id;
            if (fail_id == DEAD_STATE_ID) {
                new_fail_id = DEAD_STATE_ID;
            } else {
                for (;;) {
                    uint32_t cf;
                    if (edge_get(&nfa->states[fail_id], c, &cf)) { new_fail_id = cf; break; }
                    uint32_t next_fail_id = nfa->states[fail_id].fail;
                    if (next_fail_id == DEAD_STATE_ID) { new_fail_id = DEAD_STATE_ID; break; }
                    if (fail_id == ROOT_STATE_ID && next_fail_id == ROOT_STATE_ID) { new_fail_id = ROOT_STATE_ID; break; }
                    fail_id = next_fail_id;
                }
            }
            nfa->states[child_id].fail = new_fail_id;
            q[ql++] = child_id;
        }
    }
    *qn = ql;
    return q;
}

static int nfa_push_output(nfa_t *nfa, uint32_t value, uint32_t length, uint32_t parent) {
    if (nfa->n_outputs == nfa->cap_outputs) {
        size_t nc = nfa->cap_outputs ? nfa->cap_outputs * 2 : 256;
        orc_output *no = (orc_output *)realloc(nfa->outputs, nc * sizeof(orc_output));
        if (!no) return -1;
        nfa->outputs = no;
        nfa->cap_outputs = nc;
    }
    nfa->outputs[nfa->n_outputs].value = value;
    nfa->outputs[nfa->n_outputs].length = length;
    nfa->outputs[nfa->n_outputs].parent = parent;
    nfa->n_outputs++;
    return 0;
}

/* nfa_builder.rs:203-222 */
static int nfa_build_outputs(nfa_t *nfa, const uint32_t *q, size_t qn) {
    {
        nfa_state *s = &nfa->states[ROOT_STATE_ID];
        uint32_t last_pos = 0;
        for (uint32_t k = s->n_out; k-- > 0;) {
            if (nfa_push_output(nfa, s->out_vals[k], s->out_lens[k], last_pos)) return -1;
            last_pos = (uint32_t)nfa->n_outputs;
        }
        s->output_pos = last_pos;
    }
    for (size_t i = 0; i < qn; i++) {
        nfa_state *s = &nfa->states[q[i]];
        uint32_t last_pos = nfa->states[s->fail].output_pos;
        for (uint32_t k = s->n_out; k-- > 0;) {
            if (nfa_push_output(nfa, s->out_vals[k], s->out_lens[k], last_pos)) return -1;
            last_pos = (uint32_t)nfa->n_outputs;
        }
        s->output_pos = last_pos;
    }
    return 0;
}

static void nfa_free(nfa_t *nfa) {
    for (size_t i = 0; i < nfa->n_states; i++) {
        free(nfa->states[i].labels);
        free(nfa->states[i].children);
        free(nfa->states[i].out_vals);
        free(nfa->states[i].out_lens);
    }
    free(nfa->states);
    free(nfa->outputs);
}

/* ================================================================ double array
 * bytewise/builder.rs:267-400 */
typedef struct { orc_state *states; size_t n, cap; } da_t;

static int da_resize(da_t *da, size_t n) {
    if (n > da->cap) {
        size_t nc = da->cap ? da->cap : 256;
        while (nc < n) nc *= 2;
        orc_state *ns = (orc_state *)realloc(da->states, nc * sizeof(orc_state));
        if (!ns) return -1;
        da->states = ns;
        da->cap = nc;
    }
    if (n > da->n) memset(da->states + da->n, 0, (n - da->n) * sizeof(orc_state)); /* State::default() */
    da->n = n;
    return 0;
}

/* builder.rs:391-400 */
static void remove_invalid_checks(da_t *da, uint32_t block_idx, const helper_t *h) {
    uint32_t unused_base;
    if (h_unused_base_in_block(h, block_idx, &unused_base)) {
        for (uint32_t c = 0; c <= 255; c++) {
            uint32_t idx = unused_base ^ c;
            if (idx == ROOT_STATE_IDX || idx == DEAD_STATE_IDX || !h_at(h, idx)->used_index)
                da->states[idx].opos_ch = set_b(da->states[idx].opos_ch, (uint8_t)c);
        }
    }
}

/* builder.rs:372-388 */
static int extend_array(da_t *da, helper_t *h) {
    if (da->n > (size_t)(0xffffffffu - BLOCK_LEN)) return ORC_ERR_AUTOMATON_SCALE;
    uint32_t closed;
    if (h_dropped_block(h, &closed)) remove_invalid_checks(da, closed, h);
    int rc = h_push_block(h);
    if (rc) return rc;
    if (da_resize(da, da->n + BLOCK_LEN)) return ORC_ERR_AUTOMATON_SCALE;
    return ORC_OK;
}

/* builder.rs:347-370 */
static uint32_t find_base(const da_t *da, const uint8_t *labels, uint32_t n_labels, const helper_t *h) {
    if (h->has_head) {
        uint32_t idx = h->head_idx;
        for (;;) {
            uint32_t base = idx ^ labels[0];
            int ok = !h_at(h, base)->used_base;
            for (uint32_t k = 0; ok && k < n_labels; k++)
                if (h_at(h, base ^ labels[k])->used_index) ok = 0;
            if (ok && base != 0) return base; /* NonZeroU32::new(base) */
            uint32_t next = h_at(h, idx)->next; /* VacantIter, build_helper.rs:219-226 */
            if (next == h->head_idx) break;
            idx = next;
        }
    }
    return (uint32_t)da->n;
}

static int build_double_array(da_t *da, const nfa_t *nfa, uint32_t num_free_blocks) {
    /* init_array, builder.rs:336-344 */
    helper_t h;
    memset(&h, 0, sizeof(h));
    uint64_t cap64 = (uint64_t)BLOCK_LEN * num_free_blocks; /* build_helper.rs:31-33 checked_mul */
    if (cap64 > 0xffffffffull) return ORC_ERR_AUTOMATON_SCALE;
    if (da_resize(da, BLOCK_LEN)) return ORC_ERR_AUTOMATON_SCALE;
    h.capacity = (uint32_t)cap64;
    h.block_len = BLOCK_LEN;
    h.num_free_blocks = num_free_blocks;
    h.items = (list_item *)calloc(h.capacity, sizeof(list_item));
    if (!h.items) return ORC_ERR_AUTOMATON_SCALE;
    int rc = h_push_block(&h);
    if (rc) { free(h.items); return rc; }
    h_use_index(&h, ROOT_STATE_IDX);
    h_use_index(&h, DEAD_STATE_IDX);

    uint32_t *state_id_map = (uint32_t *)malloc(nfa->n_states * sizeof(uint32_t));
    uint32_t *stack = (uint32_t *)malloc((nfa->n_states + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < nfa->n_states; i++) state_id_map[i] = DEAD_STATE_IDX;
    state_id_map[ROOT_STATE_ID] = ROOT_STATE_IDX;
    size_t sp = 0;
    stack[sp++] = ROOT_STATE_ID;

    /* Arranges base & check values (builder.rs:273-305) */
    while (sp > 0) {
        uint32_t state_id = stack[--sp];
        const nfa_state *s = &nfa->states[state_id];
        uint32_t state_idx = state_id_map[state_id];
        if (s->n_edges == 0) continue;
        uint32_t base = find_base(da, s->labels, s->n_edges, &h);
        if (base >= da->n) {
            rc = extend_array(da, &h);
            if (rc) goto done;
        }
        for (uint32_t e = 0; e < s->n_edges; e++) {
            uint8_t c = s->labels[e];
            uint32_t child_idx = base ^ c;
            h_use_index(&h, child_idx);
            da->states[child_idx].opos_ch = set_b(da->states[child_idx].opos_ch, c);
            state_id_map[s->children[e]] = child_idx;
            stack[sp++] = s->children[e];
        }
        da->states[state_idx].base = base;
        h_at(&h, base)->used_base = 1;
    }

    /* Sets fail & output_pos values (builder.rs:307-326) */
    for (size_t i = 0; i < nfa->n_states; i++) {
        if (i == DEAD_STATE_ID) continue;
        const nfa_state *s = &nfa->states[i];
        uint32_t idx = state_id_map[i];
        if (s->output_pos > U24_MAX) { rc = ORC_ERR_AUTOMATON_SCALE; goto done; }
        da->states[idx].opos_ch = set_a(da->states[idx].opos_ch, s->output_pos);
        if (s->fail == DEAD_STATE_ID) da->states[idx].fail = DEAD_STATE_IDX;
        else da->states[idx].fail = state_id_map[s->fail];
    }
    /* builder.rs:328-330 */
    for (uint32_t b = h_active_block_start(&h); b < h.num_blocks; b++) remove_invalid_checks(da, b, &h);
    rc = ORC_OK;
done:
    free(state_id_map);
    free(stack);
    free(h.items);
    return rc;
}

/* bytewise.rs:1040-1056 */
static uint32_t *build_root_table(const orc_state *states, size_t n) {
    uint32_t *t = (uint32_t *)calloc(256, sizeof(uint32_t));
    if (n > 0 && states[ROOT_STATE_IDX].base != 0) {
        uint32_t base = states[ROOT_STATE_IDX].base;
        for (uint32_t c = 0; c <= 255; c++) {
            uint32_t child_idx = base ^ c;
            if (child_idx < n && opos_b(states[child_idx].opos_ch) == (uint8_t)c) t[c] = child_idx;
        }
    }
    return t;
}

/* bytewise/builder.rs:152-244 */
int orc_build(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n,
              uint8_t match_kind, uint32_t num_free_blocks, orc_pma **out) {
    *out = NULL;
    if (num_free_blocks < 1 || match_kind > 2) return ORC_ERR_INVALID_ARGUMENT; /* builder.rs:113 (assert) */
    if (!values && n > 0xffffffffull) return ORC_ERR_INVALID_CONVERSION;       /* builder.rs:160-165 */
    nfa_t nfa;
    memset(&nfa, 0, sizeof(nfa));
    nfa.match_kind = match_kind;
    int rc = ORC_OK;
    uint32_t *q = NULL;
    da_t da = {0};
    if (nfa_push_state(&nfa) || nfa_push_state(&nfa)) { rc = ORC_ERR_AUTOMATON_SCALE; goto fail; } /* root, dead */
    /* build_sparse_nfa, builder.rs:246-265 */
    for (size_t i = 0; i < n; i++) {
        rc = nfa_add(&nfa, blob + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), values ? values[i] : (uint32_t)i);
        if (rc) goto fail;
    }
    if (nfa.len > U24_MAX) { rc = ORC_ERR_AUTOMATON_SCALE; goto fail; }
    size_t qn = 0;
    q = match_kind == ORC_STANDARD ? nfa_build_fails(&nfa, &qn) : nfa_build_fails_leftmost(&nfa, &qn);
    if (nfa_build_outputs(&nfa, q, qn)) { rc = ORC_ERR_AUTOMATON_SCALE; goto fail; }
    rc = build_double_array(&da, &nfa, num_free_blocks);
    if (rc) goto fail;

    orc_pma *p = (orc_pma *)calloc(1, sizeof(orc_pma));
    p->match_kind = match_kind;
    p->num_states = (uint32_t)(nfa.n_states - 1); /* -1 is for dead state */
    p->outputs = (orc_output *)malloc((nfa.n_outputs ? nfa.n_outputs : 1) * sizeof(orc_output));
    memcpy(p->outputs, nfa.outputs, nfa.n_outputs * sizeof(orc_output));
    p->n_outputs = nfa.n_outputs;
    if (match_kind != ORC_STANDARD) { /* builder.rs:220-231 */
        p->lstates = (orc_lstate *)malloc(da.n * sizeof(orc_lstate));
        p->fails = (uint32_t *)malloc(da.n * sizeof(uint32_t));
        for (size_t i = 0; i < da.n; i++) {
            p->lstates[i].base = da.states[i].base;
            p->lstates[i].opos_ch = da.states[i].opos_ch;
            p->fails[i] = da.states[i].fail;
        }
        p->n_lstates = p->n_fails = da.n;
        free(da.states);
    } else {
        p->states = (orc_state *)realloc(da.states, da.n * sizeof(orc_state)); /* shrink_to_fit */
        p->n_states = da.n;
        p->root_table = build_root_table(p->states, p->n_states);
        p->n_root = 256;
    }
    free(q);
    nfa_free(&nfa);
    *out = p;
    return ORC_OK;
fail:
    free(q);
    free(da.states);
    nfa_free(&nfa);
    return rc;
}

void orc_free_pma(orc_pma *p) {
    if (!p) return;
    free(p->states); free(p->root_table); free(p->lstates); free(p->fails); free(p->outputs);
    free(p);
}
void orc_free(void *p) { free(p); }

/* bytewise.rs:764-770 */
size_t orc_heap_bytes(const orc_pma *p) {
    return p->n_states * 12 + p->n_root * 4 + p->n_lstates * 8 + p->n_fails * 4 + p->n_outputs * 12;
}

uint32_t orc_max_pattern_len(const orc_pma *p) {
    uint32_t m = 0;
    for (size_t i = 0; i < p->n_outputs; i++) if (p->outputs[i].length > m) m = p->outputs[i].length;
    return m;
}

/* ================================================================ serialisation
 * bytewise.rs:801-820; serializer.rs:38-131 (LE u32, Vec = u32 len + items) */
static void put_u32(uint8_t **w, uint32_t x) {
    (*w)[0] = (uint8_t)x; (*w)[1] = (uint8_t)(x >> 8); (*w)[2] = (uint8_t)(x >> 16); (*w)[3] = (uint8_t)(x >> 24);
    *w += 4;
}
int orc_serialize(const orc_pma *p, uint8_t **buf, size_t *len) {
    size_t total = 4 + p->n_states * 12 + 4 + p->n_lstates * 8 + 4 + p->n_fails * 4 + 4 + p->n_outputs * 12 + 1 + 4;
    uint8_t *b = (uint8_t *)malloc(total), *w = b;
    if (!b) return ORC_ERR_AUTOMATON_SCALE;
    put_u32(&w, (uint32_t)p->n_states);
    for (size_t i = 0; i < p->n_states; i++) { put_u32(&w, p->states[i].base); put_u32(&w, p->states[i].fail); put_u32(&w, p->states[i].opos_ch); }
    put_u32(&w, (uint32_t)p->n_lstates);
    for (size_t i = 0; i < p->n_lstates; i++) { put_u32(&w, p->lstates[i].base); put_u32(&w, p->lstates[i].opos_ch); }
    put_u32(&w, (uint32_t)p->n_fails);
    for (size_t i = 0; i < p->n_fails; i++) put_u32(&w, p->fails[i]);
    put_u32(&w, (uint32_t)p->n_outputs);
    for (size_t i = 0; i < p->n_outputs; i++) { put_u32(&w, p->outputs[i].value); put_u32(&w, p->outputs[i].length); put_u32(&w, p->outputs[i].parent); }
    *w++ = p->match_kind;
    put_u32(&w, p->num_states);
    *buf = b;
    *len = total;
    return ORC_OK;
}

typedef struct { const uint8_t *p; size_t left; } rd_t;
static int get_u32(rd_t *r, uint32_t *x) { /* serializer.rs:46-53 */
    if (r->left < 4) return -1;
    *x = (uint32_t)r->p[0] | ((uint32_t)r->p[1] << 8) | ((uint32_t)r->p[2] << 16) | ((uint32_t)r->p[3] << 24);
    r->p += 4; r->left -= 4;
    return 0;
}
/* serializer.rs:110-125: Vec header + allocation guard using the IN-MEMORY element size */
static int get_vec_len(rd_t *r, size_t mem_size, uint32_t *n) {
    if (get_u32(r, n)) return -1;
    if ((uint64_t)*n * mem_size > r->left) return -1;
    return 0;
}

/* bytewise.rs:868-964 */
int orc_deserialize(const uint8_t *src, size_t len, orc_pma **out, size_t *consumed) {
    *out = NULL;
    rd_t r = {src, len};
    orc_pma *p = (orc_pma *)calloc(1, sizeof(orc_pma));
    uint32_t n, x;
    if (get_vec_len(&r, 12, &n)) goto bad;
    p->states = (orc_state *)malloc((n ? n : 1) * sizeof(orc_state)); p->n_states = n;
    for (uint32_t i = 0; i < n; i++) if (get_u32(&r, &p->states[i].base) || get_u32(&r, &p->states[i].fail) || get_u32(&r, &p->states[i].opos_ch)) goto bad;
    if (get_vec_len(&r, 8, &n)) goto bad;
    p->lstates = (orc_lstate *)malloc((n ? n : 1) * sizeof(orc_lstate)); p->n_lstates = n;
    for (uint32_t i = 0; i < n; i++) if (get_u32(&r, &p->lstates[i].base) || get_u32(&r, &p->lstates[i].opos_ch)) goto bad;
    if (get_vec_len(&r, 4, &n)) goto bad;
    p->fails = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)); p->n_fails = n;
    for (uint32_t i = 0; i < n; i++) if (get_u32(&r, &p->fails[i])) goto bad;
    if (get_vec_len(&r, 12, &n)) goto bad;
    p->outputs = (orc_output *)malloc((n ? n : 1) * sizeof(orc_output)); p->n_outputs = n;
    for (uint32_t i = 0; i < n; i++) if (get_u32(&r, &p->outputs[i].value) || get_u32(&r, &p->outputs[i].length) || get_u32(&r, &p->outputs[i].parent)) goto bad;
    if (r.left < 1) goto bad;
    x = *r.p++; r.left--;
    p->match_kind = (x == 1) ? ORC_LEFTMOST_LONGEST : (x == 2) ? ORC_LEFTMOST_FIRST : ORC_STANDARD; /* lib.rs:366-374 */
    if (get_u32(&r, &p->num_states)) goto bad;
    if (p->match_kind == ORC_STANDARD) { p->root_table = build_root_table(p->states, p->n_states); p->n_root = 256; }

    if (p->match_kind != ORC_STANDARD) { /* bytewise.rs:894-925 */
        if (p->n_states != 0) goto bad;
        if (p->n_lstates == 0) goto bad;
        if (p->n_lstates % 256 != 0) goto bad;
        if (p->n_fails != p->n_lstates) goto bad;
        for (size_t i = 0; i < p->n_lstates; i++) {
            if (p->lstates[i].base != 0 && p->lstates[i].base >= p->n_lstates) goto bad;
            uint32_t op = opos_a(p->lstates[i].opos_ch);
            if (op != 0 && (size_t)(op - 1) >= p->n_outputs) goto bad;
        }
        for (size_t i = 0; i < p->n_fails; i++) if (p->fails[i] >= p->n_lstates) goto bad;
    } else { /* bytewise.rs:926-954 */
        if (p->n_lstates != 0 || p->n_fails != 0) goto bad;
        if (p->n_states == 0) goto bad;
        if (p->n_states % 256 != 0) goto bad;
        for (size_t i = 0; i < p->n_states; i++) {
            if (p->states[i].base != 0 && p->states[i].base >= p->n_states) goto bad;
            if (p->states[i].fail >= p->n_states) goto bad;
            uint32_t op = opos_a(p->states[i].opos_ch);
            if (op != 0 && (size_t)(op - 1) >= p->n_outputs) goto bad;
        }
    }
    for (size_t i = 0; i < p->n_outputs; i++) { /* bytewise.rs:955-962 */
        uint32_t par = p->outputs[i].parent;
        if (par != 0 && (size_t)(par - 1) >= i) goto bad;
    }
    *out = p;
    if (consumed) *consumed = len - r.left;
    return ORC_OK;
bad:
    orc_free_pma(p);
    return ORC_ERR_INVALID_AUTOMATON;
}

/* ================================================================ transitions */
/* bytewise.rs:1063-1088 */
static inline uint32_t next_state_id(const orc_pma *p, uint32_t state_id, uint8_t c) {
    for (;;) {
        if (state_id == ROOT_STATE_IDX) return p->root_table[c];
        const orc_state *s = &p->states[state_id];
        if (s->base != 0) {
            uint32_t child_idx = s->base ^ c;
            if (opos_b(p->states[child_idx].opos_ch) == c) return child_idx;
        }
        state_id = s->fail;
    }
}

/* bytewise.rs:1094-1128 */
static inline uint32_t next_state_id_leftmost(const orc_pma *p, uint32_t state_id, uint8_t c) {
    for (;;) {
        const orc_lstate *s = &p->lstates[state_id];
        if (s->base != 0) {
            uint32_t child_idx = s->base ^ c;
            if (opos_b(p->lstates[child_idx].opos_ch) == c) return child_idx;
        }
        if (state_id == ROOT_STATE_IDX) return ROOT_STATE_IDX;
        uint32_t fail_id = p->fails[state_id];
        if (fail_id == DEAD_STATE_IDX) return ROOT_STATE_IDX;
        state_id = fail_id;
    }
}

/* ================================================================ iterators
 * Each `*_next` returns 1 and fills (length,end,value) for Some(Match), 0 for None. */

/* FindIterator, bytewise/iter.rs:44-114 */
typedef struct { const orc_pma *p; const uint8_t *h; size_t len, hpos; int first_call; } find_it;
static int find_next(find_it *it, uint64_t *length, uint64_t *end, uint32_t *value) {
    const orc_pma *p = it->p;
    uint32_t op = opos_a(p->states[ROOT_STATE_IDX].opos_ch);
    if (op != 0) { /* iter.rs:60-85 */
        *value = p->outputs[op - 1].value;
        *length = 0;
        if (it->first_call) { it->first_call = 0; *end = 0; return 1; }
        if (it->hpos < it->len) { *end = it->hpos + 1; it->hpos++; return 1; }
        return 0;
    }
    uint32_t state_id = ROOT_STATE_IDX;
    while (it->hpos < it->len) { /* iter.rs:88-111 */
        size_t pos = it->hpos;
        uint8_t c = it->h[it->hpos++];
        state_id = next_state_id(p, state_id, c);
        uint32_t o = opos_a(p->states[state_id].opos_ch);
        if (o != 0) {
            const orc_output *out = &p->outputs[o - 1];
            *length = out->length; *end = pos + 1; *value = out->value;
            return 1;
        }
    }
    return 0;
}

int orc_find_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND; /* bytewise.rs:194-197 */
    find_it it = {p, hay, len, 0, 1};
    mvec v = {0};
    uint64_t l, e; uint32_t val;
    while (find_next(&it, &l, &e, &val)) if (mv_push(&v, l, e, val)) return ORC_ERR_AUTOMATON_SCALE;
    return mv_finish(&v, out, n);
}

/* FindOverlappingIterator, bytewise/iter.rs:117-177 */
typedef struct { const orc_pma *p; const uint8_t *h; size_t len, hpos; uint32_t state_id; size_t pos; uint32_t output_pos; } ovl_it;
static int ovl_next(ovl_it *it, uint64_t *length, uint64_t *end, uint32_t *value) {
    const orc_pma *p = it->p;
    if (it->output_pos != 0) { /* iter.rs:134-148 */
        const orc_output *out = &p->outputs[it->output_pos - 1];
        it->output_pos = out->parent;
        *length = out->length; *end = it->pos; *value = out->value;
        return 1;
    }
    while (it->hpos < it->len) { /* iter.rs:149-174 */
        size_t pos = it->hpos;
        uint8_t c = it->h[it->hpos++];
        it->state_id = next_state_id(p, it->state_id, c);
        uint32_t o = opos_a(p->states[it->state_id].opos_ch);
        if (o != 0) {
            it->pos = pos + 1;
            const orc_output *out = &p->outputs[o - 1];
            it->output_pos = out->parent;
            *length = out->length; *end = it->pos; *value = out->value;
            return 1;
        }
    }
    return 0;
}

int orc_find_overlapping_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND; /* bytewise.rs:299-302 */
    ovl_it it = {p, hay, len, 0, ROOT_STATE_IDX, 0, opos_a(p->states[ROOT_STATE_IDX].opos_ch)};
    mvec v = {0};
    uint64_t l, e; uint32_t val;
    while (ovl_next(&it, &l, &e, &val)) if (mv_push(&v, l, e, val)) return ORC_ERR_AUTOMATON_SCALE;
    return mv_finish(&v, out, n);
}

/* FindOverlappingNoSuffixIterator, bytewise/iter.rs:180-244 */
int orc_find_overlapping_no_suffix_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t state_id = ROOT_STATE_IDX;
    uint32_t op = opos_a(p->states[ROOT_STATE_IDX].opos_ch);
    if (op != 0) if (mv_push(&v, 0, 0, p->outputs[op - 1].value)) return ORC_ERR_AUTOMATON_SCALE; /* first_call */
    for (size_t pos = 0; pos < len; pos++) {
        state_id = next_state_id(p, state_id, hay[pos]);
        uint32_t o = opos_a(p->states[state_id].opos_ch);
        if (o != 0) {
            const orc_output *x = &p->outputs[o - 1];
            if (mv_push(&v, x->length, pos + 1, x->value)) return ORC_ERR_AUTOMATON_SCALE;
        }
    }
    return mv_finish(&v, out, n);
}

/* LeftmostFindIterator, bytewise/iter.rs:247-341 */
typedef struct { const orc_pma *p; const uint8_t *h; size_t len; size_t pos; uint32_t init_output_pos; int skip_empty; } lm_it;
static int lm_next(lm_it *it, uint64_t *length, uint64_t *end, uint32_t *value) {
    const orc_pma *p = it->p;
    uint32_t state_id = ROOT_STATE_IDX;
    uint32_t last_output_pos = it->init_output_pos;
    for (;;) { /* 'a: loop */
        int cont = 0;
        for (size_t pos = it->pos; pos < it->len; pos++) { /* .enumerate().skip(self.pos), evaluated at loop entry */
            uint8_t c = it->h[pos];
            state_id = next_state_id_leftmost(p, state_id, c);
            if (state_id == ROOT_STATE_IDX) {
                if (last_output_pos != 0) {
                    uint32_t output_pos = last_output_pos;
                    size_t e = it->pos;
                    if (last_output_pos == it->init_output_pos) {
                        it->pos += 1;
                        if (it->skip_empty) { it->skip_empty = 0; cont = 1; break; } /* continue 'a */
                    } else {
                        it->skip_empty = 1;
                    }
                    const orc_output *out = &p->outputs[output_pos - 1];
                    *length = out->length; *end = e; *value = out->value;
                    return 1;
                }
            } else {
                uint32_t o = opos_a(p->lstates[state_id].opos_ch);
                if (o != 0) { last_output_pos = o; it->pos = pos + 1; }
            }
        }
        if (!cont) break;
    }
    if (it->pos == it->len) it->init_output_pos = 0; /* iter.rs:320-322 */
    if (last_output_pos != 0) {
        const orc_output *out = &p->outputs[last_output_pos - 1];
        *length = out->length; *end = it->pos; *value = out->value;
        return 1;
    }
    it->pos = it->len;
    return 0;
}

int orc_leftmost_find_iter(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind == ORC_STANDARD) return ORC_ERR_MATCH_KIND; /* bytewise.rs:551-554 */
    lm_it it = {p, hay, len, 0, opos_a(p->lstates[ROOT_STATE_IDX].opos_ch), 0};
    mvec v = {0};
    uint64_t l, e; uint32_t val;
    while (lm_next(&it, &l, &e, &val)) {
        if (mv_push(&v, l, e, val)) return ORC_ERR_AUTOMATON_SCALE;
        if (v.n > 2 * len + 4) { free(v.m); return ORC_ERR_DIVERGED; } /* SURVEY §8a note D */
    }
    return mv_finish(&v, out, n);
}

/* FindStepper, bytewise/iter.rs:344-401, driven as tests/aho_corasick_crate_test.rs:422-445 */
int orc_find_stepper(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t state_id = ROOT_STATE_IDX;
    size_t pos = 0;
    uint32_t output_pos = opos_a(p->states[ROOT_STATE_IDX].opos_ch);
    uint32_t root_op = output_pos;
    if (output_pos != 0) { /* matches() before any consume */
        const orc_output *x = &p->outputs[output_pos - 1];
        if (mv_push(&v, x->length, pos, x->value)) return ORC_ERR_AUTOMATON_SCALE;
    }
    for (size_t i = 0; i < len; i++) {
        /* consume(), iter.rs:357-381 */
        pos += 1;
        if (root_op == 0) {
            state_id = next_state_id(p, state_id, hay[i]);
            output_pos = opos_a(p->states[state_id].opos_ch);
            if (output_pos != 0) state_id = ROOT_STATE_IDX;
        }
        /* matches(), iter.rs:386-400 */
        if (output_pos != 0) {
            const orc_output *x = &p->outputs[output_pos - 1];
            if (mv_push(&v, x->length, pos, x->value)) return ORC_ERR_AUTOMATON_SCALE;
        }
    }
    return mv_finish(&v, out, n);
}

/* FindOverlappingStepper(+Iterator), bytewise/iter.rs:404-475, driven as tests/...:477-499 */
int orc_find_overlapping_stepper(const orc_pma *p, const uint8_t *hay, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t state_id = ROOT_STATE_IDX;
    size_t pos = 0;
    for (size_t i = 0;; i++) {
        uint32_t op = opos_a(p->states[state_id].opos_ch); /* matches() */
        while (op != 0) {
            const orc_output *x = &p->outputs[op - 1];
            op = x->parent;
            if (mv_push(&v, x->length, pos, x->value)) return ORC_ERR_AUTOMATON_SCALE;
        }
        if (i == len) break;
        state_id = next_state_id(p, state_id, hay[i]); /* consume() */
        pos += 1;
    }
    return mv_finish(&v, out, n);
}

/* ================================================================ count + checksum
 * Over all matches, with h = low32(mix64(value << 32 | length)) (mix64 = SplitMix64 finaliser):
 *     S1 = sum h            (mod 2^32)
 *     S2 = sum h * low32(end)   (mod 2^32)
 *     checksum = (S1 << 32) | S2
 * Order independent, and linear in per-output-list constants, so a scanner can fold a whole
 * output list with one multiply-add (the HIP path does).  This file computes it the slow,
 * obvious way: match by match. */
uint64_t orc_matches_checksum(const orc_match *m, size_t n) {
    cksum k = {0, 0};
    for (size_t i = 0; i < n; i++) ck_add(&k, m[i].value, m[i].end - m[i].start, m[i].end);
    return ck_fin(k);
}

/* Scans hay[from..to) with the literal iterator, counting only matches with end > lo.
 * `from` is at most Lmax-1 bytes before `lo` (SURVEY §8a note A). */
static void count_range(const orc_pma *p, const uint8_t *hay, size_t from, size_t to, size_t lo,
                        int emit_root, uint64_t *count, cksum *sum) {
    ovl_it it = {p, hay + from, to - from, 0, ROOT_STATE_IDX, 0,
                 emit_root ? opos_a(p->states[ROOT_STATE_IDX].opos_ch) : 0};
    uint64_t l, e, c = 0; uint32_t val;
    cksum k = {0, 0};
    while (ovl_next(&it, &l, &e, &val)) {
        uint64_t ge = e + from;
        if (ge > lo || (emit_root && ge == 0)) { c++; ck_add(&k, val, l, ge); }
    }
    *count = c; *sum = k;
}

typedef struct { const orc_pma *p; const uint8_t *hay; size_t from, to, lo; int emit_root; uint64_t count; cksum sum; } cr_job;
static void *cr_thread(void *a) {
    cr_job *j = (cr_job *)a;
    count_range(j->p, j->hay, j->from, j->to, j->lo, j->emit_root, &j->count, &j->sum);
    return NULL;
}

int orc_overlapping_count(const orc_pma *p, const uint8_t *hay, size_t len, int threads,
                          uint64_t *count, uint64_t *checksum) {
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    if (threads < 1) threads = 1;
    if (threads == 1 || len < (size_t)threads * 4096) {
        cksum k;
        count_range(p, hay, 0, len, 0, 1, count, &k);
        *checksum = ck_fin(k);
        return ORC_OK;
    }
    uint32_t lmax = orc_max_pattern_len(p);
    size_t halo = lmax > 0 ? lmax - 1 : 0;
    cr_job *jobs = (cr_job *)calloc((size_t)threads, sizeof(cr_job));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        size_t lo = len / (size_t)threads * (size_t)t;
        size_t hi = (t == threads - 1) ? len : len / (size_t)threads * (size_t)(t + 1);
        jobs[t].p = p; jobs[t].hay = hay;
        jobs[t].from = (t == 0) ? 0 : (lo > halo ? lo - halo : 0);
        jobs[t].to = hi; jobs[t].lo = lo; jobs[t].emit_root = (t == 0);
        pthread_create(&th[t], NULL, cr_thread, &jobs[t]);
    }
    uint64_t c = 0;
    cksum k = {0, 0};
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); c += jobs[t].count; k.s1 += jobs[t].sum.s1; k.s2 += jobs[t].sum.s2; }
    free(jobs); free(th);
    *count = c; *checksum = ck_fin(k);
    return ORC_OK;
}
