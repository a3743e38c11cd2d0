/*
 * daac_oracle_charwise.c — CPU ORACLE for the charwise engine (TEST INFRASTRUCTURE; see daac_oracle.h).
 *
 * Plain-C restatement of daachorse 4.0.0's CharwiseDoubleArrayAhoCorasick<u32>: code mapper
 * (src/charwise/mapper.rs), construction (src/charwise/builder.rs over src/nfa_builder.rs with
 * `char` labels), (de)serialisation and transitions (src/charwise.rs), the UTF-8 decoder and every
 * iterator/stepper (src/charwise/iter.rs).  Citations are file:line under /root/reference.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle_internal.h"

#define ROOT_IDX 0u
#define DEAD_IDX 1u
#define INVALID_CODE 0xffffffffu /* mapper.rs:7 */

/* ================================================================ sparse NFA over chars */
typedef struct {
    uint32_t *labels;   /* code points, ascending (EdgeMap<char>, edge_map.rs) */
    uint32_t *children;
    uint32_t n_edges, cap_edges;
    uint32_t fail;
    uint32_t *out_vals, *out_lens;
    uint32_t n_out, cap_out;
    uint32_t output_pos;
} cnfa_state;

typedef struct {
    cnfa_state *states;
    size_t n_states, cap_states;
    orc_output *outputs;
    size_t n_outputs, cap_outputs;
    size_t len;
    uint8_t match_kind;
} cnfa_t;

static int cnfa_push_state(cnfa_t *nfa) {
    if (nfa->n_states == nfa->cap_states) {
        size_t nc = nfa->cap_states ? nfa->cap_states * 2 : 1024;
        cnfa_state *ns = (cnfa_state *)realloc(nfa->states, nc * sizeof(cnfa_state));
        if (!ns) return -1;
        nfa->states = ns;
        nfa->cap_states = nc;
    }
    memset(&nfa->states[nfa->n_states], 0, sizeof(cnfa_state));
    nfa->n_states++;
    return 0;
}

static int cedge_get(const cnfa_state *s, uint32_t c, uint32_t *child) {
    uint32_t lo = 0, hi = s->n_edges;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (s->labels[mid] < c) lo = mid + 1; else hi = mid;
    }
    if (lo < s->n_edges && s->labels[lo] == c) { *child = s->children[lo]; return 1; }
    return 0;
}

static int cedge_insert(cnfa_state *s, uint32_t c, uint32_t child) {
    if (s->n_edges == s->cap_edges) {
        uint32_t nc = s->cap_edges ? s->cap_edges * 2 : 2;
        s->labels = (uint32_t *)realloc(s->labels, nc * sizeof(uint32_t));
        s->children = (uint32_t *)realloc(s->children, nc * sizeof(uint32_t));
        if (!s->labels || !s->children) return -1;
        s->cap_edges = nc;
    }
    uint32_t pos = 0;
    while (pos < s->n_edges && s->labels[pos] < c) pos++;
    memmove(s->labels + pos + 1, s->labels + pos, (s->n_edges - pos) * sizeof(uint32_t));
    memmove(s->children + pos + 1, s->children + pos, (s->n_edges - pos) * sizeof(uint32_t));
    s->labels[pos] = c;
    s->children[pos] = child;
    s->n_edges++;
    return 0;
}

static uint32_t utf8_len(uint32_t c) { return c < 0x80 ? 1 : c < 0x800 ? 2 : c < 0x10000 ? 3 : 4; }

/* nfa_builder.rs:78-113 with L = char: pattern_len is the BYTE length (num_bytes = len_utf8, :26-30) */
static int cnfa_add(cnfa_t *nfa, const uint32_t *chars, size_t n, uint32_t value) {
    uint64_t blen = 0;
    for (size_t i = 0; i < n; i++) blen += utf8_len(chars[i]);
    if (blen > 0xffffffffull) return ORC_ERR_INVALID_ARGUMENT;
    uint32_t state_id = 0;
    for (size_t i = 0; i < n; i++) {
        if (nfa->match_kind == ORC_LEFTMOST_FIRST && nfa->states[state_id].n_out != 0) return ORC_OK;
        uint32_t next;
        if (cedge_get(&nfa->states[state_id], chars[i], &next)) {
            state_id = next;
        } else {
            next = (uint32_t)nfa->n_states;
            if (cnfa_push_state(nfa)) return ORC_ERR_AUTOMATON_SCALE;
            if (cedge_insert(&nfa->states[state_id], chars[i], next)) return ORC_ERR_AUTOMATON_SCALE;
            state_id = next;
        }
    }
    cnfa_state *s = &nfa->states[state_id];
    if (s->n_out == s->cap_out) {
        uint32_t nc = s->cap_out ? s->cap_out * 2 : 1;
        s->out_vals = (uint32_t *)realloc(s->out_vals, nc * sizeof(uint32_t));
        s->out_lens = (uint32_t *)realloc(s->out_lens, nc * sizeof(uint32_t));
        if (!s->out_vals || !s->out_lens) return ORC_ERR_AUTOMATON_SCALE;
        s->cap_out = nc;
    }
    s->out_vals[s->n_out] = value;
    s->out_lens[s->n_out] = (uint32_t)blen;
    s->n_out++;
    nfa->len++;
    return ORC_OK;
}

/* nfa_builder.rs:115-144 / 146-201 (same algorithm as the bytewise file, char labels) */
static uint32_t *cnfa_build_fails(cnfa_t *nfa, size_t *qn, int leftmost) {
    uint32_t *q = (uint32_t *)malloc((nfa->n_states + 1) * sizeof(uint32_t));
    size_t ql = 0;
    const cnfa_state *root = &nfa->states[0];
    for (uint32_t i = 0; i < root->n_edges; i++) q[ql++] = root->children[i];
    if (leftmost && root->n_out != 0)
        for (uint32_t i = 0; i < root->n_edges; i++) nfa->states[root->children[i]].fail = 1;
    size_t qi = 0;
    while (qi < ql) {
        cnfa_state *s = &nfa->states[q[qi++]];
        if (leftmost && s->n_out != 0) s->fail = 1;
        for (uint32_t e = 0; e < s->n_edges; e++) {
            uint32_t c = s->labels[e], child_id = s->children[e], fail_id = s->fail, nf;
            if (leftmost && fail_id == 1) {
                nf = 1;
            } else {
                for (;;) {
                    uint32_t cf;
                    if (cedge_get(&nfa->states[fail_id], c, &cf)) { nf = cf; break; }
                    uint32_t next = nfa->states[fail_id].fail;
                    if (leftmost && next == 1) { nf = 1; break; }
                    if (fail_id == 0 && next == 0) { nf = 0; break; }
                    fail_id = next;
                }
            }
            nfa->states[child_id].fail = nf;
            q[ql++] = child_id;
        }
    }
    *qn = ql;
    return q;
}

static int cnfa_push_output(cnfa_t *nfa, uint32_t value, uint32_t length, uint32_t parent) {
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
static int cnfa_build_outputs(cnfa_t *nfa, const uint32_t *q, size_t qn) {
    for (size_t k = 0; k <= qn; k++) {
        cnfa_state *s = &nfa->states[k == 0 ? 0 : q[k - 1]];
        uint32_t last_pos = k == 0 ? 0 : nfa->states[s->fail].output_pos;
        for (uint32_t i = s->n_out; i-- > 0;) {
            if (cnfa_push_output(nfa, s->out_vals[i], s->out_lens[i], last_pos)) return -1;
            last_pos = (uint32_t)nfa->n_outputs;
        }
        s->output_pos = last_pos;
    }
    return 0;
}

static void cnfa_free(cnfa_t *nfa) {
    for (size_t i = 0; i < nfa->n_states; i++) {
        free(nfa->states[i].labels); free(nfa->states[i].children);
        free(nfa->states[i].out_vals); free(nfa->states[i].out_lens);
    }
    free(nfa->states);
    free(nfa->outputs);
}

/* ================================================================ UTF-8 decoding
 * charwise/iter.rs:64-98 (unchecked decode; the haystack is valid UTF-8) */
static inline uint32_t decode_at(const uint8_t *h, size_t i, size_t *end) {
    uint32_t first = h[i];
    if (first < 0x80) { *end = i + 1; return first; }
    uint32_t c = h[i + 1] & 0x3f;
    if (first < 0xe0) { *end = i + 2; return ((first & 0x1f) << 6) | c; }
    c = (c << 6) | (h[i + 2] & 0x3f);
    if (first < 0xf0) { *end = i + 3; return ((first & 0x0f) << 12) | c; }
    c = (c << 6) | (h[i + 3] & 0x3f);
    *end = i + 4;
    return ((first & 0x07) << 18) | c;
}

static int decode_all(const uint8_t *p, size_t len, uint32_t **out, size_t *n) {
    uint32_t *v = (uint32_t *)malloc((len ? len : 1) * sizeof(uint32_t));
    size_t k = 0, i = 0;
    while (i < len) { size_t e; v[k++] = decode_at(p, i, &e); i = e; }
    *out = v; *n = k;
    return 0;
}

/* ================================================================ mapper + double array
 * mapper.rs:16-34 */
typedef struct { uint32_t c, f; } cf_t;
static int cf_cmp(const void *a, const void *b) {
    const cf_t *x = (const cf_t *)a, *y = (const cf_t *)b;
    if (x->f != y->f) return x->f > y->f ? -1 : 1; /* frequency descending */
    return x->c < y->c ? -1 : x->c > y->c;       /* then code point ascending */
}

static inline uint32_t mapper_get(const orc_cpma *p, uint32_t c) { /* mapper.rs:36-42 */
    if (c >= p->n_table) return INVALID_CODE;
    return p->table[c];
}

/* charwise/builder.rs:241-359 */
static int cbuild_double_array(orc_cpma *p, const cnfa_t *nfa, uint32_t num_free_blocks) {
    uint32_t block_len = 2; /* alphabet_size.next_power_of_two().max(2), :308 */
    while (block_len < p->alphabet_size) block_len <<= 1;
    uint64_t cap64 = (uint64_t)block_len * num_free_blocks;
    if (cap64 > 0xffffffffull) return ORC_ERR_AUTOMATON_SCALE;
    helper_t h;
    memset(&h, 0, sizeof(h));
    h.capacity = (uint32_t)cap64;
    h.block_len = block_len;
    h.num_free_blocks = num_free_blocks;
    h.items = (list_item *)calloc(h.capacity, sizeof(list_item));
    if (!h.items) return ORC_ERR_AUTOMATON_SCALE;
    size_t n = block_len, cap = block_len;
    orc_cstate *st = (orc_cstate *)malloc(cap * sizeof(orc_cstate));
    const orc_cstate dflt = {0, DEAD_IDX, DEAD_IDX, 0}; /* State::default(), charwise.rs:1103-1112 */
    for (size_t i = 0; i < n; i++) st[i] = dflt;
    int rc = h_push_block(&h);
    if (rc) { free(h.items); free(st); return rc; }
    h_use_index(&h, ROOT_IDX);
    h_use_index(&h, DEAD_IDX);

    uint32_t *state_id_map = (uint32_t *)malloc(nfa->n_states * sizeof(uint32_t));
    uint32_t *stack = (uint32_t *)malloc((nfa->n_states + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < nfa->n_states; i++) state_id_map[i] = DEAD_IDX;
    state_id_map[0] = ROOT_IDX;
    size_t sp = 0;
    stack[sp++] = 0;
    cf_t *mapped = NULL;
    size_t mapped_cap = 0;
    while (sp > 0) {
        uint32_t state_id = stack[--sp];
        const cnfa_state *s = &nfa->states[state_id];
        uint32_t state_idx = state_id_map[state_id];
        if (s->n_edges == 0) continue;
        if (s->n_edges > mapped_cap) { mapped_cap = s->n_edges * 2; mapped = (cf_t *)realloc(mapped, mapped_cap * sizeof(cf_t)); }
        for (uint32_t e = 0; e < s->n_edges; e++) { mapped[e].c = mapper_get(p, s->labels[e]); mapped[e].f = s->children[e]; }
        /* mapped.sort_unstable_by_key(|x| x.0): codes are distinct within a state */
        for (uint32_t a = 1; a < s->n_edges; a++) {
            cf_t x = mapped[a];
            uint32_t b = a;
            while (b > 0 && mapped[b - 1].c > x.c) { mapped[b] = mapped[b - 1]; b--; }
            mapped[b] = x;
        }
        /* find_base, :320-333 (no used-base tracking: CHECK holds the parent, bases may be shared) */
        uint32_t base = 0;
        int found = 0;
        if (h.has_head) {
            uint32_t idx = h.head_idx;
            for (;;) {
                uint32_t b = idx ^ mapped[0].c;
                int ok = b != 0;
                for (uint32_t k = 0; ok && k < s->n_edges; k++)
                    if (h_at(&h, b ^ mapped[k].c)->used_index) ok = 0;
                if (ok) { base = b; found = 1; break; }
                uint32_t next = h_at(&h, idx)->next;
                if (next == h.head_idx) break;
                idx = next;
            }
        }
        if (!found) base = (uint32_t)n ^ mapped[0].c;
        if (n <= base) { /* extend_array, :346-358 */
            if (n > (size_t)(0xffffffffu - block_len)) { rc = ORC_ERR_AUTOMATON_SCALE; goto done; }
            rc = h_push_block(&h);
            if (rc) goto done;
            if (n + block_len > cap) { cap = (n + block_len) * 2; st = (orc_cstate *)realloc(st, cap * sizeof(orc_cstate)); }
            for (size_t i = n; i < n + block_len; i++) st[i] = dflt;
            n += block_len;
        }
        for (uint32_t e = 0; e < s->n_edges; e++) {
            uint32_t child_idx = base ^ mapped[e].c;
            h_use_index(&h, child_idx);
            st[child_idx].check = state_idx;
            state_id_map[mapped[e].f] = child_idx;
            stack[sp++] = mapped[e].f;
        }
        st[state_idx].base = base;
    }
    for (size_t i = 0; i < nfa->n_states; i++) { /* :284-302 */
        if (i == 1) continue;
        uint32_t idx = state_id_map[i];
        st[idx].output_pos = nfa->states[i].output_pos;
        st[idx].fail = nfa->states[i].fail == 1 ? DEAD_IDX : state_id_map[nfa->states[i].fail];
    }
    p->states = (orc_cstate *)realloc(st, n * sizeof(orc_cstate));
    p->n_states = n;
    st = NULL;
    rc = ORC_OK;
done:
    free(st);
    free(mapped);
    free(state_id_map);
    free(stack);
    free(h.items);
    return rc;
}

/* charwise/builder.rs:178-239 */
int orc_cbuild(const uint8_t *blob, const uint64_t *offsets, const uint32_t *values, size_t n, uint8_t match_kind,
               uint32_t num_free_blocks, orc_cpma **out) {
    *out = NULL;
    if (num_free_blocks < 1 || match_kind > 2) return ORC_ERR_INVALID_ARGUMENT;
    cnfa_t nfa;
    memset(&nfa, 0, sizeof(nfa));
    nfa.match_kind = match_kind;
    orc_cpma *p = (orc_cpma *)calloc(1, sizeof(orc_cpma));
    uint32_t *freqs = NULL, *q = NULL;
    size_t n_freq = 0;
    int rc = ORC_OK;
    if (cnfa_push_state(&nfa) || cnfa_push_state(&nfa)) { rc = ORC_ERR_AUTOMATON_SCALE; goto fail; }
    for (size_t i = 0; i < n; i++) {
        uint32_t *chars;
        size_t nc;
        decode_all(blob + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), &chars, &nc);
        rc = cnfa_add(&nfa, chars, nc, values ? values[i] : (uint32_t)i);
        /* frequencies are counted for every input pattern, pruned or not (:217-229) */
        for (size_t k = 0; k < nc && !rc; k++) {
            if (n_freq <= chars[k]) {
                freqs = (uint32_t *)realloc(freqs, ((size_t)chars[k] + 1) * sizeof(uint32_t));
                memset(freqs + n_freq, 0, ((size_t)chars[k] + 1 - n_freq) * sizeof(uint32_t));
                n_freq = (size_t)chars[k] + 1;
            }
            freqs[chars[k]]++;
        }
        free(chars);
        if (rc) goto fail;
    }
    { /* CodeMapper::new, mapper.rs:16-34 */
        size_t ns = 0;
        for (size_t c = 0; c < n_freq; c++) ns += freqs[c] != 0;
        cf_t *sorted = (cf_t *)malloc((ns ? ns : 1) * sizeof(cf_t));
        ns = 0;
        for (size_t c = 0; c < n_freq; c++) if (freqs[c]) { sorted[ns].c = (uint32_t)c; sorted[ns].f = freqs[c]; ns++; }
        qsort(sorted, ns, sizeof(cf_t), cf_cmp);
        p->table = (uint32_t *)malloc((n_freq ? n_freq : 1) * sizeof(uint32_t));
        p->n_table = n_freq;
        for (size_t c = 0; c < n_freq; c++) p->table[c] = INVALID_CODE;
        for (size_t i = 0; i < ns; i++) p->table[sorted[i].c] = (uint32_t)i;
        p->alphabet_size = (uint32_t)ns;
        free(sorted);
    }
    size_t qn = 0;
    q = cnfa_build_fails(&nfa, &qn, match_kind != ORC_STANDARD);
    if (cnfa_build_outputs(&nfa, q, qn)) { rc = ORC_ERR_AUTOMATON_SCALE; goto fail; }
    rc = cbuild_double_array(p, &nfa, num_free_blocks);
    if (rc) goto fail;
    p->outputs = (orc_output *)malloc((nfa.n_outputs ? nfa.n_outputs : 1) * sizeof(orc_output));
    memcpy(p->outputs, nfa.outputs, nfa.n_outputs * sizeof(orc_output));
    p->n_outputs = nfa.n_outputs;
    p->match_kind = match_kind;
    p->num_states = (uint32_t)(nfa.n_states - 1);
    free(freqs); free(q);
    cnfa_free(&nfa);
    *out = p;
    return ORC_OK;
fail:
    free(freqs); free(q);
    cnfa_free(&nfa);
    orc_cfree_pma(p);
    return rc;
}

void orc_cfree_pma(orc_cpma *p) {
    if (!p) return;
    free(p->states); free(p->table); free(p->outputs);
    free(p);
}

size_t orc_cheap_bytes(const orc_cpma *p) { return p->n_states * 16 + p->n_table * 4 + p->n_outputs * 12; } /* charwise.rs:813-817 */

/* ================================================================ serialisation (charwise.rs:831-952) */
static void cput_u32(uint8_t **w, uint32_t x) {
    (*w)[0] = (uint8_t)x; (*w)[1] = (uint8_t)(x >> 8); (*w)[2] = (uint8_t)(x >> 16); (*w)[3] = (uint8_t)(x >> 24);
    *w += 4;
}
int orc_cserialize(const orc_cpma *p, uint8_t **buf, size_t *len) {
    size_t total = 4 + p->n_states * 16 + 4 + p->n_table * 4 + 4 + 4 + p->n_outputs * 12 + 1 + 4;
    uint8_t *b = (uint8_t *)malloc(total), *w = b;
    cput_u32(&w, (uint32_t)p->n_states);
    for (size_t i = 0; i < p->n_states; i++) {
        cput_u32(&w, p->states[i].base); cput_u32(&w, p->states[i].check); cput_u32(&w, p->states[i].fail); cput_u32(&w, p->states[i].output_pos);
    }
    cput_u32(&w, (uint32_t)p->n_table);
    for (size_t i = 0; i < p->n_table; i++) cput_u32(&w, p->table[i]);
    cput_u32(&w, p->alphabet_size);
    cput_u32(&w, (uint32_t)p->n_outputs);
    for (size_t i = 0; i < p->n_outputs; i++) { cput_u32(&w, p->outputs[i].value); cput_u32(&w, p->outputs[i].length); cput_u32(&w, p->outputs[i].parent); }
    *w++ = p->match_kind;
    cput_u32(&w, p->num_states);
    *buf = b; *len = total;
    return ORC_OK;
}

typedef struct { const uint8_t *p; size_t left; } crd_t;
static int cget_u32(crd_t *r, uint32_t *x) {
    if (r->left < 4) return -1;
    *x = (uint32_t)r->p[0] | ((uint32_t)r->p[1] << 8) | ((uint32_t)r->p[2] << 16) | ((uint32_t)r->p[3] << 24);
    r->p += 4; r->left -= 4;
    return 0;
}
static int cget_vec(crd_t *r, size_t mem, uint32_t *n) { return (cget_u32(r, n) || (uint64_t)*n * mem > r->left) ? -1 : 0; }

int orc_cdeserialize(const uint8_t *src, size_t len, orc_cpma **out, size_t *consumed) {
    *out = NULL;
    crd_t r = {src, len};
    orc_cpma *p = (orc_cpma *)calloc(1, sizeof(orc_cpma));
    uint32_t n, x;
    if (cget_vec(&r, 16, &n)) goto bad;
    p->states = (orc_cstate *)malloc((n ? n : 1) * sizeof(orc_cstate)); p->n_states = n;
    for (uint32_t i = 0; i < n; i++)
        if (cget_u32(&r, &p->states[i].base) || cget_u32(&r, &p->states[i].check) || cget_u32(&r, &p->states[i].fail) || cget_u32(&r, &p->states[i].output_pos)) goto bad;
    if (cget_vec(&r, 4, &n)) goto bad;
    p->table = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)); p->n_table = n;
    for (uint32_t i = 0; i < n; i++) if (cget_u32(&r, &p->table[i])) goto bad;
    if (cget_u32(&r, &p->alphabet_size)) goto bad;
    if (cget_vec(&r, 12, &n)) goto bad;
    p->outputs = (orc_output *)malloc((n ? n : 1) * sizeof(orc_output)); p->n_outputs = n;
    for (uint32_t i = 0; i < n; i++) if (cget_u32(&r, &p->outputs[i].value) || cget_u32(&r, &p->outputs[i].length) || cget_u32(&r, &p->outputs[i].parent)) goto bad;
    if (r.left < 1) goto bad;
    x = *r.p++; r.left--;
    p->match_kind = (x == 1) ? ORC_LEFTMOST_LONGEST : (x == 2) ? ORC_LEFTMOST_FIRST : ORC_STANDARD;
    if (cget_u32(&r, &p->num_states)) goto bad;
    /* validation, charwise.rs:912-950 */
    for (size_t i = 0; i < p->n_table; i++) if (p->table[i] != INVALID_CODE && p->table[i] >= p->alphabet_size) goto bad;
    {
        uint64_t block_len = 2;
        while (block_len < p->alphabet_size) block_len <<= 1;
        if (p->n_states == 0 || p->n_states % block_len != 0) goto bad;
    }
    for (size_t i = 0; i < p->n_states; i++) {
        if (p->states[i].base != 0 && p->states[i].base >= p->n_states) goto bad;
        if (p->states[i].fail >= p->n_states) goto bad;
        if (p->states[i].output_pos != 0 && (size_t)(p->states[i].output_pos - 1) >= p->n_outputs) goto bad;
    }
    for (size_t i = 0; i < p->n_outputs; i++) if (p->outputs[i].parent != 0 && (size_t)(p->outputs[i].parent - 1) >= i) goto bad;
    *out = p;
    if (consumed) *consumed = len - r.left;
    return ORC_OK;
bad:
    orc_cfree_pma(p);
    return ORC_ERR_INVALID_AUTOMATON;
}

/* ================================================================ transitions (charwise.rs:1022-1092) */
static inline uint32_t cnext(const orc_cpma *p, uint32_t state_id, uint32_t c) {
    uint32_t m = mapper_get(p, c);
    if (m == INVALID_CODE) return ROOT_IDX;
    for (;;) {
        const orc_cstate *s = &p->states[state_id];
        if (s->base != 0) {
            uint32_t child = s->base ^ m;
            if (p->states[child].check == state_id) return child;
        }
        if (state_id == ROOT_IDX) return ROOT_IDX;
        state_id = s->fail;
    }
}
static inline uint32_t cnext_leftmost(const orc_cpma *p, uint32_t state_id, uint32_t c) {
    uint32_t m = mapper_get(p, c);
    if (m == INVALID_CODE) return ROOT_IDX;
    for (;;) {
        const orc_cstate *s = &p->states[state_id];
        if (s->base != 0) {
            uint32_t child = s->base ^ m;
            if (p->states[child].check == state_id) return child;
        }
        if (state_id == ROOT_IDX) return ROOT_IDX;
        if (s->fail == DEAD_IDX) return ROOT_IDX;
        state_id = s->fail;
    }
}

/* ================================================================ iterators (charwise/iter.rs) */
int orc_cfind_iter(const orc_cpma *p, const uint8_t *h, size_t len, orc_match **out, size_t *n) { /* :101-157 */
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t rop = p->states[ROOT_IDX].output_pos;
    size_t i = 0;
    if (rop != 0) {
        uint32_t value = p->outputs[rop - 1].value;
        if (mv_push(&v, 0, 0, value)) return ORC_ERR_AUTOMATON_SCALE;
        while (i < len) { size_t e; decode_at(h, i, &e); i = e; if (mv_push(&v, 0, e, value)) return ORC_ERR_AUTOMATON_SCALE; }
        return mv_finish(&v, out, n);
    }
    uint32_t st = ROOT_IDX;
    while (i < len) {
        size_t e;
        uint32_t c = decode_at(h, i, &e);
        i = e;
        st = cnext(p, st, c);
        uint32_t o = p->states[st].output_pos;
        if (o != 0) {
            if (mv_push(&v, p->outputs[o - 1].length, e, p->outputs[o - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
            st = ROOT_IDX; /* every next() restarts from ROOT (:137) */
        }
    }
    return mv_finish(&v, out, n);
}

int orc_cfind_overlapping_iter(const orc_cpma *p, const uint8_t *h, size_t len, orc_match **out, size_t *n) { /* :160-221 */
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    for (uint32_t o = p->states[ROOT_IDX].output_pos; o != 0; o = p->outputs[o - 1].parent)
        if (mv_push(&v, p->outputs[o - 1].length, 0, p->outputs[o - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
    uint32_t st = ROOT_IDX;
    size_t i = 0;
    while (i < len) {
        size_t e;
        uint32_t c = decode_at(h, i, &e);
        i = e;
        st = cnext(p, st, c);
        for (uint32_t o = p->states[st].output_pos; o != 0; o = p->outputs[o - 1].parent)
            if (mv_push(&v, p->outputs[o - 1].length, e, p->outputs[o - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
    }
    return mv_finish(&v, out, n);
}

int orc_cfind_overlapping_no_suffix_iter(const orc_cpma *p, const uint8_t *h, size_t len, orc_match **out, size_t *n) { /* :224-303 */
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t rop = p->states[ROOT_IDX].output_pos;
    if (rop != 0 && mv_push(&v, 0, 0, p->outputs[rop - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
    uint32_t st = ROOT_IDX;
    size_t i = 0;
    while (i < len) {
        size_t e;
        uint32_t c = decode_at(h, i, &e);
        i = e;
        st = cnext(p, st, c);
        uint32_t o = p->states[st].output_pos;
        if (o != 0 && mv_push(&v, p->outputs[o - 1].length, e, p->outputs[o - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
    }
    return mv_finish(&v, out, n);
}

/* LeftmostFindIterator, charwise/iter.rs:306-400 */
typedef struct { const orc_cpma *p; const uint8_t *h; size_t len, pos; uint32_t init_output_pos; int skip_empty; } clm_it;
static int clm_next(clm_it *it, uint64_t *length, uint64_t *end, uint32_t *value) {
    const orc_cpma *p = it->p;
    uint32_t state_id = ROOT_IDX, last_output_pos = it->init_output_pos;
    for (;;) {
        int cont = 0;
        size_t skips = 0, i = it->pos;
        while (i < it->len) { /* self.haystack[self.pos..].chars(), the slice is taken at loop entry */
            size_t e;
            uint32_t c = decode_at(it->h, i, &e);
            size_t clen = e - i;
            i = e;
            skips += clen;
            state_id = cnext_leftmost(p, state_id, c);
            if (state_id == ROOT_IDX) {
                if (last_output_pos != 0) {
                    uint32_t output_pos = last_output_pos;
                    size_t en = it->pos;
                    if (last_output_pos == it->init_output_pos) {
                        it->pos += clen;
                        if (it->skip_empty) { it->skip_empty = 0; cont = 1; break; }
                    } else {
                        it->skip_empty = 1;
                    }
                    *length = p->outputs[output_pos - 1].length; *end = en; *value = p->outputs[output_pos - 1].value;
                    return 1;
                }
            } else if (p->states[state_id].output_pos != 0) {
                last_output_pos = p->states[state_id].output_pos;
                it->pos += skips;
                skips = 0;
            }
        }
        if (!cont) break;
    }
    if (it->pos == it->len) it->init_output_pos = 0;
    if (last_output_pos != 0) {
        *length = p->outputs[last_output_pos - 1].length; *end = it->pos; *value = p->outputs[last_output_pos - 1].value;
        return 1;
    }
    it->pos = it->len;
    return 0;
}

int orc_cleftmost_find_iter(const orc_cpma *p, const uint8_t *h, size_t len, orc_match **out, size_t *n) {
    if (p->match_kind == ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    clm_it it = {p, h, len, 0, p->states[ROOT_IDX].output_pos, 0};
    mvec v = {0};
    uint64_t l, e; uint32_t val;
    while (clm_next(&it, &l, &e, &val)) {
        if (mv_push(&v, l, e, val)) return ORC_ERR_AUTOMATON_SCALE;
        if (v.n > 2 * len + 4) { free(v.m); return ORC_ERR_DIVERGED; }
    }
    return mv_finish(&v, out, n);
}

/* FindStepper / FindOverlappingStepper driven char by char (tests/aho_corasick_crate_test.rs:446-464, 500-521) */
int orc_cfind_stepper(const orc_cpma *p, const uint8_t *h, size_t len, orc_match **out, size_t *n) { /* iter.rs:403-461 */
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t st = ROOT_IDX, root_op = p->states[ROOT_IDX].output_pos, output_pos = root_op;
    size_t pos = 0, i = 0;
    if (output_pos != 0 && mv_push(&v, p->outputs[output_pos - 1].length, 0, p->outputs[output_pos - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
    while (i < len) {
        size_t e;
        uint32_t c = decode_at(h, i, &e);
        pos += e - i;
        i = e;
        if (root_op == 0) {
            st = cnext(p, st, c);
            output_pos = p->states[st].output_pos;
            if (output_pos != 0) st = ROOT_IDX;
        }
        if (output_pos != 0 && mv_push(&v, p->outputs[output_pos - 1].length, pos, p->outputs[output_pos - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
    }
    return mv_finish(&v, out, n);
}

int orc_cfind_overlapping_stepper(const orc_cpma *p, const uint8_t *h, size_t len, orc_match **out, size_t *n) { /* iter.rs:464-534 */
    if (p->match_kind != ORC_STANDARD) return ORC_ERR_MATCH_KIND;
    mvec v = {0};
    uint32_t st = ROOT_IDX;
    size_t pos = 0, i = 0;
    for (;;) {
        for (uint32_t o = p->states[st].output_pos; o != 0; o = p->outputs[o - 1].parent)
            if (mv_push(&v, p->outputs[o - 1].length, pos, p->outputs[o - 1].value)) return ORC_ERR_AUTOMATON_SCALE;
        if (i >= len) break;
        size_t e;
        uint32_t c = decode_at(h, i, &e);
        pos += e - i;
        i = e;
        st = cnext(p, st, c);
    }
    return mv_finish(&v, out, n);
}
