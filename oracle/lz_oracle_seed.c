/* lz_oracle_seed.c -- CPU ORACLE (test infrastructure only; see lz_oracle.h).
 *
 * Restates, from the reference's behaviour (lastz 1.04.58):
 *   - strict spaced-seed compilation      src/seeds.c:321-640 (parse_one_seed), :1399 (best_shift)
 *   - seed application                    src/seeds.c:1335-1378 (apply_seed)
 *   - DNA score sets                      src/dna_utilities.c:137-148,215-300,497-552
 *   - ungapped-alignment entropy          src/dna_utilities.c:2882-2940
 *   - position table build                src/pos_table.c:144-196,396-476,1042-1110,1326-1344
 *   - seed hit search                     src/seed_search.c:322-574 (private_hit_search)
 *   - table walk                          src/seed_search.c:810-875 (find_table_matches)
 *   - simple-hit processor + diag hash    src/seed_search.c:1056-1192, src/diag_hash.h:61-101
 *   - x-drop ungapped extension           src/seed_search.c:2528-2959
 * Sequential, single-threaded, deliberately naive: this is the checker.
 */
#include "lz_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ seeds */

static int bit_count32(uint32_t x) { int n = 0; while (x) { x &= x - 1; n++; } return n; }

/* src/seeds.c:1399-1417 */
static int best_shift(uint32_t uncovered, uint64_t seed_bits)
{
    int best_cov = -1, best = -1, shift;
    for (shift = 0; seed_bits != 0; seed_bits >>= 1, shift++) {
        int cov = bit_count32((uint32_t)(seed_bits & uncovered));
        if (cov > best_cov) { best_cov = cov; best = shift; }
    }
    return best;
}

uint32_t lzo_apply_seed(const lzo_seed* sd, uint64_t w)
{
    uint32_t packed = 0;
    for (int p = 0; p < sd->num_parts; p++)
        packed |= (uint32_t)(w >> sd->shift[p]) & sd->mask[p];
    return packed;
}

/* strict seeds only ('1' and '0'/'X'/'x'); src/seeds.c:321-640 with
 * maintainFlippedBitOrder defined (src/seeds.c:165,603-613) */
int lzo_seed_from_pattern(const char* pattern, int with_trans, lzo_seed* out)
{
    const char* s = pattern;
    const char* e = pattern + strlen(pattern);
    uint64_t seed_bits = 0, flip_bits = 0;
    int length = 0, weight = 0;

    memset(out, 0, sizeof(*out));
    while (s < e && (*s == '0' || *s == 'X' || *s == 'x')) s++;
    if (s >= e) return -1;
    e--;
    while (*e == '0' || *e == 'X' || *e == 'x') e--;

    for (const char* c = s; c <= e; c++) {
        if (*c == '1') {
            seed_bits = (seed_bits << 2) + 3;
            flip_bits = (flip_bits << 2) + 2;
            length++; weight += 2;
        } else if (*c == '0' || *c == 'X' || *c == 'x') {
            seed_bits <<= 2; flip_bits <<= 2; length++;
        } else
            return -2;                              /* T / half-weight: out of scope */
    }
    if (length > 31 || weight > 31 || weight == 0) return -3;

    uint32_t wbits = (uint32_t)((1ULL << weight) - 1);
    uint32_t covered = (uint32_t)(seed_bits & wbits);
    uint64_t rem = seed_bits - covered;
    int np = 1;
    out->shift[0] = 0; out->mask[0] = covered;
    while (covered != wbits) {
        int sh = best_shift((~covered) & wbits, rem);
        uint32_t mask = (uint32_t)(rem >> sh) & (~covered) & wbits;
        covered += mask;
        rem -= ((uint64_t)mask) << sh;
        if (np >= LZO_MAX_PARTS) return -4;
        out->shift[np] = sh; out->mask[np] = mask; np++;
    }
    out->num_parts = np;
    out->length = length;
    out->weight = weight;
    out->with_trans = with_trans;

    /* flips: rightmost unpacked bit first (src/seeds.c:603-613) */
    int nf = 0;
    while (flip_bits != 0) {
        uint64_t right = flip_bits - (flip_bits & (flip_bits - 1));
        flip_bits -= right;
        out->flips[nf++] = lzo_apply_seed(out, right);
    }
    out->num_flips = nf;

    /* probe order: exact word, then f1 (and f1^f2 for f2 after f1) -- src/seed_search.c:522-549 */
    int npb = 0;
    out->probe_xor[npb++] = 0;
    if (with_trans == 1) {
        for (int i = 0; i < nf; i++) out->probe_xor[npb++] = out->flips[i];
    } else if (with_trans >= 2) {
        for (int i = 0; i < nf; i++) {
            if (npb >= LZO_MAX_PROBES) return -5;
            out->probe_xor[npb++] = out->flips[i];
            for (int j = i + 1; j < nf; j++) {
                if (npb >= LZO_MAX_PROBES) return -5;
                out->probe_xor[npb++] = out->flips[i] ^ out->flips[j];
            }
        }
    }
    out->num_probes = npb;
    return 0;
}

/* ---------------------------------------------------------------- scoring */

void lzo_hoxd70(int32_t t[16])
{
    static const int32_t h[16] = {  91, -114,  -31, -123,
                                  -114,  100, -125,  -31,
                                   -31, -125,  100, -114,
                                  -123,  -31, -114,   91 };
    memcpy(t, h, sizeof(h));
}

void lzo_upper_nuc_to_bits(int8_t tbl[256])
{
    memset(tbl, -1, 256);
    tbl['A'] = 0; tbl['C'] = 1; tbl['G'] = 2; tbl['T'] = 3;
}

#define SUB(m, r, c) ((m)[((size_t)(r) << 8) | (size_t)(c)])

/* src/dna_utilities.c:215-300 (new_dna_score_set) */
void lzo_dna_score_set(const int32_t tmpl[16], int32_t bad, int32_t fill, int32_t* sub)
{
    static const char nuc[4] = { 'A', 'C', 'G', 'T' };
    for (int c = 0; c < 256; c++) SUB(sub, 0, c) = LZO_VERY_BAD;
    for (int r = 1; r < 256; r++) {
        SUB(sub, r, 0) = LZO_VERY_BAD;
        for (int c = 1; c < 256; c++) SUB(sub, r, c) = fill;
    }
    for (int c = 0; c < 256; c++)
        SUB(sub, 'X', c) = SUB(sub, 'x', c) = SUB(sub, c, 'X') = SUB(sub, c, 'x') = bad;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            int ru = nuc[r], cu = nuc[c], rl = ru + 32, cl = cu + 32;
            SUB(sub, ru, cu) = SUB(sub, ru, cl) = SUB(sub, rl, cu) = SUB(sub, rl, cl) = tmpl[r * 4 + c];
        }
}

/* src/dna_utilities.c:497-552 (masked_score_set), DNA rows and columns */
void lzo_masked_score_set(const int32_t* sub, int32_t* out)
{
    static const char lower[4] = { 'a', 'c', 'g', 't' };
    memcpy(out, sub, sizeof(int32_t) * 65536);
    int32_t bad = SUB(sub, 'A', 'X');
    for (int k = 0; k < 4; k++)
        for (int c = 1; c < 256; c++) SUB(out, lower[k], c) = bad;
    for (int c = 1; c < 256; c++) SUB(out, 'N', c) = bad;
    for (int c = 1; c < 256; c++) SUB(out, 'n', c) = bad;
    for (int c = 1; c < 256; c++) SUB(out, 'X', c) = bad;
    for (int k = 0; k < 4; k++)
        for (int r = 1; r < 256; r++) SUB(out, r, lower[k]) = bad;
    for (int r = 1; r < 256; r++) SUB(out, r, 'N') = bad;
    for (int r = 1; r < 256; r++) SUB(out, r, 'n') = bad;
    for (int r = 1; r < 256; r++) SUB(out, r, 'X') = bad;
}

/* src/dna_utilities.c:2888-2936 (compute_entropy, lowerOk=false) */
double lzo_entropy(const uint8_t* s, const uint8_t* t, int len)
{
    int cA = 0, cC = 0, cG = 0, cT = 0;
    for (int ix = 0; ix < len; ix++) {
        if (s[ix] != t[ix]) continue;
        switch (s[ix]) { case 'A': cA++; break; case 'C': cC++; break;
                         case 'G': cG++; break; case 'T': cT++; break; default: break; }
    }
    if (cA + cC + cG + cT < 20) return 1.0;
    double pA = ((double)cA) / ((double)len), pC = ((double)cC) / ((double)len);
    double pG = ((double)cG) / ((double)len), pT = ((double)cT) / ((double)len);
    double qA = (cA != 0) ? log(pA) : 0.0, qC = (cC != 0) ? log(pC) : 0.0;
    double qG = (cG != 0) ? log(pG) : 0.0, qT = (cT != 0) ? log(pT) : 0.0;
    return -(pA * qA + pC * qC + pG * qG + pT * qT) / log(4.0);
}

/* --------------------------------------------------------- position table */

#define NO_PREV 0xFFFFFFFFu

lzo_postable* lzo_build_position_table(const uint8_t* t, uint32_t tlen,
                                       uint32_t start, uint32_t end,
                                       const int8_t* ctb, const lzo_seed* sd, uint32_t step)
{
    if (step < 1) return NULL;
    if (end == 0) end = tlen;
    if (end <= start || end > tlen) return NULL;

    lzo_postable* pt = (lzo_postable*)calloc(1, sizeof(*pt));
    pt->start = start; pt->end = end; pt->step = step;
    pt->adj_start = start - (start % step);                     /* pos_table.c:1062 */
    pt->word_entries = 1u << sd->weight;
    pt->prev_entries = 1 + (end - pt->adj_start) / step;
    pt->last = (uint32_t*)calloc(pt->word_entries, sizeof(uint32_t));
    pt->prev = (uint32_t*)calloc(pt->prev_entries, sizeof(uint32_t));

    uint32_t seed_len = (uint32_t)sd->length;
    if (tlen < seed_len) return pt;

    /* pos_table.c:396-476 (record_seed_positions), written as a state machine */
    const uint8_t* s = t + start;
    const uint8_t* stop = t + end;
    while (s < stop) {
        uint64_t w = 0;
        uint32_t nts;
        int restart = 0;
        if (step > seed_len) {                                  /* :425-429 */
            uint32_t pos = (uint32_t)(s - t);
            s = s + (step - 1) - ((pos + seed_len - 1) % step);
        }
    collect:
        w = 0;
        for (nts = 1; nts < seed_len && s < stop; nts++) {
            int ww = ctb[*(s++)];
            if (ww < 0) { restart = 1; break; }
            w = (w << 2) | (uint64_t)ww;
        }
        if (restart) continue;
        while (s < stop) {
            int ww = ctb[*(s++)];
            if (ww < 0) { restart = 1; break; }
            w = (w << 2) | (uint64_t)ww;
            uint32_t pos = (uint32_t)(s - t);
            if (pos % step != 0) continue;
            uint32_t packed = lzo_apply_seed(sd, w);
            /* add_word, pos_table.c:1326-1344: prepend */
            uint32_t ix = (pos - pt->adj_start) / step;
            uint32_t old = pt->last[packed];
            pt->prev[ix] = (old == 0) ? NO_PREV : old;
            pt->last[packed] = ix;
            pt->words_in_table++;
            if (step > seed_len) { s += step - seed_len; goto collect; }   /* :468-472 */
        }
        (void)restart;
    }
    return pt;
}

void lzo_free_position_table(lzo_postable* pt)
{
    if (!pt) return;
    free(pt->last); free(pt->prev); free(pt);
}

uint64_t lzo_position_table_to_csr(const lzo_postable* pt, uint32_t* wstart, uint32_t* wpos)
{
    uint64_t n = 0;
    for (uint32_t w = 0; w < pt->word_entries; w++) {
        wstart[w] = (uint32_t)n;
        if (pt->last[w] == 0) continue;
        for (uint32_t p = pt->last[w]; p != NO_PREV; p = pt->prev[p])
            wpos[n++] = pt->adj_start + pt->step * p;
    }
    wstart[pt->word_entries] = (uint32_t)n;
    return n;
}

/* ------------------------------------------------------------ seed search */

typedef struct search_ctx {
    const uint8_t* t; uint32_t tlen;
    const uint8_t* q; uint32_t qlen;
    const lzo_postable* pt;
    const lzo_seed* sd;
    const int32_t* sub;
    int32_t xdrop, hsp_threshold, hsp_zero_threshold;
    int entropic, mode;
    uint32_t* diag_end; uint32_t diag_mask;
    lzo_hsp* out; uint64_t n_out, cap_out;
    lzo_search_stats st;
} search_ctx;

static void report(search_ctx* c, uint32_t pos1, uint32_t pos2, uint32_t length, int32_t s)
{
    if (c->n_out == c->cap_out) {
        c->cap_out = c->cap_out ? c->cap_out * 2 : 1024;
        c->out = (lzo_hsp*)realloc(c->out, c->cap_out * sizeof(lzo_hsp));
    }
    lzo_hsp* h = &c->out[c->n_out++];
    h->pos1 = pos1; h->pos2 = pos2; h->length = length; h->score = s;
}

/* src/seed_search.c:1056-1192 + :2528-2959 */
static void process_simple_hit(search_ctx* c, uint32_t pos1, uint32_t pos2, uint32_t length)
{
    const uint8_t* v1 = c->t; const uint8_t* v2 = c->q;
    int32_t diag = (int32_t)pos1 - (int32_t)pos2;                       /* diag_hash.h:61 */
    uint32_t h = ((uint32_t)diag) & c->diag_mask;                       /* diag_hash.h:62 */

    if (c->diag_end[h] == 0xFFFFFFFFu) c->diag_end[h] = 0;              /* :1097-1111 */
    if (c->diag_end[h] > pos2 - length) return;                         /* :1113 */

    c->st.extensions++;

    /* left extension (loop 1), :2598-2632 */
    uint32_t old_end = c->diag_end[h];
    int32_t block2 = (int32_t)old_end;
    const uint8_t* stop = (block2 + diag > 0) ? v1 + block2 + diag : v1;
    const uint8_t* s1 = v1 + pos1; const uint8_t* s2 = v2 + pos2;
    const uint8_t* left_start = s1;
    int32_t run = 0, left = 0;
    while (s1 > stop && run >= left - c->xdrop) {
        --s1; --s2;
        run += SUB(c->sub, *s1, *s2);
        if (run > left) { left_start = s1; left = run; }
    }
    const uint8_t* p1 = s1;

    /* right extension (loop 2), :2663-2694 */
    block2 = (int32_t)c->qlen;
    stop = ((int32_t)c->tlen <= block2 + diag) ? v1 + c->tlen : v1 + block2 + diag;
    s1 = v1 + pos1; s2 = v2 + pos2;
    const uint8_t* right_stop = s1;
    int32_t right = 0; run = 0;
    while (s1 < stop && run >= right - c->xdrop) {
        run += SUB(c->sub, *s1, *s2);
        s1++; s2++;
        if (run > right) { right_stop = s1; right = run; }
    }
    const uint8_t* right_block = s1;
    int32_t sim = left + right;

    /* record extent, :2785-2789 */
    uint32_t extent = (uint32_t)(((int32_t)(right_block - v1)) - diag);
    if (extent > c->diag_end[h]) c->diag_end[h] = extent;
    c->st.bp_extended += (uint64_t)(right_block - p1);                  /* :2818 */

    pos1 = (uint32_t)(right_stop - v1);                                 /* :2825-2827 */
    pos2 = (uint32_t)(((int32_t)pos1) - diag);
    length = (uint32_t)(right_stop - left_start);

    /* entropy adjustment, :2851-2874 (fixed score threshold) */
    if (c->entropic && sim >= c->hsp_zero_threshold && sim <= 3 * c->hsp_threshold) {
        double qq = lzo_entropy(v1 + pos1 - length, v2 + pos2 - length, (int)length);
        sim = (int32_t)(sim * qq);       /* "similarity *= q" on an s32 */
    }
    if (sim < c->hsp_threshold) return;                                 /* :2907-2933 */
    c->st.hsps++;
    report(c, pos1, pos2, length, sim);
}

/* src/seed_search.c:810-875 */
static void find_table_matches(search_ctx* c, uint32_t packed, uint32_t pos2)
{
    const lzo_postable* pt = c->pt;
    uint32_t seed_len = (uint32_t)c->sd->length;
    if (pt->last[packed] == 0) return;
    for (uint32_t pos = pt->last[packed]; pos != NO_PREV; pos = pt->prev[pos]) {
        uint32_t pos1 = pt->adj_start + pt->step * pos;
        c->st.raw_hits++;
        if (c->mode == LZO_MODE_PLAIN) report(c, pos1, pos2, seed_len, 0);  /* :995-1029 */
        else                           process_simple_hit(c, pos1, pos2, seed_len);
    }
}

int lzo_seed_hit_search(const uint8_t* t, uint32_t tlen, const lzo_postable* pt,
                        const uint8_t* q, uint32_t qlen, uint32_t start, uint32_t end,
                        const int8_t* ctb, const lzo_seed* sd,
                        const int32_t* masked_sub, int32_t xdrop,
                        int32_t hsp_threshold, int entropic, int mode,
                        uint32_t diag_hash_size,
                        lzo_hsp** out, uint64_t* n_out, lzo_search_stats* stats)
{
    search_ctx c; memset(&c, 0, sizeof(c));
    if (end == 0) end = qlen;
    if (end <= start || end > qlen) return -1;
    if (diag_hash_size == 0) diag_hash_size = 65536;                    /* diag_hash.h:56 */
    c.t = t; c.tlen = tlen; c.q = q; c.qlen = qlen; c.pt = pt; c.sd = sd;
    c.sub = masked_sub; c.xdrop = xdrop; c.hsp_threshold = hsp_threshold;
    c.hsp_zero_threshold = hsp_threshold > 0 ? hsp_threshold : 0;       /* lastz.c:2937-2939 */
    c.entropic = entropic; c.mode = mode;
    c.diag_mask = diag_hash_size - 1;
    c.diag_end = (uint32_t*)malloc(sizeof(uint32_t) * diag_hash_size);
    memset(c.diag_end, 0xFF, sizeof(uint32_t) * diag_hash_size);        /* empty_diag_hash */

    int seed_len = sd->length;
    if (qlen >= (uint32_t)seed_len) {
        /* private_hit_search, src/seed_search.c:464-574 */
        const uint8_t* qp = q + start; const uint8_t* qstop = q + end;
        while (qp < qstop) {
            uint64_t w = 0; int nts, bad = 0;
            for (nts = 1; nts < seed_len && qp < qstop; nts++) {
                int ww = ctb[*(qp++)];
                if (ww < 0) { bad = 1; break; }
                w = (w << 2) | (uint64_t)ww;
            }
            if (bad) continue;
            for (; qp < qstop; qp++) {
                int ww = ctb[*qp];
                if (ww < 0) break;          /* "goto empty": qp is NOT advanced past the bad char;
                                               the collect loop's *(q++) consumes it */
                w = (w << 2) | (uint64_t)ww;
                uint32_t pos2 = (uint32_t)(qp - q) + 1;
                uint32_t packed = lzo_apply_seed(sd, w);
                c.st.words++;
                for (int p = 0; p < sd->num_probes; p++)
                    find_table_matches(&c, packed ^ sd->probe_xor[p], pos2);
            }
        }
    }
    free(c.diag_end);
    *out = c.out; *n_out = c.n_out;
    if (stats) *stats = c.st;
    return 0;
}

void lzo_free(void* p) { free(p); }
