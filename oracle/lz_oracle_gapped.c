/* lz_oracle_gapped.c -- CPU ORACLE (test infrastructure only; see lz_oracle.h).
 *
 * Restates the gapped stage of the reference (lastz 1.04.58, src/gapped_extend.c):
 *   - reduce_to_points / segment_peak            :463-559
 *   - gapped_extend (anchor loop, non-partitioned, non-identical sequences) :1012-1604
 *   - batched_segments sort order                :1633-1675, src/segment.c:1748-1771
 *   - ydrop_align, lop_initial/final_indels      :2459-2683
 *   - ydrop_one_sided_align                      :3388-3868  (macro "prune" :2977-2987)
 *   - msp_left_right / get_above_below / align_left_right / insert_align :3953-4245
 *   - update_LR_bounds / next,prev_sweep_seg     :4588-4850
 *   - update_active_segs / build_active_seg / add_new_active / filter :4885-5130
 *   - format_alignment / save_seg                :5153-5275
 *   - score_alignment                            :5631-5680
 *   - edit scripts                               src/edit_script.c:141-424,708-801
 * Sequential and naive on purpose: this is the checker, not the product.
 */
#include "lz_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define SUB(m, r, c) ((m)[((size_t)(r) << 8) | (size_t)(c)])
#define SDIFF(a, b) (((int32_t)(a)) - ((int32_t)(b)))
#define NEG_INF LZO_NEG_INF

enum { OP_INS = 1, OP_DEL = 2, OP_SUB = 3 };
enum { DIAG_SEG = 0, HORZ_SEG = 1, VERT_SEG = 2 };
enum { C_FROM_C = 0, C_FROM_I = 1, C_FROM_D = 2, I_EXTEND = 4, D_EXTEND = 8, CID_BITS = 3 };
#define MAX_RPT ((1u << 30) - 1)

/* ------------------------------------------------------------ edit script */

typedef struct escript { uint32_t* op; uint32_t len, size, tail_op; } escript;

static escript* es_new(void)
{
    escript* s = (escript*)calloc(1, sizeof(*s));
    s->size = 12; s->op = (uint32_t*)calloc(s->size, sizeof(uint32_t));
    return s;
}
static void es_free(escript* s) { if (s) { free(s->op); free(s); } }
static void es_room(escript* s, uint32_t more)
{
    if (s->len + more + 1 <= s->size) return;
    while (s->len + more + 1 > s->size) s->size += s->size / 2 + 16;
    s->op = (uint32_t*)realloc(s->op, s->size * sizeof(uint32_t));
}
static void es_put(escript* s, uint32_t op, uint32_t rpt)
{
    es_room(s, 1);
    s->op[s->len++] = (op & 3) | (rpt << 2);
    s->tail_op = op;
}
/* src/edit_script.c:261-300 */
static void es_add(escript* s, uint32_t op, uint32_t rpt)
{
    if ((s->tail_op & 3) == op && s->len > 0) {
        uint32_t* tail = s->op + s->len - 1;
        uint32_t tr = *tail >> 2;
        if (tr + rpt <= MAX_RPT) { *tail += rpt << 2; return; }
        *tail = (op & 3) | (MAX_RPT << 2);
        rpt = tr + rpt - MAX_RPT;
    }
    while (rpt > MAX_RPT) { es_put(s, op, MAX_RPT); rpt -= MAX_RPT; }
    es_put(s, op, rpt);
}
/* src/edit_script.c:352-396 */
static void es_append(escript* dst, const escript* src)
{
    if (src->len == 0) return;
    es_room(dst, src->len + 1);
    const uint32_t* s = src->op;
    uint32_t to_copy = src->len;
    uint32_t s_op = *s & 3;
    if (dst->len > 0 && s_op == dst->tail_op) {
        uint32_t* d = dst->op + dst->len - 1;
        uint32_t dr = *d >> 2, sr = *s >> 2;
        if (dr + sr <= MAX_RPT) *d += sr << 2;
        else { *d = s_op | (MAX_RPT << 2); d[1] = s_op | ((dr + sr - MAX_RPT) << 2); dst->len++; }
        s++; to_copy--;
    }
    memcpy(dst->op + dst->len, s, to_copy * sizeof(uint32_t));
    dst->len += to_copy;
    dst->tail_op = src->tail_op;
}
static void es_reverse(escript* s)
{
    if (s->len < 2) return;
    for (uint32_t i = 0, j = s->len - 1; i < j; i++, j--) { uint32_t t = s->op[i]; s->op[i] = s->op[j]; s->op[j] = t; }
}

/* ------------------------------------------------- alignment bookkeeping */

typedef struct aseg {
    char type; uint32_t b1, b2, e1, e2;
    struct aseg *next, *prev;
} aseg;

typedef struct galn {
    uint32_t pos1, pos2, end1, end2;
    aseg *first, *last;
    escript* script; int32_t s; uint32_t beg1, beg2, aend1, aend2;   /* the alignel */
    int have_align;
    struct galn *left_align1, *right_align1, *left_align2, *right_align2;
    aseg *left_seg1, *right_seg1, *left_seg2, *right_seg2;
    struct galn *next, *prev;
} galn;

typedef struct activeseg {
    aseg* seg; uint32_t x, last_row; char type, filter; struct activeseg* next;
} activeseg;

typedef struct alignio {
    const uint8_t *seq1, *seq2, *rev1, *rev2;
    uint32_t len1, len2, low1, low2, high1, high2, anchor1, anchor2;
    const int32_t* sub; int32_t gap_open, gap_extend, ydrop; int trim_to_peak;
    uint8_t* tb; uint32_t tb_size;
    galn *left_align, *right_align; aseg *left_seg, *right_seg;
    galn *above_list, *below_list;
    int32_t s; uint32_t start1, start2, stop1, stop2; escript* script;
    lzo_gapped_stats* st;
} alignio;

/* src/gapped_extend.c:515-559 */
static uint32_t segment_peak(const uint8_t* s1, const uint8_t* s2, uint32_t len, const int32_t* sub)
{
    const uint8_t *t1 = s1, *t2 = s2;
    if (len <= 31) return len / 2;
    int32_t sim = 0; uint32_t ix;
    for (ix = 0; ix < 31; ix++) sim += SUB(sub, *t1++, *t2++);
    int32_t best = sim; uint32_t peak = 31 / 2;
    for (; ix < len; ix++) {
        sim -= SUB(sub, *s1++, *s2++);
        sim += SUB(sub, *t1++, *t2++);
        if (sim > best) { best = sim; peak = ix - 31 / 2; }
    }
    return peak;
}

void lzo_reduce_to_points(const uint8_t* t, const uint8_t* q, const int32_t* sub, lzo_segment* segs, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        uint32_t peak = segment_peak(t + segs[i].pos1, q + segs[i].pos2, segs[i].length, sub);
        segs[i].pos1 += peak; segs[i].pos2 += peak; segs[i].length = 0;
    }
}

/* src/segment.c:1748-1771 */
static int by_decreasing_score(const void* pa, const void* pb)
{
    const lzo_segment *a = (const lzo_segment*)pa, *b = (const lzo_segment*)pb;
    if (a->s != b->s)           return (a->s < b->s) ? 1 : -1;            /* score, descending */
    if (a->length != b->length) return (a->length < b->length) ? -1 : 1;  /* shorter first     */
    if (a->pos2 != b->pos2)     return (a->pos2 < b->pos2) ? -1 : 1;
    if (a->pos1 != b->pos1)     return (a->pos1 < b->pos1) ? -1 : 1;
    if (a->id != b->id)         return (a->id < b->id) ? -1 : 1;
    return 0;
}

/* src/gapped_extend.c:3953-4028 */
static int msp_left_right(galn* obi, galn* m)
{
    uint32_t pos1 = m->pos1, pos2 = m->pos2;
    uint32_t right = 0xFFFFFFFFu, left = 0xFFFFFFFFu;
    galn *m_right = NULL, *m_left = NULL; aseg *b_right = NULL, *b_left = NULL;
    for (; obi != NULL && obi->pos1 <= pos1; obi = obi->next) {
        if (obi->end1 < pos1) continue;
        aseg* bp;
        for (bp = obi->first; bp != NULL; bp = bp->next) if (bp->e1 >= pos1) break;
        if (bp == NULL) continue;
        if (bp->type == HORZ_SEG) { fprintf(stderr, "oracle: msp_left_right horizontal\n"); abort(); }
        int32_t x = (bp->type == DIAG_SEG) ? SDIFF(bp->b2, pos2) + SDIFF(pos1, bp->b1) : SDIFF(bp->b2, pos2);
        if (x == 0) return 0;
        if (x > 0 && (uint32_t)x < right) { right = (uint32_t)x; m_right = obi; b_right = bp; }
        else if (x < 0 && (uint32_t)(-x) < left) { left = (uint32_t)(-x); m_left = obi; b_left = bp; }
    }
    m->right_align1 = m->right_align2 = m_right; m->right_seg1 = m->right_seg2 = b_right;
    m->left_align1 = m->left_align2 = m_left;    m->left_seg1 = m->left_seg2 = b_left;
    return 1;
}

/* src/gapped_extend.c:4043-4059 */
static void get_above_below(alignio* io, galn* obi, galn* oed)
{
    uint32_t pos1 = io->anchor1; galn* mp;
    for (mp = oed; mp != NULL; mp = mp->prev) if (mp->end1 < pos1) break;
    io->below_list = mp;
    for (mp = obi; mp != NULL; mp = mp->next) if (mp->pos1 > pos1) break;
    io->above_list = mp;
}

/* src/gapped_extend.c:4078-4180 */
static void align_left_right(galn* obi, galn* m)
{
    uint32_t pos1 = m->pos1, pos2 = m->pos2, end1 = m->end1, end2 = m->end2;
    uint32_t rob = 0xFFFFFFFFu, rot = 0xFFFFFFFFu, lob = 0xFFFFFFFFu, lot = 0xFFFFFFFFu;
    galn *m_rob = NULL, *m_rot = NULL, *m_lob = NULL, *m_lot = NULL;
    aseg *b_rob = NULL, *b_rot = NULL, *b_lob = NULL, *b_lot = NULL;
    for (; obi != NULL; obi = obi->next) {
        if (obi->pos1 > end1 || obi->end1 < pos1) continue;
        aseg* bp; int32_t x;
        for (bp = obi->first; bp != NULL; bp = bp->next) if (bp->type != HORZ_SEG && bp->e1 >= pos1) break;
        if (bp != NULL && bp->b1 <= pos1) {
            x = (bp->type == DIAG_SEG) ? SDIFF(bp->b2, pos2) + SDIFF(pos1, bp->b1) : SDIFF(bp->b2, pos2);
            if (x > 0 && (uint32_t)x < rob) { rob = (uint32_t)x; m_rob = obi; b_rob = bp; }
            else if (x < 0 && (uint32_t)(-x) < lob) { lob = (uint32_t)(-x); m_lob = obi; b_lob = bp; }
        }
        for (; bp != NULL; bp = bp->next) if (bp->type != HORZ_SEG && bp->e1 >= end1) break;
        if (bp != NULL && bp->type != HORZ_SEG && bp->e1 >= end1) {
            x = (bp->type == DIAG_SEG) ? SDIFF(bp->b2, end2) + SDIFF(end1, bp->b1) : SDIFF(bp->b2, end2);
            if (x > 0 && (uint32_t)x < rot) { rot = (uint32_t)x; m_rot = obi; b_rot = bp; }
            else if (x < 0 && (uint32_t)(-x) < lot) { lot = (uint32_t)(-x); m_lot = obi; b_lot = bp; }
        }
    }
    m->right_align1 = m_rob; m->right_seg1 = b_rob; m->right_align2 = m_rot; m->right_seg2 = b_rot;
    m->left_align1 = m_lob;  m->left_seg1 = b_lob;  m->left_align2 = m_lot;  m->left_seg2 = b_lot;
}

/* src/gapped_extend.c:4210-4245 */
static void insert_align(galn* m, galn** p_obi, galn** p_oed)
{
    galn *mp, *mq, *obi = *p_obi, *oed = *p_oed;
    for (mq = NULL, mp = obi; mp != NULL; mq = mp, mp = mp->next) if (mp->pos1 >= m->pos1) break;
    if (mq != NULL) { mq->next = m; m->next = mp; } else { m->next = obi; obi = m; }
    for (mq = NULL, mp = oed; mp != NULL; mq = mp, mp = mp->prev) if (mp->end1 <= m->end1) break;
    if (mq != NULL) { mq->prev = m; m->prev = mp; } else { m->prev = oed; oed = m; }
    *p_obi = obi; *p_oed = oed;
}

/* src/gapped_extend.c:4754-4850 */
static int32_t next_sweep_seg(int look_right, aseg** bp, galn** mp, uint32_t row, uint32_t a1, uint32_t a2)
{
    *bp = (*bp)->next;
    if (*bp != NULL) {
        if ((*bp)->type == HORZ_SEG && (*bp = (*bp)->next) == NULL) { fprintf(stderr, "oracle: last seg horizontal\n"); abort(); }
        return SDIFF((*bp)->b2, a2);
    }
    if (look_right) { *bp = (*mp)->right_seg2; *mp = (*mp)->right_align2; }
    else            { *bp = (*mp)->left_seg2;  *mp = (*mp)->left_align2; }
    if (*bp == NULL) return 0;
    if ((*bp)->type == DIAG_SEG) return (int32_t)row + SDIFF((*bp)->b2, a2) - SDIFF((*bp)->b1, a1);
    return SDIFF((*bp)->b2, a2);
}
static int32_t prev_sweep_seg(int look_right, aseg** bp, galn** mp, uint32_t row, uint32_t a1, uint32_t a2)
{
    *bp = (*bp)->prev;
    if (*bp != NULL) {
        if ((*bp)->type == HORZ_SEG && (*bp = (*bp)->prev) == NULL) { fprintf(stderr, "oracle: first seg horizontal\n"); abort(); }
        return SDIFF(a2, (*bp)->e2);
    }
    if (look_right) { *bp = (*mp)->right_seg1; *mp = (*mp)->right_align1; }
    else            { *bp = (*mp)->left_seg1;  *mp = (*mp)->left_align1; }
    if (*bp == NULL) return 0;
    if ((*bp)->type == DIAG_SEG) return (int32_t)row + SDIFF(a2, (*bp)->e2) - SDIFF(a1, (*bp)->e1);
    return SDIFF(a2, (*bp)->e2);
}

static uint32_t special_min(uint32_t ry, int32_t r)      /* :4571-4586 */
{
    if (r <= 0) return 0;
    if ((uint32_t)r < ry) return (uint32_t)r;
    return ry;
}

/* src/gapped_extend.c:4588-4700 */
static void update_lr_bounds(int reversed, aseg** right_seg, aseg** left_seg, galn** right_align, galn** left_align,
                             uint32_t row, uint32_t a1, uint32_t a2, int32_t* pL, int32_t* pR, uint32_t* pLY, uint32_t* pRY)
{
    int32_t L = *pL, R = *pR; uint32_t LY = *pLY, RY = *pRY;
    if (!reversed) {
        if (*left_seg != NULL) {
            if ((*left_seg)->e1 >= row + a1) { if ((*left_seg)->type == DIAG_SEG) L++; }
            else L = next_sweep_seg(0, left_seg, left_align, row, a1, a2) + 1;
        }
        if (*left_seg != NULL) LY = (uint32_t)(((int32_t)LY > L) ? (int32_t)LY : L);
        if (*right_seg != NULL) {
            if ((*right_seg)->e1 >= row + a1) { if ((*right_seg)->type == DIAG_SEG) R++; }
            else R = next_sweep_seg(1, right_seg, right_align, row, a1, a2) - 1;
        }
        if (*right_seg != NULL) RY = special_min(RY, R);
    } else {
        if (*right_seg != NULL) {
            if ((*right_seg)->b1 <= a1 - row) { if ((*right_seg)->type == DIAG_SEG) L++; }
            else L = prev_sweep_seg(1, right_seg, right_align, row, a1, a2) + 1;
        }
        if (*right_seg != NULL) LY = (uint32_t)(((int32_t)LY > L) ? (int32_t)LY : L);
        if (*left_seg != NULL) {
            if ((*left_seg)->b1 <= a1 - row) { if ((*left_seg)->type == DIAG_SEG) R++; }
            else R = prev_sweep_seg(0, left_seg, left_align, row, a1, a2) - 1;
        }
        if (*left_seg != NULL) RY = special_min(RY, R);
    }
    *pL = L; *pR = R; *pLY = LY; *pRY = RY;
}

/* src/gapped_extend.c:4992-5035; mk is indexed by column (base already offset) */
static void build_active_seg(int reversed, activeseg* act, uint32_t* mk, int64_t mk_off, uint32_t row,
                             uint32_t a1, uint32_t a2, uint32_t LY, uint32_t RY)
{
    act->type = act->seg->type;
    if (!reversed) { act->x = act->seg->b2 - a2; act->last_row = act->seg->e1 - a1; }
    else           { act->x = a2 - act->seg->e2; act->last_row = a1 - act->seg->b1; }
    if (act->type != HORZ_SEG) {
        if (act->x >= LY && act->x <= RY) mk[(int64_t)act->x - mk_off] = row;
    } else {
        uint32_t horz_end = (!reversed) ? act->seg->e2 - a2 : a2 - act->seg->b2;
        uint32_t i_min = LY > act->x ? LY : act->x;
        uint32_t i_max = RY < horz_end ? RY : horz_end;
        if (i_min <= i_max)
            for (uint32_t i = i_min; i <= i_max; i++) mk[(int64_t)i - mk_off] = row;
    }
}

static void filter_active_segs(activeseg** active, int filter)
{
    activeseg *prev = NULL, *act = *active;
    while (act != NULL) {
        if (act->filter == filter) { prev = act; act = act->next; }
        else if (prev != NULL) { prev->next = act->next; free(act); act = prev->next; }
        else { *active = act->next; free(act); act = *active; }
    }
}

/* src/gapped_extend.c:4885-4965 */
static void update_active_segs(int reversed, activeseg** p_active, galn** p_list, uint32_t* mk, int64_t mk_off,
                               uint32_t row, uint32_t a1, uint32_t a2, uint32_t LY, uint32_t RY)
{
    activeseg* active = *p_active; galn* list = *p_list;
    for (activeseg* act = active; act != NULL; act = act->next) {
        if (act->last_row >= row) {
            if (act->type == DIAG_SEG) act->x++;
            if (act->x >= LY && act->x <= RY) mk[(int64_t)act->x - mk_off] = row;
        } else if ((act->seg = (reversed ? act->seg->prev : act->seg->next)) != NULL) {
            build_active_seg(reversed, act, mk, mk_off, row, a1, a2, LY, RY);
            if (act->type == HORZ_SEG) {
                act->seg = reversed ? act->seg->prev : act->seg->next;
                build_active_seg(reversed, act, mk, mk_off, row, a1, a2, LY, RY);
            }
        } else act->filter = 1;
    }
    if (!reversed) {
        while (list != NULL && list->pos1 - a1 == row) {
            activeseg* act = (activeseg*)malloc(sizeof(*act));
            act->filter = 0; act->seg = list->first; act->next = active;
            build_active_seg(reversed, act, mk, mk_off, row, a1, a2, LY, RY);
            active = act; list = list->next;
        }
    } else {
        while (list != NULL && a1 - list->end1 == row) {
            activeseg* act = (activeseg*)malloc(sizeof(*act));
            act->filter = 0; act->seg = list->last; act->next = active;
            build_active_seg(reversed, act, mk, mk_off, row, a1, a2, LY, RY);
            active = act; list = list->prev;
        }
    }
    filter_active_segs(&active, 0);
    *p_active = active; *p_list = list;
}

/* ----------------------------------------------------- the one-sided DP */

typedef struct dprow { int32_t *cc, *dd; uint32_t* mk; uint32_t len; } dprow;

static void dp_ready(dprow* d, uint32_t needed)
{
    if (needed <= d->len) return;
    uint32_t nl = needed + needed / 4 + 64;
    d->cc = (int32_t*)realloc(d->cc, nl * sizeof(int32_t));
    d->dd = (int32_t*)realloc(d->dd, nl * sizeof(int32_t));
    d->mk = (uint32_t*)realloc(d->mk, nl * sizeof(uint32_t));
    memset(d->cc + d->len, 0, (nl - d->len) * sizeof(int32_t));
    memset(d->dd + d->len, 0, (nl - d->len) * sizeof(int32_t));
    memset(d->mk + d->len, 0, (nl - d->len) * sizeof(uint32_t));
    d->len = nl;
}

static uint32_t* g_tb_row = NULL; static uint32_t g_tb_row_len = 0;
static void tbrow_needed(uint32_t n)
{
    if (n <= g_tb_row_len) return;
    g_tb_row_len = n + n / 16 + 131072;
    g_tb_row = (uint32_t*)realloc(g_tb_row, (size_t)g_tb_row_len * sizeof(uint32_t));
}

/* src/gapped_extend.c:3388-3868.  A,B are 1-based (pointer to the char before). */
static int32_t ydrop_one_sided_align(alignio* io, int reversed, const uint8_t* A, const uint8_t* B,
                                     uint32_t M, uint32_t N, escript* script, uint32_t* p_end1, uint32_t* p_end2)
{
    if (N == 0 || M == 0) { *p_end1 = *p_end2 = 0; return 0; }
    io->st->extensions++;

    const int32_t* all_sub = io->sub;
    int32_t gap_e = io->gap_extend, gap_oe = io->gap_open + gap_e, ydrop = io->ydrop;
    uint8_t* tb = io->tb; int32_t tb_len = (int32_t)io->tb_size;
    int32_t ydrop_tail;
    if (gap_e != 0) ydrop_tail = ydrop / gap_e + 6;
    else ydrop_tail = (N < 500000u) ? (int32_t)N + 1 : 500000;

    int32_t L = 0, R = (int32_t)N + 1;
    uint32_t a1 = io->anchor1, a2 = io->anchor2;
    aseg* left_seg = io->left_seg; aseg* right_seg = io->right_seg;
    if (left_seg != NULL)  { L = SDIFF(left_seg->b2, a2);  if (left_seg->type == DIAG_SEG)  L -= SDIFF(left_seg->b1, a1); }
    if (right_seg != NULL) { R = SDIFF(right_seg->b2, a2); if (right_seg->type == DIAG_SEG) R -= SDIFF(right_seg->b1, a1); }
    if (reversed) {                                                     /* note (14) */
        if (left_seg == NULL && right_seg != NULL)      { L = -R + 1; R = (int32_t)N + 1; }
        else if (left_seg != NULL && right_seg == NULL) { R = -L - 1; L = 0; }
        else if (left_seg != NULL && right_seg != NULL) { int32_t t = -L - 1; L = -R + 1; R = t; }
    }
    activeseg* active = NULL;
    galn* right_align = io->right_align; galn* left_align = io->left_align;
    galn* align_list = (!reversed) ? io->above_list : io->below_list;

    tbrow_needed(2);
    g_tb_row[0] = 0;
    uint8_t* tbp = tb;

    int32_t tb_needed = ydrop_tail;
    if (tb_needed > tb_len) { fprintf(stderr, "oracle: not enough space in trace_back array\n"); abort(); }
    dprow dyn; memset(&dyn, 0, sizeof(dyn));
    dp_ready(&dyn, (uint32_t)tb_needed);

    /* row 0 (:3576-3596) */
    uint32_t wq = 0;                       /* write index (dq) */
    int32_t c_temp = 0, c;
    dyn.cc[wq] = 0; c = dyn.dd[wq] = -gap_oe; wq++;
    *(tbp++) = 0;
    uint32_t col;
    for (col = 1; col <= N && c_temp >= -ydrop; col++) {
        dp_ready(&dyn, wq + 1);
        dyn.cc[wq] = c_temp = c;
        dyn.dd[wq] = c - gap_oe; wq++;
        c -= gap_e;
        *(tbp++) = C_FROM_I;
    }
    io->st->dp_cells += col;
    uint32_t LY = 0, RY = col;

    uint32_t end1 = 0, end2 = 0, row;
    int32_t best = 0, boundary = NEG_INF; int end_is_boundary = 0;

    for (row = 1; row <= M; row++) {
        uint32_t prev_ly = LY;
        update_lr_bounds(reversed, &right_seg, &left_seg, &right_align, &left_align, row, a1, a2, &L, &R, &LY, &RY);
        /* mask array is addressed by column: physical index = col - prev_ly */
        {   /* the reference may stamp cells up to column RY inclusive; make sure they exist */
            uint32_t top = (RY >= prev_ly) ? RY - prev_ly + 2 : 2;
            dp_ready(&dyn, top);
        }
        update_active_segs(reversed, &active, &align_list, dyn.mk, (int64_t)prev_ly, row, a1, a2, LY, RY);

        tbrow_needed(row + 2);
        if (row + 1 > io->st->max_rows) io->st->max_rows = row + 1;
        if (RY < LY) RY = LY;
        tb_needed = (int32_t)(RY - LY) + ydrop_tail;
        if ((int32_t)(tbp - tb) + tb_needed >= tb_len) { io->st->truncations++; goto dp_finished; }
        g_tb_row[row] = (uint32_t)((tbp - tb) - (int64_t)LY);

        dp_ready(&dyn, (uint32_t)tb_needed + (LY - prev_ly) + 2);
        if ((uint32_t)tb_needed > io->st->max_cols) io->st->max_cols = (uint32_t)tb_needed;
        wq = 0;                                  /* dq: writes start at col == LY */
        uint32_t rp = LY - prev_ly;              /* dp: reads  start at col == LY (prev-row indexing) */

        const int32_t* sub = all_sub + ((size_t)A[row] << 8);
        uint32_t left_col;
        col = left_col = LY;
        const uint8_t* b = B + col + 1;
        uint32_t np_col = col;
        int32_t i = NEG_INF, d; c = NEG_INF;
        uint8_t link = 0;

        for (; col < RY && (uint32_t)(b - B) <= N + 1; col++) {
            d = dyn.dd[rp];
            int pruned = 0;
            if (active != NULL && dyn.mk[rp] == row) pruned = 1;
            else if (d > c || i > c) {
                if (d >= i) { c = d; link = C_FROM_D | I_EXTEND | D_EXTEND; }
                else        { c = i; link = C_FROM_I | I_EXTEND | D_EXTEND; }
                if (c < best - ydrop) pruned = 1;
                else { i -= gap_e; dyn.dd[wq] = d - gap_e; }
            } else {
                if (c < best - ydrop) pruned = 1;
                else {
                    if (c >= best) { best = c; end1 = row; end2 = col; end_is_boundary = 0; }
                    if (!io->trim_to_peak && c >= boundary && (row == M || col == N))
                        { boundary = c; end1 = row; end2 = col; end_is_boundary = 1; }
                    int32_t c_open = c - gap_oe;
                    d -= gap_e;
                    if (c_open > d) { dyn.dd[wq] = c_open; link = C_FROM_C; }
                    else            { dyn.dd[wq] = d;      link = C_FROM_C | D_EXTEND; }
                    i -= gap_e;
                    if (c_open > i) i = c_open; else link |= I_EXTEND;
                }
            }
            if (pruned) {                                   /* macro "prune", :2977-2987 */
                c = dyn.cc[rp] + sub[*(b++)];
                if (col == LY) LY++;
                else { i = dyn.dd[wq] = dyn.cc[wq] = NEG_INF; wq++; }
                rp++;
                *(tbp++) = 0;
                continue;
            }
            np_col = col;
            int32_t c_next = dyn.cc[rp++] + sub[*(b++)];
            dyn.cc[wq++] = c;
            c = c_next;
            *(tbp++) = link;
        }
        io->st->dp_cells += col - left_col;

        if (LY >= RY) goto dp_finished;

        int32_t NN = (right_seg != NULL && R > 0) ? R - 1 : (int32_t)N;
        if (RY > np_col + 1) RY = np_col + 1;
        else {
            while (i >= best - ydrop && (int32_t)RY <= NN) {
                dp_ready(&dyn, wq + 2);
                dyn.cc[wq] = i; dyn.dd[wq] = i - gap_oe; wq++;
                i -= gap_e;
                *(tbp++) = C_FROM_I;
                RY++;
            }
        }
        if ((int32_t)RY <= NN) {
            dp_ready(&dyn, wq + 2);
            dyn.dd[wq] = dyn.cc[wq] = NEG_INF;
            RY++;
        }
    }

dp_finished:
    *p_end1 = row = end1;
    *p_end2 = col = end2;
    {
        uint8_t op, prev_op, link;
        for (prev_op = 0; row >= 1 || col > 0; prev_op = op) {
            link = tb[(uint32_t)(g_tb_row[row] + col)];      /* u32 wrap, as tbRow[row]+col */
            op = link & CID_BITS;
            if (prev_op == C_FROM_I && (link & I_EXTEND)) op = C_FROM_I;
            if (prev_op == C_FROM_D && (link & D_EXTEND)) op = C_FROM_D;
            if (op == C_FROM_I)      { col--; es_add(script, OP_INS, 1); }
            else if (op == C_FROM_D) { row--; es_add(script, OP_DEL, 1); }
            else                     { row--; col--; es_add(script, OP_SUB, 1); }
        }
    }
    filter_active_segs(&active, 2);
    free(dyn.cc); free(dyn.dd); free(dyn.mk);
    return end_is_boundary ? boundary : best;
}

/* src/gapped_extend.c:5631-5675 */
static int32_t score_alignment(const int32_t* sub, int32_t gap_open, int32_t gap_extend,
                               const uint8_t* seq1, uint32_t pos1, const uint8_t* seq2, uint32_t pos2, const escript* sc)
{
    const uint8_t *s1 = seq1 + pos1, *s2 = seq2 + pos2;
    int32_t sim = 0;
    for (uint32_t k = 0; k < sc->len; k++) {
        uint32_t rpt = sc->op[k] >> 2, op = sc->op[k] & 3;
        if (rpt == 0) continue;
        if (op == OP_SUB) { const uint8_t* stop = s1 + rpt; while (s1 < stop) sim += SUB(sub, *(s1++), *(s2++)); }
        else if (op == OP_INS) { sim -= gap_open + (int32_t)(rpt * (uint32_t)gap_extend); s2 += rpt; }
        else if (op == OP_DEL) { sim -= gap_open + (int32_t)(rpt * (uint32_t)gap_extend); s1 += rpt; }
    }
    return sim;
}

/* src/gapped_extend.c:2459-2683 */
static void ydrop_align(alignio* io)
{
    uint32_t a1 = io->anchor1, a2 = io->anchor2, end1, end2;
    io->st->anchors_extended++;
    escript* script = es_new();
    int32_t score_left = ydrop_one_sided_align(io, 1, io->rev1 + io->len1 - a1 - 2, io->rev2 + io->len2 - a2 - 2,
                                               (a1 + 1) - io->low1, (a2 + 1) - io->low2, script, &end1, &end2);
    io->start1 = a1 + 1 - end1; io->start2 = a2 + 1 - end2;
    escript* script_right = es_new();
    int32_t score_right = ydrop_one_sided_align(io, 0, io->seq1 + a1, io->seq2 + a2,
                                                io->high1 - (a1 + 1), io->high2 - (a2 + 1), script_right, &end1, &end2);
    io->stop1 = a1 + end1; io->stop2 = a2 + end2;
    es_reverse(script_right);
    es_append(script, script_right);
    es_free(script_right);
    io->s = score_right + score_left;
    io->script = script;

    if (script->len != 0) {
        if ((script->op[0] & 3) != OP_SUB) {                            /* lop_initial_indels :2589-2635 */
            uint32_t p1 = io->start1, p2 = io->start2, k;
            for (k = 0; k < script->len; k++) {
                uint32_t op = script->op[k] & 3, rpt = script->op[k] >> 2;
                if (op == OP_SUB) break; else if (op == OP_INS) p2 += rpt; else if (op == OP_DEL) p1 += rpt;
            }
            if (k == script->len) io->s = LZO_WORST_SCORE;
            else {
                io->start1 = p1; io->start2 = p2;
                script->len -= k;
                for (uint32_t j = 0; j < script->len; j++) script->op[j] = script->op[j + k];
                io->s = score_alignment(io->sub, io->gap_open, io->gap_extend, io->seq1, io->start1, io->seq2, io->start2, script);
            }
        }
        if ((script->op[script->len - 1] & 3) != OP_SUB) {              /* lop_final_indels :2640-2683 */
            uint32_t p1 = io->stop1, p2 = io->stop2, k;
            for (k = script->len; k > 0;) {
                k--;
                uint32_t op = script->op[k] & 3, rpt = script->op[k] >> 2;
                if (op == OP_SUB) { k++; break; } else if (op == OP_INS) p2 -= rpt; else if (op == OP_DEL) p1 -= rpt;
            }
            if (k == 0) io->s = LZO_WORST_SCORE;
            else {
                io->stop1 = p1; io->stop2 = p2;
                script->len = k;
                io->s = score_alignment(io->sub, io->gap_open, io->gap_extend, io->seq1, io->start1, io->seq2, io->start2, script);
            }
        }
    }
}

/* src/gapped_extend.c:5225-5275 */
static void seg_to_tail(galn* m, aseg* bp)
{
    bp->prev = m->first->prev; bp->next = m->first;
    m->first->prev->next = bp; m->first->prev = bp;
}
static void save_seg(galn* m, uint32_t b1, uint32_t b2, uint32_t e1, uint32_t e2)
{
    aseg* bp = (aseg*)malloc(sizeof(*bp));
    bp->b1 = b1; bp->b2 = b2; bp->e1 = e1; bp->e2 = e2; bp->type = DIAG_SEG;
    if (m->first == NULL) { m->first = bp->prev = bp->next = bp; return; }
    aseg* bq = (aseg*)malloc(sizeof(*bq));
    bq->type = (b1 == m->first->prev->e1 + 1) ? HORZ_SEG : VERT_SEG;
    bq->b1 = m->first->prev->e1 + 1; bq->b2 = m->first->prev->e2 + 1;
    bq->e1 = b1 - 1; bq->e2 = b2 - 1;
    seg_to_tail(m, bq); seg_to_tail(m, bp);
}

/* src/gapped_extend.c:5153-5198 */
static void format_alignment(alignio* io, galn* m)
{
    uint32_t beg1 = io->start1 + 1, end1 = io->stop1 + 1, beg2 = io->start2 + 1, end2 = io->stop2 + 1;
    escript* sc = io->script;
    uint32_t height = end1 - beg1 + 1, width = end2 - beg2 + 1, i, j, op_ix = 0;
    for (i = j = 0; i < height || j < width;) {
        uint32_t start_i = i, start_j = j, run = 0;
        while (op_ix < sc->len && (sc->op[op_ix] & 3) == OP_SUB) { run += sc->op[op_ix] >> 2; op_ix++; }
        i += run; j += run;
        save_seg(m, beg1 + start_i - 1, beg2 + start_j - 1, beg1 + i - 2, beg2 + j - 2);
        if (i < height || j < width) {
            if (op_ix < sc->len) {
                uint32_t op = sc->op[op_ix] & 3, rpt = sc->op[op_ix] >> 2;
                if (op == OP_INS) j += rpt; else if (op == OP_DEL) i += rpt;
                op_ix++;
            }
        }
    }
    m->script = sc; m->beg1 = beg1; m->beg2 = beg2; m->aend1 = end1; m->aend2 = end2; m->s = io->s;
    m->have_align = 1;
}

static void free_segs(galn* m)
{
    aseg* bp = m->first;
    while (bp != NULL) { aseg* bq = bp->next; free(bp); bp = bq; }
    m->first = m->last = NULL;
}

int lzo_gapped_extend(const uint8_t* t, uint32_t tlen, const uint8_t* q, uint32_t qlen,
                      const int32_t* sub, int32_t gap_open, int32_t gap_extend,
                      lzo_segment* anchors, uint32_t n_anchors,
                      int32_t ydrop, int trim_to_peak, int32_t score_thresh, uint32_t tb_size,
                      lzo_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops,
                      lzo_gapped_stats* stats)
{
    return lzo_gapped_extend_opts(t, tlen, q, qlen, sub, gap_open, gap_extend, anchors, n_anchors, ydrop, trim_to_peak,
                                  /* all_bounds */ 0, score_thresh, tb_size, out, n_out, ops, n_ops, stats);
}

/* all_bounds: gapped_extend's allBounds (--allgappedbounds, src/gapped_extend.c:1411-1429): an alignment below the
 * score threshold still bounds the later extensions and is dropped only when the list is written (:1475-1566) */
int lzo_gapped_extend_opts(const uint8_t* t, uint32_t tlen, const uint8_t* q, uint32_t qlen,
                           const int32_t* sub, int32_t gap_open, int32_t gap_extend,
                           lzo_segment* anchors, uint32_t n_anchors,
                           int32_t ydrop, int trim_to_peak, int all_bounds, int32_t score_thresh, uint32_t tb_size,
                           lzo_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops,
                           lzo_gapped_stats* stats)
{
    lzo_gapped_stats st; memset(&st, 0, sizeof(st));
    *out = NULL; *n_out = 0; *ops = NULL; *n_ops = 0;
    if (tlen == qlen && memcmp(t, q, tlen) == 0) return 1;   /* identical sequences: trivial
                                                                self-alignment path not restated */
    if (tb_size == 0) tb_size = 80u * 1024u * 1024u;          /* src/lastz.c:395 */

    uint8_t* rev1 = (uint8_t*)malloc((size_t)tlen + 1);       /* copy_reverse_of_string */
    uint8_t* rev2 = (uint8_t*)malloc((size_t)qlen + 1);
    for (uint32_t k = 0; k < tlen; k++) rev1[k] = t[tlen - 1 - k];
    for (uint32_t k = 0; k < qlen; k++) rev2[k] = q[qlen - 1 - k];
    rev1[tlen] = 0; rev2[qlen] = 0;

    qsort(anchors, n_anchors, sizeof(lzo_segment), by_decreasing_score);       /* :1675 */
    galn* msp = (galn*)calloc((size_t)n_anchors + 1, sizeof(galn));
    for (uint32_t k = 0; k < n_anchors; k++) { msp[k].pos1 = anchors[k].pos1; msp[k].pos2 = anchors[k].pos2; }
    st.anchors = n_anchors;

    alignio io; memset(&io, 0, sizeof(io));
    io.seq1 = t; io.seq2 = q; io.rev1 = rev1; io.rev2 = rev2;
    io.low1 = 0; io.len1 = io.high1 = tlen; io.low2 = 0; io.len2 = io.high2 = qlen;
    io.sub = sub; io.gap_open = gap_open; io.gap_extend = gap_extend; io.ydrop = ydrop;
    io.trim_to_peak = trim_to_peak;
    io.tb = (uint8_t*)malloc(tb_size); io.tb_size = tb_size; io.st = &st;

    galn *obi = NULL, *oed = NULL;
    for (uint32_t k = 0; k < n_anchors; k++) {
        galn* mp = &msp[k];
        if (!msp_left_right(obi, mp)) continue;
        io.left_align = mp->left_align1; io.right_align = mp->right_align1;
        io.left_seg = mp->left_seg1; io.right_seg = mp->right_seg1;
        io.anchor1 = mp->pos1; io.anchor2 = mp->pos2;
        get_above_below(&io, obi, oed);
        ydrop_align(&io);
        format_alignment(&io, mp);
        mp->pos1 = io.start1; mp->pos2 = io.start2; mp->end1 = io.stop1; mp->end2 = io.stop2;
        if (mp->first == NULL) { es_free(mp->script); mp->script = NULL; mp->have_align = 0; continue; }
        mp->last = mp->first->prev;
        mp->first->prev = mp->last->next = NULL;
        if (!all_bounds && mp->s < score_thresh) {                            /* :1419-1429 */
            es_free(mp->script); mp->script = NULL; mp->have_align = 0; free_segs(mp); continue;
        }
        align_left_right(obi, mp);
        insert_align(mp, &obi, &oed);
    }

    uint64_t na = 0, nops = 0;
    for (galn* mp = obi; mp != NULL; mp = mp->next) if (mp->s >= score_thresh) { na++; nops += mp->script->len; }
    lzo_align* res = (lzo_align*)malloc((na ? na : 1) * sizeof(lzo_align));
    uint32_t* op_buf = (uint32_t*)malloc((nops ? nops : 1) * sizeof(uint32_t));
    uint64_t ia = 0, io_ = 0;
    for (galn* mp = obi; mp != NULL; mp = mp->next) {
        if (mp->s < score_thresh) continue;
        res[ia].beg1 = mp->beg1; res[ia].beg2 = mp->beg2; res[ia].end1 = mp->aend1; res[ia].end2 = mp->aend2;
        res[ia].s = mp->s; res[ia].script_len = mp->script->len; res[ia].script_off = (uint32_t)io_;
        memcpy(op_buf + io_, mp->script->op, mp->script->len * sizeof(uint32_t));
        io_ += mp->script->len; ia++;
    }
    for (uint32_t k = 0; k < n_anchors; k++) { if (msp[k].script) es_free(msp[k].script); free_segs(&msp[k]); }
    free(msp); free(io.tb); free(rev1); free(rev2);
    *out = res; *n_out = na; *ops = op_buf; *n_ops = nops;
    if (stats) *stats = st;
    return 0;
}
