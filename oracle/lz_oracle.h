/* lz_oracle.h -- CPU ORACLE for the LASTZ seed -> HSP -> gapped-extension hot path.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<
 * This is a plain-C restatement of the reference algorithm (lastz 1.04.58), written
 * from the reference's documented behaviour, each function citing the reference
 * file:line it follows.  It exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check the HIP path.  Nothing under lastz_amd/
 * (the product) may include, link, import or execute anything under oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this oracle
 * against (a) the reference's own golden vectors (test_data/base_test.*.lav, copied
 * as data fixtures into tests/golden/) and (b) stage-level outputs of the pristine
 * reference binary (oracle/_ref/lastz, built by oracle/Makefile from the sources
 * where they lie) on seeded synthetic inputs.
 *
 * Conventions follow the reference's default build: unspos=u32, sgnpos=s32,
 * score=s32 (src/sequences.h:68-75, src/dna_utilities.h:87-88).
 */
#ifndef LZ_ORACLE_H
#define LZ_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZO_MAX_PARTS   16
#define LZO_MAX_PROBES  128

/* src/dna_utilities.h:130-139 (score_type=I) */
#define LZO_WORST_SCORE   (-0x7FFFFFFF-1)
#define LZO_NEG_INF       ((int32_t)(0.9*LZO_WORST_SCORE))
#define LZO_VERY_BAD      (-((LZO_NEG_INF-LZO_WORST_SCORE)/2))

/* ---- spaced seed (output of src/seeds.c:parse_one_seed for strict seeds) ---- */
typedef struct lzo_seed {
    int      length;                    /* seed length in bases                  */
    int      weight;                    /* seed weight in bits (2 per '1')       */
    int      num_parts;
    int      shift[LZO_MAX_PARTS];
    uint32_t mask[LZO_MAX_PARTS];
    int      with_trans;                /* 0, 1 or 2                             */
    int      num_flips;
    uint32_t flips[32];                 /* seed->transFlips, in list order       */
    int      num_probes;                /* 1 + flips (+ pairs)                   */
    uint32_t probe_xor[LZO_MAX_PROBES]; /* XOR masks in the reference's probe order
                                           (src/seed_search.c:522-549)           */
} lzo_seed;

int      lzo_seed_from_pattern(const char* pattern, int with_trans, lzo_seed* out);
uint32_t lzo_apply_seed(const lzo_seed* sd, uint64_t w);

/* ---- scoring (src/dna_utilities.c:137-148,215-300,497-552) ---- */
void lzo_hoxd70(int32_t tmpl[16]);
void lzo_dna_score_set(const int32_t tmpl[16], int32_t bad, int32_t fill, int32_t* sub /*[256*256]*/);
void lzo_masked_score_set(const int32_t* sub, int32_t* out);
void lzo_upper_nuc_to_bits(int8_t tbl[256]);
double lzo_entropy(const uint8_t* s, const uint8_t* t, int len);

/* ---- position table, reference layout (src/pos_table.h:126-165) ---- */
typedef struct lzo_postable {
    uint32_t* last;         /* [word_entries] 0 = empty                          */
    uint32_t* prev;         /* [prev_entries] 0xFFFFFFFF = end of chain          */
    uint32_t  word_entries, prev_entries;
    uint32_t  start, end, adj_start, step;
    uint64_t  words_in_table;
} lzo_postable;

lzo_postable* lzo_build_position_table(const uint8_t* t, uint32_t tlen,
                                       uint32_t start, uint32_t end,
                                       const int8_t* char_to_bits,
                                       const lzo_seed* sd, uint32_t step);
void lzo_free_position_table(lzo_postable* pt);
/* flatten to CSR (word -> positions in chain order, i.e. descending) for comparing
 * with the device table */
uint64_t lzo_position_table_to_csr(const lzo_postable* pt, uint32_t* wstart /*[entries+1]*/,
                                   uint32_t* wpos /*[words_in_table]*/);

/* ---- seed hit search + diag hash + x-drop (src/seed_search.c) ---- */
typedef struct lzo_hsp {       /* what the reporter callback receives            */
    uint32_t pos1, pos2;       /* END positions (exclusive) in target / query    */
    uint32_t length;
    int32_t  score;
} lzo_hsp;

typedef struct lzo_search_stats {   /* collect_stats counters, src/seed_search.h:196-248 */
    uint64_t words;        /* "words in seq 2"  */
    uint64_t raw_hits;     /* "raw seed hits"   */
    uint64_t extensions;   /* "GF extensions"   */
    uint64_t bp_extended;  /* "bp extended"     */
    uint64_t hsps;         /* "HSPs"            */
} lzo_search_stats;

#define LZO_MODE_XDROP 0   /* process_for_simple_hit + xdrop_extend_seed_hit */
#define LZO_MODE_PLAIN 1   /* process_for_plain_hit (raw hits, no diag hash) */

int lzo_seed_hit_search(const uint8_t* t, uint32_t tlen, const lzo_postable* pt,
                        const uint8_t* q, uint32_t qlen, uint32_t start, uint32_t end,
                        const int8_t* char_to_bits, const lzo_seed* sd,
                        const int32_t* masked_sub, int32_t xdrop,
                        int32_t hsp_threshold, int entropic, int mode,
                        uint32_t diag_hash_size,
                        lzo_hsp** out, uint64_t* n_out, lzo_search_stats* stats);
void lzo_free(void* p);

/* ---- gapped stage (src/gapped_extend.c) ---- */
typedef struct lzo_segment {   /* src/segment.h:46-60 (fields used on the path) */
    uint32_t pos1, pos2, length;
    int32_t  s;
    int32_t  id;
} lzo_segment;

void lzo_reduce_to_points(const uint8_t* t, const uint8_t* q, const int32_t* sub,
                          lzo_segment* segs, uint32_t n);

typedef struct lzo_align {     /* alignel, src/edit_script.h:48-61               */
    uint32_t beg1, beg2, end1, end2;   /* 1-based, inclusive                     */
    int32_t  s;
    uint32_t script_len;               /* number of run-length ops               */
    uint32_t script_off;               /* offset into the ops array              */
} lzo_align;

typedef struct lzo_gapped_stats {
    uint64_t anchors, anchors_extended, extensions, dp_cells;
    uint32_t max_rows, max_cols;
    uint64_t truncations;
} lzo_gapped_stats;

/* ops: (count<<2)|op with op 1=ins 2=del 3=sub (src/edit_script.h:30-46) */
int lzo_gapped_extend(const uint8_t* t, uint32_t tlen, const uint8_t* q, uint32_t qlen,
                      const int32_t* sub, int32_t gap_open, int32_t gap_extend,
                      lzo_segment* anchors, uint32_t n_anchors,
                      int32_t ydrop, int trim_to_peak, int32_t score_thresh,
                      uint32_t tb_size,
                      lzo_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops,
                      lzo_gapped_stats* stats);
int lzo_gapped_extend_opts(const uint8_t* t, uint32_t tlen, const uint8_t* q, uint32_t qlen,
                           const int32_t* sub, int32_t gap_open, int32_t gap_extend,
                           lzo_segment* anchors, uint32_t n_anchors,
                           int32_t ydrop, int trim_to_peak, int all_bounds, int32_t score_thresh,
                           uint32_t tb_size,
                           lzo_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops,
                           lzo_gapped_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
