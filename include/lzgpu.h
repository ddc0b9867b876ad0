/* lzgpu.h -- C ABI of the MI355X (gfx950) drop-in for LASTZ's seed -> HSP -> gapped hot path.
 *
 * The reference (lastz 1.04.58) has no plugin API; the boundary for this path is three ordinary
 * C functions called from src/lastz.c (and src/tweener.c).  Each entry point below names the
 * reference interface it replaces; INTEGRATION.md shows the ~40-line overlay that binds the
 * reference's structs (seq, seed, postable, hitprocinfo, segtable, alignel) to these PODs.
 *
 *   B1  build_seed_position_table   src/pos_table.h:230-232   (callers src/lastz.c:1205,1211,1302)
 *   B2  seed_hit_search             src/seed_search.h:265-276 (caller  src/lastz.c:3112)
 *   B3  reduce_to_points + gapped_extend  src/gapped_extend.h:151-159 (callers src/lastz.c:3401,3419)
 *
 * Conventions (reference default build): positions are u32 (unspos), scores s32, sequences are
 * one ASCII byte per base.  All functions are called from one host thread (the reference is
 * single-threaded and non-reentrant, src/seed_search.c:364-365).  The library is a process-wide singleton: one device
 * context, one resident target / table, function-static caches (the last scoring matrix, compared by value, with its
 * class codes and look-up table; the DP arenas) -- calls from two threads, even on different sequences, are not
 * supported, with ONE exception (round 4): B3 has its own stream, timer and buffers, so one thread may be inside
 * lzgpu_gapped_extend / lzgpu_gapped_extend_batch on resident query slots while another is inside
 * lzgpu_seed_hit_search / lzgpu_query_upload on OTHER slots -- the gapped stage of unit k beside the search of unit
 * k+1 (bench.py --gpus N; src/lastz.c:3401-3419 runs them one after the other).  The exception covers RESIDENT slots only:
 * a query handed in as a host pointer is uploaded to a transient slot by the call itself, and two concurrent calls must not
 * both do that for the same sequence's sake -- upload it once (lzgpu_query_upload) and name the slot.  lzgpu_last_error() is
 * per calling thread.  Two B2 calls or two B3 calls at once remain unsupported.  The library itself uses a few worker threads for host-side loops and joins them before returning.
 *
 * Return codes, every int-returning entry point:
 *     0   done, results are complete and bit-identical to the reference's
 *    >0   not handled (an LZGPU_NH_* reason): nothing was produced; the caller runs the
 *         reference CPU routine instead (never partial results)
 *    <0   fatal (HIP error / out of memory); the reference-side binding calls suicidef()
 *         (src/utilities.c:1866-1884), matching the reference's "errors are fatal" behaviour
 * There is NO CPU fallback inside this library: without a usable gfx950 device every entry
 * point fails with LZGPU_ERR_NO_DEVICE.
 */
#ifndef LZGPU_H
#define LZGPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZGPU_MAX_PARTS   16
#define LZGPU_MAX_PROBES  128

#define LZGPU_ERR_NO_DEVICE   (-1)
#define LZGPU_ERR_HIP         (-2)
#define LZGPU_ERR_OOM         (-3)
#define LZGPU_ERR_ARG         (-4)
#define LZGPU_ERR_STATE       (-5)

#define LZGPU_NH_SEED          1   /* seed not strict / too heavy (half-weight, resolving, >28 bits) */
#define LZGPU_NH_SCORE_CLASSES 2   /* score matrix needs more than 32 row or column classes        */
#define LZGPU_NH_HITS_OVERFLOW 3   /* one query position has more raw hits than the chunk capacity */
#define LZGPU_NH_HSP_OVERFLOW  4   /* more candidate HSPs than the output capacity                 */
#define LZGPU_NH_SIZE          5   /* sequence >= 2^31 bases                                        */
#define LZGPU_NH_IDENTICAL     6   /* identical sequences (trivial self-alignment path)            */
#define LZGPU_NH_UNSUPPORTED   7   /* option outside the fast-path predicate  */
#define LZGPU_NH_PAIRED_LIMIT  8   /* the alignments pair more bases than max_paired_bases allows   */

/* Output of the reference's seed parser (struct seed, src/seeds.h:37-76) for a strict seed:
 * packed = OR_i ((w >> shift[i]) & mask[i]) (apply_seed, src/seeds.c:1335-1378), plus the XOR
 * masks to probe, in the reference's probe order (src/seed_search.c:522-549): probe 0 is the
 * exact word (xor 0), then the transition flips (and flip pairs for withTrans==2). */
typedef struct lz_seed_desc {
    int32_t  length;                       /* seed->length  (bases)                 */
    int32_t  weight_bits;                  /* seed->weight  (bits, 2 per match)     */
    int32_t  num_parts;                    /* seed->numParts                        */
    int32_t  shift[LZGPU_MAX_PARTS];       /* seed->shift[]                         */
    uint32_t mask[LZGPU_MAX_PARTS];        /* seed->mask[]                          */
    int32_t  num_probes;
    uint32_t probe_xor[LZGPU_MAX_PROBES];
} lz_seed_desc;

/* Convenience for callers without the reference's parser: compile a strict seed pattern
 * ('1' match, '0'/'X'/'x' don't care) exactly as src/seeds.c:321-640 does, with_trans in {0,1,2}
 * as set by --transition / --notransition / --twins (src/lastz.c).  Returns 0 or LZGPU_NH_SEED. */
int lzgpu_seed_from_pattern(const char* pattern, int with_trans, lz_seed_desc* out);

/* 0 if a gfx950 device is usable by this process, else LZGPU_ERR_NO_DEVICE. */
int lzgpu_probe(void);
/* lzgpu_init on a thread of its own (returns at once); any later call of the library waits for it to finish.  For a
 * host that has seconds of its own start-up work before it first needs the device (integration/lzgpu_shim.c). */
void lzgpu_init_async(int device_index);
/* Bind this process to one device (one process per GPU; LOCAL_RANK under torch.distributed).  The calling THREAD is
 * bound as well, and so is every thread that later enters the library (HIP's current device is per host thread). */
int lzgpu_init(int device_index);
int lzgpu_device_index(void);              /* the device this process is bound to, -1 before lzgpu_init */
void lzgpu_shutdown(void);
void lzgpu_free(void* p);
const char* lzgpu_last_error(void);

/* ---- B1: position table -------------------------------------------------------------------
 * Replaces build_seed_position_table(seq, start, end, upperCharToBits, seed, step)
 * (src/pos_table.c:144-196).  Uploads the target bytes t[0..tlen) (seq->v, src/sequences.h:385)
 * and builds, on the device, the table of all seed words whose window lies in [start,end)
 * (end==0 means tlen), contains only bytes with char_to_bits[b] >= 0 and ends on a multiple of
 * step (src/pos_table.c:396-476).  Device layout is CSR (word -> end positions in DESCENDING
 * order, i.e. the order the reference's last[]/prev[] chain yields them, src/pos_table.c:1341). */
int lzgpu_table_prepare(const uint8_t* t, uint32_t tlen, uint32_t start, uint32_t end,
                        const int8_t char_to_bits[256], const lz_seed_desc* seed, uint32_t step);
/* Copy the table back in the reference's own layout (postable.last[1<<weight],
 * postable.prev[1+(end-adjStart)/step], src/pos_table.h:126-165) so that host code which reads
 * the table (capsule writer, --tableonly, masking) keeps working.  Either pointer may be NULL. */
int lzgpu_table_export(uint32_t* last, uint32_t* prev);
/* Rebuild the table from the target bytes already resident in HBM (same geometry as the last
 * lzgpu_table_prepare); used to time B1 without the host->device copy. */
int lzgpu_table_rebuild(void);
uint64_t lzgpu_table_num_words(void);

/* Multi-GPU: the table is built once (rank 0) and broadcast over RCCL/xGMI by the caller, which
 * owns the communicator.  The three device buffers are plain allocations; non-root ranks call
 * lzgpu_table_adopt() with the root's geometry to allocate them, the caller broadcasts
 * buf[i] (bytes[i] each), then every rank calls lzgpu_table_commit(). */
typedef struct lz_table_geom {
    uint32_t tlen, start, end, step;
    uint64_t num_words;
    lz_seed_desc seed;
    int8_t   char_to_bits[256];
} lz_table_geom;
int lzgpu_table_geom(lz_table_geom* out);
int lzgpu_table_adopt(const lz_table_geom* geom);
int lzgpu_table_buffers(void* dev_ptr[3], uint64_t bytes[3]);   /* target bytes, wstart, wpos */
int lzgpu_table_commit(void);
/* The whole exchange in one call, for callers that have no communicator of their own (the reference-side
 * binding, integration/lzgpu_shim.c, when lastz runs as one process per GPU): every rank of `world` calls
 * lzgpu_table_share(rank, world, dir) at the same point; rank 0 must hold a table (lzgpu_table_prepare), the
 * others receive its geometry (written to <dir>/geom by rank 0), adopt it, receive the three buffers and commit.
 * Transport: RCCL over xGMI (ncclBroadcast, librccl loaded on first use; the unique id travels through
 * <dir>/nccl_id), or -- LZGPU_SHARE_TRANSPORT=file, for ranks that share one device, where RCCL refuses to
 * run -- through files in <dir> (host staging; tests only).  `dir` must be a directory all ranks see and that
 * is empty at the start of the run. */
int lzgpu_table_share(int rank, int world, const char* dir);
/* On-disk form of the same payload (the role of the reference's capsule files, src/capsule.c, whose
 * format is machine-dependent and not reused): a versioned little-endian file -- magic "LZGPUTAB", version, the
 * geometry (lz_table_geom), the three buffers (target bytes, wstart, wpos) and an FNV-1a checksum -- so that a
 * target's table is built once and loaded by later runs / other nodes.  lzgpu_table_load leaves the library in the
 * state lzgpu_table_prepare would (the target bytes come from the file); LZGPU_ERR_ARG for a file that is not a
 * table of this version, LZGPU_ERR_STATE for a damaged one. */
int lzgpu_table_save(const char* path);
int lzgpu_table_load(const char* path);
/* device-to-device copy on the library's stream (lets a caller that received the broadcast in
 * its own allocation hand it over without knowing which HIP runtime object owns the stream). */
int lzgpu_device_copy(void* dst_dev, const void* src_dev, uint64_t bytes);

/* ---- B2: seed hit search ------------------------------------------------------------------
 * Replaces seed_hit_search(seq1, pt, seq2, start, end, selfCompare, charToBits, hitSeed,
 * searchLimit, reportSearchLimit, bandWidth, processor, processorInfo) for
 * processor == process_for_simple_hit with gfExtend == gfexXDrop (src/seed_search.c:322-574,
 * 810-875, 1056-1192, 2528-2959), or process_for_plain_hit when extend==0 (:995-1029).
 * The fields mirror hitprocinfo (src/seed_search.h:112-156). */
typedef struct lz_search_args {
    const uint8_t* query;          /* seq2->v, or NULL to use a resident query slot             */
    uint32_t       qlen;           /* seq2->len                                                  */
    int32_t        query_slot;     /* used when query==NULL: slot filled by lzgpu_query_upload   */
    uint32_t       start, end;     /* search interval in the query; end==0 means qlen            */
    const int32_t* sub;            /* hp->scoring->sub (maskedScoring): [256][256], row=target   */
    int32_t        xdrop;          /* hp->xDrop                                                  */
    int32_t        hsp_threshold;  /* hp->hspThreshold.s ('S' thresholds only)                   */
    int32_t        entropic;       /* hp->entropicHsp                                            */
    int32_t        extend;         /* 1: gfexXDrop via simple-hit processor; 0: plain raw hits   */
} lz_search_args;

typedef struct lz_hsp {            /* exactly what hp->reporter receives (src/seed_search.h:62-81) */
    uint32_t pos1, pos2;           /* END of the HSP in target / query (exclusive)               */
    uint32_t length;
    int32_t  score;
} lz_hsp;

/* HSPs come back in the reference's discovery order (query position ascending, probe order,
 * target position descending).  *out is released with lzgpu_free(). */
int lzgpu_seed_hit_search(const lz_search_args* args, lz_hsp** out, uint64_t* n_out);

/* ---- B1 + B2 for many small rectangles of the two sequences at once ----------------------------
 * What src/tweener.c:769-829 (bounded_align) does for every in-between window of `lastz --inner=<score>`:
 * build_seed_position_table on target[t_off, t_off + t_len) with the inner seed, seed_hit_search of
 * query[q_off, q_off + q_len) with process_for_simple_hit / gfexXDrop / an 'S' threshold and no entropy
 * (:300-317) -- one workgroup per window, all windows in one launch.  The target is the resident one; windows
 * are at most 20480 bases a side, the seed has one probe and at most 14 bits of weight; anything else returns
 * LZGPU_NH_UNSUPPORTED before any work is done.  HSPs come back window by window in the reference's reporting
 * order, positions relative to the window; counts[k] = HSPs of window k. */
typedef struct lz_window { uint32_t t_off, t_len, q_off, q_len; } lz_window;
typedef struct lz_window_search_args {
    const uint8_t*   query;         /* seq2->v, or NULL + query_slot                                 */
    uint32_t         qlen;
    int32_t          query_slot;
    const int32_t*   sub;           /* maskedScoring->sub                                            */
    int32_t          xdrop, hsp_threshold;
    const lz_seed_desc* seed;       /* the inner seed                                                */
    const int8_t*    char_to_bits;  /* upperCharToBits[256]                                          */
    const lz_window* windows;
    uint32_t         n_windows;
} lz_window_search_args;
int lzgpu_window_search(const lz_window_search_args* args, lz_hsp** out, uint64_t* n_out, uint32_t** counts);

/* Keep a query resident in HBM across calls (bench.py: "inputs already resident"). */
int lzgpu_query_upload(int32_t slot, const uint8_t* q, uint32_t qlen);

/* ---- B3: gapped extension -----------------------------------------------------------------
 * Replaces reduce_to_points(seq1,seq2,scoring,anchors) + gapped_extend(seq1,rev1,seq2,rev2,
 * inhibitTrivial,scoring,anchors,tb,allBounds,yDrop,trimToPeak,scoreThresh,...)
 * (src/gapped_extend.c:463-559,1012-1604), 'S' thresholds, no limit on paired bases. */
typedef struct lz_segment {        /* struct segment, src/segment.h:46-60                         */
    uint32_t pos1, pos2, length;
    int32_t  s;
    int32_t  id;
} lz_segment;

typedef struct lz_gapped_args {
    const uint8_t* query;          /* seq2->v or NULL + query_slot                                */
    uint32_t       qlen;
    int32_t        query_slot;
    const int32_t* sub;            /* scoring->sub (UNmasked, src/lastz.c:3421)                   */
    int32_t        gap_open, gap_extend;   /* scoring->gapOpen, scoring->gapExtend               */
    int32_t        ydrop;
    int32_t        score_thresh;   /* gappedThreshold.s                                           */
    uint32_t       traceback_bytes;/* tb->size (0: the reference default, 80 MiB)                 */
    lz_segment*    anchors;        /* anchors->seg (HSPs; reduced to points and re-sorted in place,
                                      as the reference does)                                      */
    uint32_t       n_anchors;
    int32_t        reduce;         /* 1: run reduce_to_points first                               */
    /* partitioned sequences ("file[multi]", src/sequences.h:240-267): the positions of the NUL bytes
       that bound the partitions, ascending -- partition i lies strictly between sep[i] and sep[i+1]
       (n partitions: n+1 entries) -- or NULL.  An anchor's extension stays inside the partition
       holding it (src/gapped_extend.c:1356-1372).  The trivial alignments the reference adds for
       identical partitions (:1191-1290) are added here too; only inhibit_trivial's test by sequence
       NAME (:1485-1545) is not: a result that holds a candidate for it returns LZGPU_NH_IDENTICAL. */
    const uint32_t* sep1;  uint32_t n_sep1;
    const uint32_t* sep2;  uint32_t n_sep2;
    /* identical sequences (identical_sequences, src/gapped_extend.c:1886-1933: same length, same bases
       ignoring case, same strand flags): the trivial self-alignment goes in front of every anchor, as in
       :1152-1189, and is dropped from the output when inhibit_trivial is set (:1483).                  */
    int32_t        strands_differ; /* seq1->revCompFlags != seq2->revCompFlags                    */
    int32_t        inhibit_trivial;/* gapped_extend's inhibitTrivial                              */
    /* a rectangle of the two sequences as a problem of its own (the tweener's in-between windows,
       src/tweener.c:769-829: extract_subsequence + gapped_extend on the pieces): the stage sees
       target[t_off, t_off + t_len) x query[q_off, q_off + q_len) as its whole sequences -- anchors, sep1 / sep2
       and the alignments returned are relative to the rectangle.  t_len == 0 / q_len == 0: the whole sequence. */
    uint32_t       t_off, t_len, q_off, q_len;
    /* gapped_extend's allBounds (--allgappedbounds, src/gapped_extend.c:1411-1429): alignments below score_thresh
       still bound the later extensions and are dropped only from the list returned.  no_trim = !trimToPeak
       (--noytrim, :3747-3750, :3866): an extension that reaches the end of either sequence may end there instead
       of at its score peak.  Both 0 = lastz's defaults.                                                           */
    int32_t        all_bounds, no_trim;
    /* gapped_extend's maxPairedBases (--querydepth, src/gapped_extend.c:1441-1459; 0 = no limit): the paired bases of
       the alignments are added up as they are found; a run that stays below the limit is the run without a limit.
       The moment the limit is exceeded the call returns LZGPU_NH_PAIRED_LIMIT with nothing produced: the reference's
       routine then repeats the stage, warns and keeps / discards what it has as --querydepth asks.                */
    uint64_t       max_paired_bases;
} lz_gapped_args;

typedef struct lz_align {          /* struct alignel, src/edit_script.h:30-46                     */
    uint32_t beg1, beg2, end1, end2;   /* origin-1, inclusive                                     */
    int32_t  s;
    uint32_t script_len;           /* number of editop words                                      */
    uint32_t script_off;           /* offset of this script in *ops                               */
} lz_align;

/* Alignments in the reference's output order (increasing start in the target,
 * src/gapped_extend.c:1475-1566).  ops are the reference's editop words:
 * (repeat<<2)|op with op 1=ins 2=del 3=sub (src/edit_script.h:48-70). */
/* The target of lzgpu_gapped_extend is the sequence last given to lzgpu_table_prepare; a caller that runs
 * the gapped stage without a seed search (lastz --segments=<file>, src/lastz.c:3036-3050) uploads it with
 * lzgpu_target_upload instead (this discards any position table held on the device). */
int lzgpu_target_upload(const uint8_t* t, uint32_t tlen);
int lzgpu_gapped_extend(const lz_gapped_args* args, lz_align** out, uint64_t* n_out,
                        uint32_t** ops, uint64_t* n_ops);
/* n independent problems against the same target in one go: the two strands of a query (src/lastz.c:3401-3419 runs
 * them one after the other), several query sequences, the in-between windows of src/tweener.c.  Every problem keeps
 * the reference's anchor order and bounds of its own (the results are those of n calls of lzgpu_gapped_extend), but
 * the one-sided DPs of all of them share the launches, so that the launch of one problem does not sit out the
 * longest DP of another.  All problems must use the same scoring (sub, gap penalties, ydrop, traceback_bytes);
 * out / n_out / ops / n_ops are arrays of n, each element freed with lzgpu_free.  The return code is that of the
 * first problem that did not return 0 (then every out[k] is NULL). */
int lzgpu_gapped_extend_batch(const lz_gapped_args* args, uint32_t n, lz_align** out, uint64_t* n_out,
                              uint32_t** ops, uint64_t* n_ops);

/* ---- instrumentation (bench.py, tests) ---------------------------------------------------- */
/* ---- N2: chaining (--chain) ------------------------------------------------------------------
 * Replaces reduce_to_chain(st, diagPen, antiPen, scale, connect) (src/chain.c:497-615) with connect =
 * chain_connect_penalty (src/lastz.c:3687-3741), as try_reduce_to_chain calls it for an unpartitioned pair
 * (src/chain.c:248-250): the highest-scoring chain of the anchors under
 *     chain[i] = scale * s_i + max(0, max over j with pos1_j < pos1_i and pos2_j < pos2_i of chain[j] - connect(j, i)).
 * A HOST routine in the reference and here (each anchor needs the finished values of every earlier one):
 * it needs no device and never touches one.  Which of several equally good predecessors wins follows the
 * reference's K-d tree traversal, so the kept set is the reference's, not merely one of equal score.
 *   diag_pen / anti_pen  : the tree's pruning bounds (the reference passes chainDiag / chainAnti for them)
 *   chain_diag/chain_anti: the connection penalty per diagonal / per anti-diagonal step
 *   overlap_sub          : scoring->sub[rowChars[0]][colChars[0]], charged (times scale) per overlapped base
 * On return *kept = malloc'd indices INTO segs of the chain's members in the order the reference leaves them in the
 * table (qSegmentsByPos1, src/segment.c:1657), *n_kept their number, *best the chain's score (rounded and clipped like
 * src/chain.c:598-606, integer scores).  Free *kept with lzgpu_free.  Integer score builds only (score_type I). */
typedef struct lz_chain_args {
    int32_t diag_pen, anti_pen;
    int32_t chain_diag, chain_anti;
    int32_t scale;                 /* chainScale = 100 (src/lastz.c:318)                           */
    int32_t overlap_sub;
} lz_chain_args;
int lzgpu_reduce_to_chain(const lz_chain_args* a, const lz_segment* segs, uint32_t n, uint32_t** kept, uint32_t* n_kept, int32_t* best);
/* k independent chaining problems (the strands of a query, the units a rank has searched) with the same penalties, each on a host
 * thread of its own: a problem is serial by nature, a series of problems is not.  segs[j] / n[j] the anchors of problem j; kept / n_kept /
 * best are arrays of k (best may be NULL), each kept[j] freed with lzgpu_free.  Results are those of k calls of lzgpu_reduce_to_chain.
 * Like that call it never touches the device, so it may run on any host thread beside B2 / B3 calls. */
int lzgpu_reduce_to_chain_batch(const lz_chain_args* a, const lz_segment* const* segs, const uint32_t* n, uint32_t k,
                                uint32_t** kept, uint32_t* n_kept, int32_t* best);

typedef struct lz_counters {       /* same events as the reference's collect_stats build          */
    uint64_t words;                /* "words in seq 2"    src/seed_search.c:514                   */
    uint64_t raw_hits;             /* "raw seed hits"     src/seed_search.c:865                   */
    uint64_t extensions;           /* "GF extensions"                                             */
    uint64_t bp_extended;          /* "bp extended"       src/seed_search.c:2818                  */
    uint64_t hsps;                 /* "HSPs"                                                      */
    uint64_t dp_cells;             /* "DP cells visited"  src/gapped_extend.c:3599,3778           */
    uint64_t gapped_extensions;    /* one-sided DPs run (including speculative re-runs)           */
    uint64_t anchors_extended;
    uint64_t truncated_extensions; /* one-sided DPs that ran out of traceback space (the reference warns
                                      "truncating alignment ...", src/gapped_extend.c:3640-3661)    */
    uint64_t dp_rows;              /* rows swept by every one-sided DP that ran on the device (speculative re-runs included):
                                      the denominator of "instructions per DP row" (bench.py, tools/dp_pmc.sh)  */
} lz_counters;
void lzgpu_counters_reset(void);
int  lzgpu_counters_get(lz_counters* out);

/* Per-kernel HIP-event timing on the library's own stream.  enable!=0 brackets every launch
 * with hipEventRecord; times are accumulated per kernel name. */
void lzgpu_profile_enable(int enable);
void lzgpu_profile_reset(void);
/* n-th kernel (0-based) -> name, launches, total ms; returns 0, or 1 past the end. */
int  lzgpu_profile_get(int n, const char** name, uint64_t* launches, double* total_ms);

/* Tuning knobs (tests use small values to exercise the multi-chunk paths). */
int lzgpu_set_hit_capacity(uint64_t max_hits_per_chunk);
int lzgpu_set_hsp_capacity(uint64_t max_candidate_hsps);
int lzgpu_set_dp_slot(uint32_t first_try_traceback_bytes_per_dp);
int lzgpu_set_dp_window(uint32_t max_anchors_speculated_per_round);   /* 0: the default, 1/32 of the anchors within [2048, 16384] */
/* the one-sided DP that swept the most rows since the last reset: out = { rows, cells, shader-clock ticks of the row
 * sweep, ticks of the traceback } (a launch lasts as long as its longest DP: ticks / rows is the figure of merit of
 * k_ydrop) */
int lzgpu_dp_longest(uint64_t out[4], int reset);

/* Sharding INSIDE one query (exact alternative (1) for the HSP stage): the only cross-hit state of
 * seed_hit_search is diagEnd[hashedDiag] (src/diag_hash.h:61-64), so the 65,536 buckets can be dealt out to
 * n_owners processes: every process enumerates the table probes, but materialises, scans and extends only
 * the hits whose bucket b satisfies b % n_owners == owner.  The HSPs a process returns are those of its
 * buckets, in discovery order among themselves; lzgpu_last_hsp_order gives, for the HSPs of the LAST search,
 * two sort words each (query position << 32 | probe index, then ~target position): merging the processes'
 * lists by them reproduces the single-process list exactly.  Counters: raw_hits / extensions / bp_extended /
 * hsps are partitioned, words is replicated.  Default (1, 0) = everything. */
int lzgpu_set_bucket_owner(uint32_t n_owners, uint32_t owner);
/* Which phase-A scanner the last lzgpu_seed_hit_search used: 0 / 1 = four bases per step through the look-up
 * table on 2-bit codes, without / with masks for bytes outside A,C,G,T; 2 = the byte-code scans (matrices or
 * xDrop values outside the table's preconditions, lastz_amd/csrc/lz_lut.hpp); -1 before the first search.
 * Results are identical in every mode; lzgpu_set_scan_mode(1|2) (or LZGPU_SCAN_MODE in the environment at
 * lzgpu_init) makes every later search use at least that mode (tests run each of them). */
int lzgpu_last_scan_mode(void);
int lzgpu_set_scan_mode(int min_mode);
int lzgpu_last_hsp_order(uint64_t* out /* [2 * n] */, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* LZGPU_H */
