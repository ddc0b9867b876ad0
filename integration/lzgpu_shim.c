/* lzgpu_shim.c -- the reference-side binding of liblzgpu.so.
 *
 * This is the one file a lastz maintainer adds.  It is written against the REFERENCE's own
 * headers (the src/ .h files, found with -I at build time; nothing of the reference is copied here) and
 * redefines the three hot-path entry points.  The reference's definitions of the same functions
 * are kept, renamed ref_* by -D flags on the three files that define them (integration/Makefile),
 * and serve every case outside the fast-path predicate -- never as a silent substitute for a
 * failing GPU path: a negative return code from the library is fatal (suicidef), exactly like
 * every other error in the reference (src/utilities.c:1866-1884).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <stdbool.h>
#include <unistd.h>
#include <time.h>
#include <stdio_ext.h>
#include "build_options.h"
#include "utilities.h"
#include "dna_utilities.h"
#include "sequences.h"
#include "seeds.h"
#include "pos_table.h"
#include "segment.h"
#include "edit_script.h"
#include "seed_search.h"
#include "gapped_extend.h"
#include "diag_hash.h"
#include "capsule.h"
#include "quantum.h"
#include "tweener.h"
#include "chain.h"
#include "lzgpu.h"
#include <sys/stat.h>
#include <errno.h>

/* the reference's own routines, renamed at compile time */
postable* ref_build_seed_position_table (seq* seq, unspos start, unspos end, const s8 upperCharToBits[], seed* seed, u32 step);
void      ref_free_position_table (postable* pt);
void      ref_mask_seed_position_table (postable* pt, seq* seq, unspos start, unspos end, const s8 upperCharToBits[], seed* hitSeed);
void      ref_limit_position_table (postable* pt, u32 limit, unspos maxChasm);
u64       ref_seed_hit_search (seq* seq1, postable* pt, seq* seq2, unspos start, unspos end, int selfCompare,
                               const s8 upperCharToBits[], seed* hitSeed, u32 searchLimit, u32 reportSearchLimit,
                               u32 bandWidth, hitprocessor processor, void* processorInfo);
alignel*  ref_gapped_extend (seq* seq1, u8* rev1, seq* seq2, u8* rev2, int inhibitTrivial, scoreset* scoring,
                             segtable* anchors, tback* tb, int allBounds, score yDrop, int trimToPeak,
                             sthresh scoreThresh, u64 maxPairedBases, int overlyPairedWarn, int overlyPairedKeep);
/* host-side readers of the table (src/pos_table.h:245-250, src/capsule.h:315, src/quantum.h:118), renamed likewise:
   the host copy of a table built on the device is filled in only when one of them (or a reference fall-back) is
   about to read it */
void      ref_dump_position_table (FILE* f, postable* pt, seed* hitSeed, int showPositions, int showCounts);
unspos    ref_count_position_table (postable* pt);
poscount* ref_position_table_count_distribution (postable* pt);
u32       ref_find_position_table_limit (postable* pt, float keep);
u64       ref_write_capsule_file (FILE* f, char* filename, seq* seq, u8* revNucs, postable* pt, seed* seed);
u32       ref_quantum_seed_hit_search (seq* seq1, postable* pt, seq* seq2, unspos start, unspos end,
                                       const s8 charToBits[], seed* hitSeed, scoreset* scoring, score ballScore,
                                       hitprocessor processor, void* processorInfo);
score     ref_try_reduce_to_chain (seq* seq1, seq* seq2, segtable* st, score diagPen, score antiPen, int scale, chainer connect);
alignel*  ref_tweener_interpolate (alignel* a, seq* seq1, seq* seq2, int selfCompare, int inhibitTrivial,
                                   const s8 charToBits[], seed* tweenSeed, scoreset* scoring, scoreset* maskedScoring,
                                   tback* tb, score xDrop, int gappedAllBounds, score yDrop, int trimToPeak, score scoreThresh,
                                   score diagPen, score antiPen, int scale, chainer connect, u32 windowSize);

/* which host objects the device copy currently mirrors */
static postable* devTable   = NULL;
static int       devTableOnHost = false;                   /* devTable->last / ->prev hold the table (lzgpu_table_export ran) */
static u8*       devTargetV = NULL;
static unspos    devTargetLen = 0;
static s8        devCharToBits[256];
static int       shimVerbose = -1;

static unspos min_target (void)
	{
	static long v = -1;
	if (v < 0) { char* e = getenv ("LZGPU_MIN_TARGET");  v = (e != NULL)? atol(e) : 10000; }
	return (unspos) v;
	}

static void note (const char* what, const char* how)
	{
	if (shimVerbose < 0) shimVerbose = (getenv ("LZGPU_VERBOSE") != NULL);
	if (shimVerbose) fprintf (stderr, "[lzgpu] %s: %s\n", what, how);
	if (getenv ("LZGPU_VERBOSE_CLOCK") != NULL)                  /* where the process's wall time goes (tools/cli_prof.sh) */
		{
		struct timespec ts;
		clock_gettime (CLOCK_MONOTONIC, &ts);
		fprintf (stderr, "[lzgpu clock] %.3f s (monotonic) after %s: %s\n", ts.tv_sec % 100000 + ts.tv_nsec * 1e-9, what, how);
		}
	}

/* ---- the reference reads its sequence files one getc() at a time (seq_getc, src/sequences.c).  glibc takes the
 * stream's lock on every such call once the process has a second thread, and the HIP runtime starts several: the
 * 50 Mbp query of the bench pair -- read after the device came up for the table -- took 1.1 s to load against 0.35 s
 * for the target of the same size, read before.  The reference touches its files from main()'s thread only, so
 * every stream it opens (and standard output) is switched to caller-side locking. */
FILE* ref_fopen_or_die (const char* name, const char* mode);
FILE* fopen_or_die (const char* name, const char* mode)
	{
	FILE* f = ref_fopen_or_die (name, mode);
	if ((f != NULL) && (getenv ("LZGPU_LOCKED_STDIO") == NULL)) __fsetlocking (f, FSETLOCKING_BYCALLER);
	return f;
	}

__attribute__((constructor)) static void unlocked_stdout (void)
	{
	if (getenv ("LZGPU_LOCKED_STDIO") == NULL) __fsetlocking (stdout, FSETLOCKING_BYCALLER);
	/* lastz is a process that lives a second or two.  The library's default of 2^31 hits per chunk (a 50 Mbp strand in
	 * one chunk) means 41 GiB of chunk buffers; what a process frees at exit the driver clears before it hands it out
	 * again, and lastz runs back to back then wait for that: with chunks of 2^30 (20 GiB) the first two runs on a fresh box
	 * spent 1.0 and 0.8 s in hipMalloc (2.3 and 2.1 s wall instead of 1.32) and every fourth run of a series 4.3 s.  Chunks
	 * of 2^28 (5.6 GiB) cost 15 ms more per search and bound that wait at a quarter (tools/cli_caps.sh, round 4). */
	if (getenv ("LZGPU_HIT_CAPACITY") == NULL) lzgpu_set_hit_capacity (1ull << 28);
	}

/* ... and a chunk for the search at hand: about an eighth of the hits it should find (random sequence: probes x target
 * words x query words / 4^weight), between 2^26 and 2^30 -- the 200 Mbp pair in 2^28 chunks is 231 chunks a strand,
 * 7.1-8.0 s for the run against 6.4 s with 2^30 */
static void chunk_for (seed* hitSeed, unspos tLen, unspos qLen)
	{
	double est;
	int    probes = 1, nf = 0, lg;
	if (getenv ("LZGPU_HIT_CAPACITY") != NULL) return;
	if ((hitSeed->withTrans != 0) && (hitSeed->transFlips != NULL)) while (hitSeed->transFlips[nf] != 0) nf++;
	if (hitSeed->withTrans == 1) probes = 1 + nf;  else if (hitSeed->withTrans >= 2) probes = 1 + nf + nf * (nf - 1) / 2;
	est = ((double) probes) * ((double) tLen) * ((double) qLen) / ((double) (1ull << hitSeed->weight)) / 8.0;
	for (lg=26 ; (lg < 30) && (((double) (1ull << lg)) < est) ; lg++) ;
	lzgpu_set_hit_capacity (1ull << lg);
	}

/* a clock line without a note: entry points of the stages, for tools/cli_prof.sh */
static void clock_mark (const char* what, const char* how)
	{
	struct timespec ts;
	if (getenv ("LZGPU_VERBOSE_CLOCK") == NULL) return;
	clock_gettime (CLOCK_MONOTONIC, &ts);
	fprintf (stderr, "[lzgpu clock] %.3f s (monotonic) at %s: %s\n", ts.tv_sec % 100000 + ts.tv_nsec * 1e-9, what, how);
	}

/* The HIP runtime and the device context take 0.2-0.3 s to come up; lastz spends that long parsing the target
 * file before it first needs the device.  With two file arguments on the command line the library starts its
 * initialisation on a thread of its own while main() is still reading options (lzgpu_init_async; every entry
 * point waits for it): 1.36 against 1.49 s wall on the 50 Mbp pair.  LZGPU_EARLY_INIT=0 turns it off.  (Before the
 * reference's streams were switched to caller-side locking, above, this LOST 0.6 s: the runtime's threads exist
 * earlier, so the target file too was read through locked getc() calls.) */
__attribute__((constructor)) static void early_device_start (int argc, char** argv, char** envp)
	{
	int k, files = 0;
	char* e = getenv ("LZGPU_EARLY_INIT");
	(void) envp;
	if ((e != NULL) && (e[0] == '0')) return;
	for (k=1 ; k<argc ; k++)
		{
		if (strncmp (argv[k], "--help", 6) == 0) return;
		if (strncmp (argv[k], "--version", 9) == 0) return;
		if (argv[k][0] != '-') files++;
		}
	if (files >= 2) lzgpu_init_async (-1);
	}

static void drop_device_table (void) { devTable = NULL;  devTargetV = NULL;  devTargetLen = 0;  devTableOnHost = false; }

/* The host copy in the reference's layout (last[] / prev[]) costs a device-to-host copy of 64 MiB + 4 bytes per
 * target base; the search and the gapped stage never read it.  It is made on demand: before any reference routine
 * that reads a position table runs on the table the device built. */
static void host_table_needed (postable* pt)
	{
	int rc;
	if ((pt == NULL) || (pt != devTable) || (devTableOnHost)) return;
	rc = lzgpu_table_export (pt->last, pt->prev);
	if (rc != 0) suicidef ("lzgpu_table_export: %s", lzgpu_last_error());
	devTableOnHost = true;
	note ("table", "copied to the host for a reference routine");
	}

/* ---- one process per GPU (lastz_amd/multi.py launches them): LZGPU_RANK / LZGPU_WORLD name this process,
 * LZGPU_SHARE_DIR is a directory all ranks see (the table rendezvous), LZGPU_UNIT_PLAN a text file of
 * "contig strand rank" lines that deals the (query sequence, strand) units out to the ranks.  Every rank runs the
 * same command on the same files; a unit a rank does not own yields no HSPs there, so that its output holds the
 * stanzas of its own units only, and the launcher merges the ranks' outputs in file order (queries in file
 * order, + strand before - strand: src/lastz.c:1592-1691). */
static int    mgRank = -1, mgWorld = 1, mgTables = 0;
static char*  mgDir = NULL;
static int*   mgPlan = NULL;   static u32 mgPlanLen = 0;        /* [2*(contig-1) + strand] -> rank */

static void multi_init (void)
	{
	char* e;  FILE* f;  unsigned c, st, r;
	if (mgRank >= 0) return;
	mgRank = 0;
	if ((e = getenv ("LZGPU_WORLD")) != NULL) mgWorld = atoi (e);
	if (mgWorld < 1) mgWorld = 1;
	if ((e = getenv ("LZGPU_RANK")) != NULL) mgRank = atoi (e);
	if ((mgRank < 0) || (mgRank >= mgWorld)) suicidef ("LZGPU_RANK=%d is not in [0,%d)", mgRank, mgWorld);
	mgDir = getenv ("LZGPU_SHARE_DIR");
	if (((mgWorld > 1) || (getenv ("LZGPU_SHARE_FORCE") != NULL)) && (mgDir == NULL)) suicide ("LZGPU_WORLD > 1 needs LZGPU_SHARE_DIR");
	if ((mgWorld > 1) && ((e = getenv ("LZGPU_UNIT_PLAN")) != NULL))
		{
		f = fopen_or_die (e, "rt");
		while (fscanf (f, "%u %u %u", &c, &st, &r) == 3)
			{
			u32 ix = 2*(c-1) + (st & 1);
			if ((c < 1) || (r >= (unsigned) mgWorld)) suicidef ("bad line in %s", e);
			if (ix >= mgPlanLen)
				{
				u32 n = 2*ix + 16, k;
				mgPlan = (int*) realloc_or_die ("lzgpu unit plan", mgPlan, n * sizeof(int));
				for (k=mgPlanLen ; k<n ; k++) mgPlan[k] = -1;
				mgPlanLen = n;
				}
			mgPlan[ix] = (int) r;
			}
		fclose_if_valid (f);
		}
	}

static int unit_is_mine (seq* seq2)
	{
	u32 contig, ix;
	multi_init ();
	if (mgWorld <= 1) return true;
	contig = (seq2->contig >= 1)? seq2->contig : 1;
	ix = 2*(contig-1) + (((seq2->revCompFlags & rcf_rev) != 0)? 1 : 0);
	if ((ix < mgPlanLen) && (mgPlan[ix] >= 0)) return (mgPlan[ix] == mgRank);
	return ((int) (ix % (u32) mgWorld) == mgRank);                      /* no plan: round robin */
	}

/* LZGPU_UNIT_MARKERS=1 (the launcher sets it for the line-oriented output formats): a comment line on standard
 * output names the unit whose records follow, so that the launcher can put the ranks' records back in file order
 * without parsing sequence names.  The reference writes a unit's records after this point and before the next
 * unit's search (src/lastz.c:3053-3452), through the same stdio buffer. */
static void unit_marker (seq* seq2)
	{
	static int want = -1;
	if (want < 0) { char* e = getenv ("LZGPU_UNIT_MARKERS");  want = ((e != NULL) && (e[0] == '1')); }
	if (!want) return;
	{
	/* one marker per unit: a (contig, strand) that is searched more than once in a row (chore files) keeps its marker */
	static u32 lastContig = 0;  static int lastRev = -1;
	u32 contig = (seq2->contig >= 1)? seq2->contig : 1;
	int rev    = ((seq2->revCompFlags & rcf_rev) != 0)? 1 : 0;
	if ((contig == lastContig) && (rev == lastRev)) return;
	lastContig = contig;  lastRev = rev;
	fprintf (stdout, "#lzgpu-unit %u %d\n", contig, rev);
	}
	}

static int fast_seed (seed* hitSeed, lz_seed_desc* sd)
	{
	int i, j, nf, np;
	if ((hitSeed->type != 'S') || (hitSeed->isHalfweight) || (hitSeed->revComp)
	 || (hitSeed->next != NULL) || (hitSeed->weight > 28) || (hitSeed->numParts > LZGPU_MAX_PARTS))
		return false;
	memset (sd, 0, sizeof(*sd));
	sd->length = hitSeed->length;  sd->weight_bits = hitSeed->weight;  sd->num_parts = hitSeed->numParts;
	for (i=0 ; i<hitSeed->numParts ; i++) { sd->shift[i] = hitSeed->shift[i];  sd->mask[i] = hitSeed->mask[i]; }
	nf = 0;
	if ((hitSeed->withTrans != 0) && (hitSeed->transFlips != NULL))
		while (hitSeed->transFlips[nf] != 0) nf++;
	np = 0;
	sd->probe_xor[np++] = 0;                                   /* src/seed_search.c:522-549 */
	if (hitSeed->withTrans == 1)
		{ for (i=0 ; i<nf ; i++) { if (np >= LZGPU_MAX_PROBES) return false;  sd->probe_xor[np++] = hitSeed->transFlips[i]; } }
	else if (hitSeed->withTrans >= 2)
		{
		for (i=0 ; i<nf ; i++)
			{
			if (np >= LZGPU_MAX_PROBES) return false;
			sd->probe_xor[np++] = hitSeed->transFlips[i];
			for (j=i+1 ; j<nf ; j++)
				{ if (np >= LZGPU_MAX_PROBES) return false;  sd->probe_xor[np++] = hitSeed->transFlips[i] ^ hitSeed->transFlips[j]; }
			}
		}
	sd->num_probes = np;
	return true;
	}

/* the NUL separators bounding a [multi] sequence's partitions (src/sequences.h:240-267), NULL if it has none */
static uint32_t* partition_separators (seq* s, uint32_t* n)
	{
	seqpartition* sp = &s->partition;
	uint32_t*     v;
	u32           i;

	*n = 0;
	if (sp->p == NULL) return NULL;
	v = (uint32_t*) malloc_or_die ("lzgpu separators", ((size_t) sp->len + 1) * sizeof(uint32_t));
	for (i=0 ; i<=sp->len ; i++) v[i] = sp->p[i].sepBefore;      /* (entry len: the final NUL) */
	*n = sp->len + 1;
	return v;
	}

/* ---- the tweener's in-between windows (lastz --inner=<score>, src/tweener.c), SURVEY 8(f) N3 ----
 * tweener_interpolate walks the outer alignments and, for every gap, calls (through bounded_align, a static function)
 * build_seed_position_table, seed_hit_search, reduce_to_chain and gapped_extend on copies of the two pieces.  The
 * windows depend on the OUTER alignments only -- what is found inside one window never changes another (innerList is
 * merged in at the very end, :465) -- so the reference routine is simply run three times with the three entry points
 * in different roles, and all windows of the (query, strand) go to the device together:
 *   pass 1  record   the hooks note every window (no HSPs are reported, so nothing is chained or extended)
 *           -> lzgpu_window_search: table + search of all windows in one launch
 *   pass 2  replay the HSPs, record the anchors the reference's own chaining hands to gapped_extend
 *           -> lzgpu_gapped_extend_batch: the windows' DPs share their launches
 *   pass 3  replay HSPs and alignments: the reference merges them as it always does
 * A window is a copy (extract_subsequence, :1052): where it lies in the two sequences is found from the outer
 * alignments (every window starts at the end of one or ends at the start of one, :424-447, :1016-1023) and
 * checked by comparing the bytes, so a wrong guess is impossible and two places with equal bytes are equally good. */
enum { twOff = 0, twRecord, twReplaySearch, twReplayAll };
typedef struct twWindow
	{
	u32        tOff, tLen, qOff, qLen;
	u32        hspStart, hspCount;      /* in twHsps */
	lz_segment* anchors;  u32 numAnchors;
	int        strandsDiffer;
	lz_align*  aligns;  uint64_t numAligns;  uint32_t* ops;
	} twWindow;
static int       twMode = twOff, twFailed = false;
static twWindow* twWin = NULL;   static u32 twNum = 0, twCap = 0, twCursor = 0;
static lz_hsp*   twHsps = NULL;
static alignel*  twOuter = NULL; static alignel* twHint = NULL;
static seq*      twSeq1 = NULL;  static seq* twSeq2 = NULL;
static hitprocinfo twHp;         static int twHaveHp = false;
static const s8* twCharToBits = NULL;
static struct { scoreset* scoring; int inhibitTrivial; score yDrop; sthresh scoreThresh; u32 tbSize; int have; } twGap;

static int window_offsets (seq* w1, seq* w2, u32* tOff, u32* qOff)
	{
	alignel* a;  int pass, lap;
	unspos   len1 = w1->len, len2 = w2->len;
	for (lap=0 ; lap<2 ; lap++)                                 /* from the last match on, then from the start */
		for (a=(lap==0)?twHint:twOuter ; a!=NULL ; a=a->next)
			{
			if ((lap == 1) && (a == twHint)) break;
			for (pass=0 ; pass<2 ; pass++)
				{
				unspos o1, o2;
				if (pass == 0)                                  /* the window ends where a begins */
					{ if ((a->beg1 < len1) || (a->beg2 < len2)) continue;  o1 = a->beg1 - len1;  o2 = a->beg2 - len2; }
				else                                            /* the window begins where a ends */
					{ if ((a->end1 < 1) || (a->end2 < 1)) continue;  o1 = a->end1 - 1;  o2 = a->end2 - 1; }
				if ((o1 + len1 > twSeq1->len) || (o2 + len2 > twSeq2->len)) continue;
				if ((memcmp (twSeq1->v + o1, w1->v, len1) != 0) || (memcmp (twSeq2->v + o2, w2->v, len2) != 0)) continue;
				*tOff = (u32) o1;  *qOff = (u32) o2;  twHint = a;
				return true;
				}
			}
	return false;
	}

static void tw_free (void)
	{
	u32 k;
	for (k=0 ; k<twNum ; k++) { free (twWin[k].anchors);  lzgpu_free (twWin[k].aligns);  lzgpu_free (twWin[k].ops); }
	free (twWin);  twWin = NULL;  twNum = twCap = twCursor = 0;
	lzgpu_free (twHsps);  twHsps = NULL;
	twMode = twOff;  twFailed = false;  twHaveHp = false;  twGap.have = false;
	}

static alignel* alignels_from (lz_align* al, uint64_t n, uint32_t* ops, seq* seq1, seq* seq2)
	{
	alignel* head = NULL, *last = NULL, *el;  uint64_t k;  u32 j;
	for (k=0 ; k<n ; k++)                                      /* increasing start, src/gapped_extend.c:1475-1566 */
		{
		el = (alignel*) malloc_or_die ("lzgpu alignel", sizeof(alignel));
		el->next = NULL;  el->isTrivial = false;  el->hspId = 0;
		el->beg1 = al[k].beg1;  el->beg2 = al[k].beg2;  el->end1 = al[k].end1;  el->end2 = al[k].end2;
		el->s = al[k].s;  el->seq1 = seq1->v;  el->seq2 = seq2->v;
		el->script = edit_script_new ();
		for (j=0 ; j<al[k].script_len ; j++)
			{
			uint32_t w = ops[al[k].script_off + j];
			edit_script_add (&el->script, edit_op_operation(w), edit_op_repeat(w));
			}
		if (head == NULL) head = last = el;  else { last->next = el;  last = el; }
		}
	return head;
	}

/* ---- B1 ---- */

postable* build_seed_position_table
   (seq* seq, unspos start, unspos end, const s8 upperCharToBits[], seed* hitSeed, u32 step)
	{
	lz_seed_desc sd;
	postable*    pt;
	unspos       e = (end == 0)? seq->len : end;
	int          rc;
	char         cachePath[1200];

	if ((twMode != twOff) && (seq->v != devTargetV))
		return new_position_table (hitSeed->weight, start, e, step, true, true, false);    /* (a window of the tweener: nobody reads its table) */
	clock_mark ("table", "called");

	/* the device holds ONE table: while the main target's table is live, any other table (the
	   tweener's 7-mer tables on <=20 kbp windows when they are not batched, src/tweener.c:791) is built by the reference */
	if ((seq->len < min_target()) || (!fast_seed (hitSeed, &sd)) || (seq->fileType == seq_type_qdna)
	 || (step < 1) || (e <= start) || (e > seq->len)
	 || ((devTable != NULL) && (seq->v != devTargetV)))
		{ note ("table", "reference path");       /* (e.g. the tweener's small windows; the device keeps the main table) */
		  return ref_build_seed_position_table (seq, start, end, upperCharToBits, hitSeed, step); }

	multi_init ();
	cachePath[0] = 0;
	if ((mgWorld == 1) && (getenv ("LZGPU_TABLE_CACHE") != NULL))   /* a table file per (target, seed, step, interval) */
		{
		u64 h = 1469598103934665603ull;  unspos i;  size_t k;
		for (i=0 ; i<seq->len ; i++) { h ^= seq->v[i];  h *= 1099511628211ull; }
		for (k=0 ; k<sizeof(sd) ; k++) { h ^= ((u8*) &sd)[k];  h *= 1099511628211ull; }
		for (k=0 ; k<256 ; k++) { h ^= (u8) upperCharToBits[k];  h *= 1099511628211ull; }
		snprintf (cachePath, sizeof(cachePath), "%s/%016llx.%u.%u.%u.%u.lztab", getenv ("LZGPU_TABLE_CACHE"),
		          (unsigned long long) h, (unsigned) seq->len, (unsigned) start, (unsigned) e, (unsigned) step);
		}
	if ((mgWorld > 1) && (mgRank != 0)) rc = 0;                    /* the table comes from rank 0 */
	else if ((cachePath[0] != 0) && (access (cachePath, R_OK) == 0) && (lzgpu_table_load (cachePath) == 0))
		{ rc = 0;  note ("table", "loaded from the table cache");  cachePath[0] = 0; }
	else rc = lzgpu_table_prepare (seq->v, seq->len, start, e, upperCharToBits, &sd, step);
	if (rc < 0) suicidef ("lzgpu_table_prepare: %s", lzgpu_last_error());
	if ((rc > 0) && (mgWorld > 1)) suicidef ("lzgpu_table_prepare declined (%d) in a multi-process run", rc);
	if (rc > 0)
		{ note ("table", "declined, reference path");  drop_device_table ();
		  return ref_build_seed_position_table (seq, start, end, upperCharToBits, hitSeed, step); }
	if ((cachePath[0] != 0) && (lzgpu_table_save (cachePath) != 0))
		fprintf (stderr, "lzgpu: warning: %s\n", lzgpu_last_error());
	if ((mgWorld > 1) || (getenv ("LZGPU_SHARE_FORCE") != NULL))
		{
		char dir[1024];                                            /* one rendezvous directory per table of the run */
		if (mgTables == 0)
			{
			/* start-up self-check of a multi-process run: this rank's library is bound to the device the launcher gave it
			   (LOCAL_RANK), or the run stops here -- a rank that lands on another rank's device would still produce the
			   right bytes, slowly, on a GPU that is not its own.  The line below is what the launcher checks. */
			char* lr = getenv ("LOCAL_RANK");
			int   dev;
			rc = lzgpu_init (-1);
			if (rc != 0) suicidef ("lzgpu_init: %s", lzgpu_last_error());
			dev = lzgpu_device_index ();
			fprintf (stderr, "[lzgpu] rank %d of %d: device %d\n", mgRank, mgWorld, dev);
			if ((lr != NULL) && (dev != atoi (lr)))
				suicidef ("rank %d is bound to device %d but LOCAL_RANK=%s (fewer visible devices than ranks?)", mgRank, dev, lr);
			}
		snprintf (dir, sizeof(dir), "%s/table%d", mgDir, mgTables++);
		if ((mgRank == 0) && (mkdir (dir, 0700) != 0) && (errno != EEXIST)) suicidef ("cannot create %s: %s", dir, strerror (errno));
		rc = lzgpu_table_share (mgRank, mgWorld, dir);
		if (rc != 0) suicidef ("lzgpu_table_share: %s", lzgpu_last_error());
		note ("table", (mgRank == 0)? "shared with the other ranks" : "received from rank 0");
		}

	/* the host object in the reference's layout (callers free it, read its geometry); its arrays are filled by
	   host_table_needed() only if the capsule writer, masking, --tableonly or a reference fall-back reads them */
	pt = new_position_table (hitSeed->weight, start, e, step, true, true, false);
	devTable = pt;  devTableOnHost = false;  devTargetV = seq->v;  devTargetLen = seq->len;
	if (getenv ("LZGPU_EAGER_EXPORT") != NULL) host_table_needed (pt);
	memcpy (devCharToBits, upperCharToBits, 256);
	note ("table", "built on the GPU");
	return pt;
	}

void free_position_table (postable* pt)
	{ if (pt == devTable) drop_device_table ();  ref_free_position_table (pt); }   /* (lastz reuses seq->v for the next target: the device copy is stale from here on) */

void mask_seed_position_table
   (postable* pt, seq* seq, unspos start, unspos end, const s8 upperCharToBits[], seed* hitSeed)
	{
	/* dynamic masking (--masking=<count>, src/masking.c) has just rewritten bases of seq->v IN PLACE and now
	   removes their seeds from the table: the device's table AND its copy of the target bytes (which the gapped
	   stage reads, with or without a table) are stale from here on */
	host_table_needed (pt);
	if ((pt == devTable) || ((seq != NULL) && (seq->v == devTargetV))) drop_device_table ();
	ref_mask_seed_position_table (pt, seq, start, end, upperCharToBits, hitSeed);
	}

void limit_position_table (postable* pt, u32 limit, unspos maxChasm)
	{ host_table_needed (pt);  if (pt == devTable) drop_device_table ();  ref_limit_position_table (pt, limit, maxChasm); }

void dump_position_table (FILE* f, postable* pt, seed* hitSeed, int showPositions, int showCounts)
	{ host_table_needed (pt);  ref_dump_position_table (f, pt, hitSeed, showPositions, showCounts); }

unspos count_position_table (postable* pt)
	{ host_table_needed (pt);  return ref_count_position_table (pt); }

poscount* position_table_count_distribution (postable* pt)
	{ host_table_needed (pt);  return ref_position_table_count_distribution (pt); }

u32 find_position_table_limit (postable* pt, float keep)
	{ host_table_needed (pt);  return ref_find_position_table_limit (pt, keep); }

u64 write_capsule_file (FILE* f, char* filename, seq* seq, u8* revNucs, postable* pt, seed* seed)
	{ host_table_needed (pt);  return ref_write_capsule_file (f, filename, seq, revNucs, pt, seed); }

u32 quantum_seed_hit_search
   (seq* seq1, postable* pt, seq* seq2, unspos start, unspos end, const s8 charToBits[], seed* hitSeed,
	scoreset* scoring, score ballScore, hitprocessor processor, void* processorInfo)
	{ host_table_needed (pt);
	  return ref_quantum_seed_hit_search (seq1, pt, seq2, start, end, charToBits, hitSeed, scoring, ballScore, processor, processorInfo); }

/* ---- B2 ---- */

u64 seed_hit_search
   (seq* seq1, postable* pt, seq* seq2, unspos start, unspos end, int selfCompare,
	const s8 upperCharToBits[], seed* hitSeed, u32 searchLimit, u32 reportSearchLimit,
	u32 bandWidth, hitprocessor processor, void* processorInfo)
	{
	hitprocinfo*   hp = (hitprocinfo*) processorInfo;
	lz_search_args a;
	lz_hsp*        h = NULL;
	uint64_t       n = 0, k;
	u64            basesHit = 0;
	int            rc;

	if (twMode == twRecord)                                     /* a window of the tweener: noted, searched later with all the others */
		{
		twWindow* w;  u32 tOff, qOff;
		if (twFailed) return 0;
		if ((processor != process_for_simple_hit) || (hp->gfExtend != gfexXDrop) || (hp->hspThreshold.t != 'S') || (hp->posFilter)
		 || (hp->minMatches >= 0) || (hp->entropicHsp) || (hp->reportEntropy) || (selfCompare) || (bandWidth != 0) || (searchLimit != 0)
		 || (start != 0) || ((end != 0) && (end != seq2->len)) || (seq1->len > 20480) || (seq2->len > 20480)
		 || (!window_offsets (seq1, seq2, &tOff, &qOff)))
			{ twFailed = true;  return 0; }
		if (twNum == twCap) { twCap = 2*twCap + 64;  twWin = (twWindow*) realloc_or_die ("lzgpu windows", twWin, twCap * sizeof(twWindow)); }
		w = &twWin[twNum++];  memset (w, 0, sizeof(*w));
		w->tOff = tOff;  w->tLen = seq1->len;  w->qOff = qOff;  w->qLen = seq2->len;
		if (!twHaveHp) { twHp = *hp;  twHaveHp = true;  twCharToBits = upperCharToBits; }
		return 0;
		}
	if ((twMode == twReplaySearch) || (twMode == twReplayAll))  /* its HSPs, in the order the reference finds them */
		{
		twWindow* w = &twWin[twCursor++];
		u32 i;
		if ((twCursor > twNum) || (w->tLen != seq1->len) || (w->qLen != seq2->len)) suicide ("lzgpu: the tweener's windows changed between passes");
		for (i=0 ; i<w->hspCount ; i++)
			basesHit += (*hp->reporter) (hp->reporterInfo, twHsps[w->hspStart+i].pos1, twHsps[w->hspStart+i].pos2,
			                             twHsps[w->hspStart+i].length, twHsps[w->hspStart+i].score);
		return basesHit;
		}

	clock_mark ("search", "called");
	/* one process per GPU: a unit of another rank yields nothing here, whichever routine would have searched it
	 * (the tweener's windows are cut from a unit this rank owns: src/tweener.c:1073-1075 gives them no file) */
	if (seq2->fileType != seq_type_nofile)
		{
		if (!unit_is_mine (seq2))
			{ note ("search", "unit of another rank");  empty_diag_hash ();  return 0; }
		unit_marker (seq2);
		}

	if ((pt != devTable) || (devTable == NULL) || (seq1->v != devTargetV) || (seq1->len != devTargetLen)
	 || (processor != process_for_simple_hit) || (hp->gfExtend != gfexXDrop)
	 || (hp->hspThreshold.t != 'S') || (hp->posFilter) || (hp->minMatches >= 0) || (hp->reportEntropy)
	 || (selfCompare) || (bandWidth != 0) || (searchLimit != 0)
	 || (seq2->fileType == seq_type_qdna) || (hp->seq1 != seq1) || (hp->seq2 != seq2)
	 || (memcmp (upperCharToBits, devCharToBits, 256) != 0)
	 || (seed_search_dbgDumpRawHits) || (seed_search_dbgShowHits) || (seed_search_dbgShowCoverage))
		{ note ("search", "reference path");  host_table_needed (pt);
		  return ref_seed_hit_search (seq1, pt, seq2, start, end, selfCompare, upperCharToBits, hitSeed,
		                              searchLimit, reportSearchLimit, bandWidth, processor, processorInfo); }

	chunk_for (hitSeed, seq1->len, seq2->len);
	memset (&a, 0, sizeof(a));
	a.query = seq2->v;  a.qlen = seq2->len;  a.query_slot = -1;
	a.start = start;    a.end = end;
	a.sub   = (const int32_t*) hp->scoring->sub;
	a.xdrop = hp->xDrop;  a.hsp_threshold = hp->hspThreshold.s;  a.entropic = hp->entropicHsp;  a.extend = 1;

	rc = lzgpu_seed_hit_search (&a, &h, &n);
	if (rc < 0) suicidef ("lzgpu_seed_hit_search: %s", lzgpu_last_error());
	if (rc > 0)
		{ note ("search", "declined, reference path");  host_table_needed (pt);
		  return ref_seed_hit_search (seq1, pt, seq2, start, end, selfCompare, upperCharToBits, hitSeed,
		                              searchLimit, reportSearchLimit, bandWidth, processor, processorInfo); }

	empty_diag_hash ();                                        /* the reference's side effect, src/seed_search.c:362 */
	for (k=0 ; k<n ; k++)                                      /* HSPs arrive in discovery order */
		basesHit += (*hp->reporter) (hp->reporterInfo, h[k].pos1, h[k].pos2, h[k].length, h[k].score);
	if ((n > 0) && (hp->anchors != NULL) && (*(hp->anchors) != NULL))
		(*(hp->anchors))->haveScores = true;                   /* src/seed_search.c:2952-2953 */
	lzgpu_free (h);
	note ("search", "done on the GPU");
	return basesHit;
	}

/* ---- B3 ---- */

alignel* gapped_extend
   (seq* seq1, u8* rev1, seq* seq2, u8* rev2, int inhibitTrivial, scoreset* scoring, segtable* anchors,
	tback* tb, int allBounds, score yDrop, int trimToPeak, sthresh scoreThresh,
	u64 maxPairedBases, int overlyPairedWarn, int overlyPairedKeep)
	{
	lz_gapped_args a;
	lz_segment*    segs;
	uint32_t*      sep1, *sep2;
	lz_align*      al = NULL;
	uint32_t*      ops = NULL;
	uint64_t       n = 0, nops = 0, k;
	u32            ix, j;
	alignel*       head = NULL, *last = NULL, *el;
	int            rc;

	if (twMode == twOff) clock_mark ("gapped", "called");
	if (twMode == twReplaySearch)                               /* a window's anchors (chained by the reference): noted, extended later */
		{
		twWindow* w = &twWin[twCursor-1];
		if (twFailed) return NULL;
		if ((allBounds) || (!trimToPeak) || (scoreThresh.t != 'S') || (maxPairedBases != 0) || (tb == NULL) || (scoring->gapExtend <= 0)
		 || ((twGap.have) && ((twGap.scoring != scoring) || (twGap.yDrop != yDrop) || (twGap.scoreThresh.s != scoreThresh.s))))
			{ twFailed = true;  return NULL; }
		if ((anchors == NULL) || (anchors->len == 0)) return NULL;
		sort_segments (anchors, qSegmentsByDecreasingScore);    /* batched_segments, src/gapped_extend.c:1675 */
		w->anchors = (lz_segment*) malloc_or_die ("lzgpu window anchors", ((size_t) anchors->len) * sizeof(lz_segment));
		for (ix=0 ; ix<anchors->len ; ix++)
			{
			w->anchors[ix].pos1 = anchors->seg[ix].pos1;  w->anchors[ix].pos2   = anchors->seg[ix].pos2;
			w->anchors[ix].s    = anchors->seg[ix].s;     w->anchors[ix].length = anchors->seg[ix].length;
			w->anchors[ix].id   = anchors->seg[ix].id;
			}
		w->numAnchors = anchors->len;
		w->strandsDiffer = (seq1->revCompFlags != seq2->revCompFlags);
		twGap.scoring = scoring;  twGap.inhibitTrivial = inhibitTrivial;  twGap.yDrop = yDrop;  twGap.scoreThresh = scoreThresh;
		twGap.tbSize = tb->size;  twGap.have = true;
		return NULL;
		}
	if (twMode == twReplayAll)
		{
		twWindow* w = &twWin[twCursor-1];
		if ((anchors != NULL) && (anchors->len != 0)) sort_segments (anchors, qSegmentsByDecreasingScore);   /* (the reference's side effect) */
		return alignels_from (w->aligns, w->numAligns, w->ops, seq1, seq2);
		}

	/* gapped stage without a seed search (--segments=<file>): put the target on the device first */
	if ((devTable == NULL) && ((seq1->v != devTargetV) || (seq1->len != devTargetLen))
	 && (seq1->len >= min_target()) && (seq1->fileType != seq_type_qdna))
		{
		rc = lzgpu_target_upload (seq1->v, seq1->len);
		if (rc < 0) suicidef ("lzgpu_target_upload: %s", lzgpu_last_error());
		if (rc == 0) { devTargetV = seq1->v;  devTargetLen = seq1->len; }
		}

	if ((devTargetV == NULL) || (seq1->v != devTargetV) || (seq1->len != devTargetLen)
	 || (scoreThresh.t != 'S')
	 || ((gapped_extend_dbgAllowBatches) && (seq1->partition.p != NULL)) || (seq2->choresFile != NULL)
	 || (tb == NULL) || (anchors == NULL) || (anchors->len == 0) || (scoring->gapExtend <= 0))
		{ note ("gapped", "reference path");
		  return ref_gapped_extend (seq1, rev1, seq2, rev2, inhibitTrivial, scoring, anchors, tb, allBounds, yDrop,
		                            trimToPeak, scoreThresh, maxPairedBases, overlyPairedWarn, overlyPairedKeep); }

	sort_segments (anchors, qSegmentsByDecreasingScore);       /* batched_segments, src/gapped_extend.c:1675 */
	segs = (lz_segment*) malloc_or_die ("lzgpu gapped_extend", ((size_t) anchors->len) * sizeof(lz_segment));
	for (ix=0 ; ix<anchors->len ; ix++)
		{
		segs[ix].pos1 = anchors->seg[ix].pos1;  segs[ix].pos2   = anchors->seg[ix].pos2;
		segs[ix].s    = anchors->seg[ix].s;     segs[ix].length = anchors->seg[ix].length;
		segs[ix].id   = anchors->seg[ix].id;
		}

	memset (&a, 0, sizeof(a));
	a.query = seq2->v;  a.qlen = seq2->len;  a.query_slot = -1;
	a.sub = (const int32_t*) scoring->sub;  a.gap_open = scoring->gapOpen;  a.gap_extend = scoring->gapExtend;
	a.ydrop = yDrop;  a.score_thresh = scoreThresh.s;  a.traceback_bytes = tb->size;
	a.anchors = segs;  a.n_anchors = anchors->len;  a.reduce = 0;   /* reduce_to_points already ran, src/lastz.c:3401 */

	a.strands_differ = (seq1->revCompFlags != seq2->revCompFlags);  a.inhibit_trivial = (inhibitTrivial != 0);
	a.all_bounds = (allBounds != 0);  a.no_trim = (!trimToPeak);
	a.max_paired_bases = maxPairedBases;                       /* (over the limit: declined, the reference's routine warns and truncates) */
	sep1 = partition_separators (seq1, &a.n_sep1);  a.sep1 = sep1;
	sep2 = partition_separators (seq2, &a.n_sep2);  a.sep2 = sep2;

	rc = lzgpu_gapped_extend (&a, &al, &n, &ops, &nops);
	free (segs);  free (sep1);  free (sep2);
	if (rc < 0) suicidef ("lzgpu_gapped_extend: %s", lzgpu_last_error());
	if (rc > 0)
		{ note ("gapped", "declined, reference path");
		  return ref_gapped_extend (seq1, rev1, seq2, rev2, inhibitTrivial, scoring, anchors, tb, allBounds, yDrop,
		                            trimToPeak, scoreThresh, maxPairedBases, overlyPairedWarn, overlyPairedKeep); }

	for (k=0 ; k<n ; k++)                                      /* increasing start, src/gapped_extend.c:1475-1566 */
		{
		el = (alignel*) malloc_or_die ("lzgpu alignel", sizeof(alignel));
		el->next = NULL;  el->isTrivial = false;  el->hspId = 0;
		el->beg1 = al[k].beg1;  el->beg2 = al[k].beg2;  el->end1 = al[k].end1;  el->end2 = al[k].end2;
		el->s = al[k].s;  el->seq1 = seq1->v;  el->seq2 = seq2->v;
		el->script = edit_script_new ();
		for (j=0 ; j<al[k].script_len ; j++)
			{
			uint32_t w = ops[al[k].script_off + j];
			edit_script_add (&el->script, edit_op_operation(w), edit_op_repeat(w));
			}
		if (head == NULL) head = last = el;  else { last->next = el;  last = el; }
		}
	lzgpu_free (al);  lzgpu_free (ops);
	{ /* the reference warns when an extension runs out of traceback space (src/gapped_extend.c:3640-3661) */
	static uint64_t warned = 0;
	lz_counters cn;
	if ((lzgpu_counters_get (&cn) == 0) && (cn.truncated_extensions > warned))
		{
		fprintf (stderr, "WARNING. %llu gapped extension(s) of %s vs %s were truncated for lack of traceback space;"
		                 " use --allocate:traceback (see the lastz documentation) to give the DP more room.\n",
		                 (unsigned long long) (cn.truncated_extensions - warned), seq1->header? seq1->header : "target", seq2->header? seq2->header : "query");
		warned = cn.truncated_extensions;
		}
	}
	note ("gapped", "done on the GPU");
	return head;
	}

/* ---- N2: chaining ---- */

/* The library's routine takes the connection penalty as chain_connect_penalty's three constants (src/lastz.c:3687-3741);
 * the reference hands over a callback.  The constants are read off the callback with three probe pairs and the closed
 * form is checked against it on a fourth; a callback that is anything else goes to the reference's routine. */
static int chain_constants (chainer connect, score diagPen, score antiPen, int scale, lz_chain_args* a)
	{
	segment p, q;
	long long v;
	memset (&p, 0, sizeof(p));  memset (&q, 0, sizeof(q));
	p.pos1 = 100;  p.pos2 = 100;  p.length = 10;               /* ends at 109 / 109 */
	q.length = 10;
	q.pos1 = 107;  q.pos2 = 107;                               /* same diagonal, 3 bases overlap */
	v = (*connect) (&p, &q, scale);
	if ((scale <= 0) || (v % (3LL * scale) != 0)) return false;
	a->overlap_sub = (int32_t) (v / (3LL * scale));
	q.pos1 = 120;  q.pos2 = 110;                               /* 10 diagonals below, no bases between in seq 2 */
	v = (*connect) (&p, &q, scale);
	if (v % 10 != 0) return false;
	a->chain_diag = (int32_t) (v / 10);
	q.pos1 = 110;  q.pos2 = 117;                               /* 7 diagonals above (|d|*diag), no bases between in seq 1; then 5 more */
	v = (*connect) (&p, &q, scale);
	if (v != 7LL * a->chain_diag) return false;
	q.pos1 = 115;  q.pos2 = 122;
	v = (*connect) (&p, &q, scale) - 7LL * a->chain_diag;
	if (v % 5 != 0) return false;
	a->chain_anti = (int32_t) (v / 5);
	q.pos1 = 141;  q.pos2 = 118;                               /* the check: 23 diagonals, 8 bases */
	if ((*connect) (&p, &q, scale) != 23LL * a->chain_diag + 8LL * a->chain_anti) return false;
	q.pos1 = 105;  q.pos2 = 111;                               /* and: 6 diagonals the other way, 5 bases overlap in seq 1 */
	if ((*connect) (&p, &q, scale) != 6LL * a->chain_diag + 5LL * scale * a->overlap_sub) return false;
	a->diag_pen = diagPen;  a->anti_pen = antiPen;  a->scale = scale;
	return true;
	}

score try_reduce_to_chain
   (seq* seq1, seq* seq2, segtable* st, score diagPen, score antiPen, int scale, chainer connect)
	{
	lz_chain_args a;
	lz_segment*   segs;
	segment*      keep;
	uint32_t*     kept = NULL;
	uint32_t      nKept = 0, ix;
	int32_t       best = 0;
	int           rc;

	if ((seq1->partition.p != NULL) || (seq2->partition.p != NULL)     /* (batches per partition: src/chain.c:252-470) */
	 || (st == NULL) || (st->len == 0) || (!chain_constants (connect, diagPen, antiPen, scale, &a)))
		{ note ("chain", "reference path");
		  return ref_try_reduce_to_chain (seq1, seq2, st, diagPen, antiPen, scale, connect); }

	segs = (lz_segment*) malloc_or_die ("lzgpu try_reduce_to_chain", ((size_t) st->len) * sizeof(lz_segment));
	for (ix=0 ; ix<st->len ; ix++)
		{
		segs[ix].pos1 = st->seg[ix].pos1;  segs[ix].pos2   = st->seg[ix].pos2;
		segs[ix].s    = st->seg[ix].s;     segs[ix].length = st->seg[ix].length;
		segs[ix].id   = st->seg[ix].id;
		}
	rc = lzgpu_reduce_to_chain (&a, segs, st->len, &kept, &nKept, &best);
	free (segs);
	if (rc < 0) suicidef ("lzgpu_reduce_to_chain: %s", lzgpu_last_error());
	if (rc > 0)
		{ note ("chain", "declined, reference path");
		  return ref_try_reduce_to_chain (seq1, seq2, st, diagPen, antiPen, scale, connect); }

	keep = (segment*) malloc_or_die ("lzgpu try_reduce_to_chain", ((size_t) nKept + 1) * sizeof(segment));
	for (ix=0 ; ix<nKept ; ix++) keep[ix] = st->seg[kept[ix]];
	for (ix=0 ; ix<nKept ; ix++) { st->seg[ix] = keep[ix];  st->seg[ix].filter = false; }
	st->len = nKept;
	free (keep);  lzgpu_free (kept);
	note ("chain", "done by the library");
	return best;
	}

/* ---- the tweener ---- */

alignel* tweener_interpolate
   (alignel* alignList, seq* seq1, seq* seq2, int selfCompare, int inhibitTrivial, const s8 charToBits[], seed* tweenSeed,
	scoreset* scoring, scoreset* maskedScoring, tback* tb, score xDrop, int gappedAllBounds, score yDrop, int trimToPeak,
	score scoreThresh, score diagPen, score antiPen, int scale, chainer connect, u32 windowSize)
	{
#define refTween() ref_tweener_interpolate (alignList, seq1, seq2, selfCompare, inhibitTrivial, charToBits, tweenSeed, scoring,    \
	                                        maskedScoring, tb, xDrop, gappedAllBounds, yDrop, trimToPeak, scoreThresh, diagPen,      \
	                                        antiPen, scale, connect, windowSize)
	lz_seed_desc sd;
	lz_window*   wins;
	lz_window_search_args sa;
	uint32_t*    counts = NULL;
	uint64_t     nh = 0;
	u32          k, nprob, at;
	int          rc;
	alignel*     res;

	if ((alignList == NULL) || (getenv ("LZGPU_NO_WINDOW_BATCH") != NULL)
	 || (devTargetV == NULL) || (seq1->v != devTargetV) || (seq1->len != devTargetLen)
	 || (seq1->partition.p != NULL) || (seq2->partition.p != NULL) || (seq2->fileType == seq_type_qdna)
	 || (!fast_seed (tweenSeed, &sd)) || (sd.num_probes != 1) || (sd.weight_bits > 14) || (windowSize > 20480)
	 || (selfCompare) || (gappedAllBounds) || (!trimToPeak) || (tb == NULL) || (scoring->gapExtend <= 0))
		{ note ("tweener", "reference path");  return refTween (); }

	/* pass 1: which windows */
	tw_free ();
	twOuter = twHint = alignList;  twSeq1 = seq1;  twSeq2 = seq2;
	twMode = twRecord;
	res = refTween ();
	if ((twFailed) || (res != alignList))
		{ tw_free ();  note ("tweener", "declined, reference path");  return refTween (); }
	if (twNum == 0) { tw_free ();  note ("tweener", "no windows");  return alignList; }

	/* table + search of all of them */
	wins = (lz_window*) malloc_or_die ("lzgpu windows", ((size_t) twNum) * sizeof(lz_window));
	for (k=0 ; k<twNum ; k++)
		{ wins[k].t_off = twWin[k].tOff;  wins[k].t_len = twWin[k].tLen;  wins[k].q_off = twWin[k].qOff;  wins[k].q_len = twWin[k].qLen; }
	rc = lzgpu_query_upload (0x7FFF0001, seq2->v, seq2->len);     /* the strand's query, resident for both batches */
	if (rc < 0) suicidef ("lzgpu_query_upload: %s", lzgpu_last_error());
	memset (&sa, 0, sizeof(sa));
	sa.query = NULL;  sa.qlen = seq2->len;  sa.query_slot = 0x7FFF0001;
	sa.sub = (const int32_t*) twHp.scoring->sub;  sa.xdrop = twHp.xDrop;  sa.hsp_threshold = twHp.hspThreshold.s;
	sa.seed = &sd;  sa.char_to_bits = twCharToBits;  sa.windows = wins;  sa.n_windows = twNum;
	if (rc == 0) rc = lzgpu_window_search (&sa, &twHsps, &nh, &counts);
	free (wins);
	if (rc < 0) suicidef ("lzgpu_window_search: %s", lzgpu_last_error());
	if (rc > 0) { tw_free ();  note ("tweener", "declined, reference path");  return refTween (); }
	for (k=0,at=0 ; k<twNum ; k++) { twWin[k].hspStart = at;  twWin[k].hspCount = counts[k];  at += counts[k]; }
	lzgpu_free (counts);
	note ("tweener", "windows searched on the GPU");

	/* pass 2: the reference chains every window's HSPs; its anchors are noted */
	twMode = twReplaySearch;  twCursor = 0;
	res = refTween ();
	if ((twFailed) || (res != alignList) || (twCursor != twNum))
		{ tw_free ();  note ("tweener", "declined, reference path");  return refTween (); }

	/* the gapped stage of all windows with anchors */
	for (k=0,nprob=0 ; k<twNum ; k++) if (twWin[k].numAnchors != 0) nprob++;
	if (nprob != 0)
		{
		lz_gapped_args* ga  = (lz_gapped_args*) malloc_or_die ("lzgpu window problems", ((size_t) nprob) * sizeof(lz_gapped_args));
		lz_align**      out = (lz_align**)  malloc_or_die ("lzgpu window problems", ((size_t) nprob) * sizeof(lz_align*));
		uint64_t*       no  = (uint64_t*)   malloc_or_die ("lzgpu window problems", ((size_t) nprob) * sizeof(uint64_t));
		uint32_t**      ops = (uint32_t**)  malloc_or_die ("lzgpu window problems", ((size_t) nprob) * sizeof(uint32_t*));
		uint64_t*       nop = (uint64_t*)   malloc_or_die ("lzgpu window problems", ((size_t) nprob) * sizeof(uint64_t));
		for (k=0,at=0 ; k<twNum ; k++)
			{
			lz_gapped_args* a;
			if (twWin[k].numAnchors == 0) continue;
			a = &ga[at++];  memset (a, 0, sizeof(*a));
			a->query = NULL;  a->qlen = seq2->len;  a->query_slot = 0x7FFF0001;
			a->sub = (const int32_t*) twGap.scoring->sub;  a->gap_open = twGap.scoring->gapOpen;  a->gap_extend = twGap.scoring->gapExtend;
			a->ydrop = twGap.yDrop;  a->score_thresh = twGap.scoreThresh.s;  a->traceback_bytes = twGap.tbSize;
			a->anchors = twWin[k].anchors;  a->n_anchors = twWin[k].numAnchors;  a->reduce = 0;
			a->strands_differ = twWin[k].strandsDiffer;  a->inhibit_trivial = (twGap.inhibitTrivial != 0);
			a->t_off = twWin[k].tOff;  a->t_len = twWin[k].tLen;  a->q_off = twWin[k].qOff;  a->q_len = twWin[k].qLen;
			}
		rc = lzgpu_gapped_extend_batch (ga, nprob, out, no, ops, nop);
		if (rc < 0) suicidef ("lzgpu_gapped_extend_batch: %s", lzgpu_last_error());
		if (rc == 0)
			for (k=0,at=0 ; k<twNum ; k++)
				{
				if (twWin[k].numAnchors == 0) continue;
				twWin[k].aligns = out[at];  twWin[k].numAligns = no[at];  twWin[k].ops = ops[at];  at++;
				}
		free (ga);  free (out);  free (no);  free (ops);  free (nop);
		if (rc > 0) { tw_free ();  note ("tweener", "declined, reference path");  return refTween (); }
		}
	note ("tweener", "windows extended on the GPU");

	/* pass 3: the reference merges what was found */
	twMode = twReplayAll;  twCursor = 0;
	res = refTween ();
	tw_free ();
	note ("tweener", "done on the GPU");
	return res;
#undef refTween
	}
