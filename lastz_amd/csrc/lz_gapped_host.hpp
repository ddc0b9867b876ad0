// lz_gapped_host.hpp -- host orchestration of B3 (reduce_to_points + gapped_extend,
// src/gapped_extend.c:463-559, 1012-1604), independent of how the one-sided DPs are executed.
//
// The reference extends anchors strictly one after another (best score first); every finished
// alignment becomes a bound for the later ones and removes the anchors lying on it.  Here the DPs
// of a WINDOW of upcoming anchors are run concurrently against a snapshot of the alignments
// committed so far, and then committed in the reference's order.  A speculative DP is accepted
// only if the reference would have run exactly the same DP: same neighbour segments at the
// anchor (msp_left_right) and no alignment committed after the snapshot touching the region the
// DP explored.  Otherwise the window is cut there and the anchor is re-run against the newer
// snapshot.  See DESIGN.md section 4.3.
#pragma once
#include <vector>
#include "lz_dp_dev.hpp"
#include "../../include/lzgpu.h"

struct LzHostSnapshot {
    std::vector<LzDpAlign> aligns;
    std::vector<LzDpSeg>   segs;
    std::vector<s32>       obi, oed;
    std::vector<u32>       obi_maxend;     // obi_maxend[o] = max end1 of aligns[obi[0..o]] (host index for msp_left_right)
};

struct LzDpExecutor {
    virtual ~LzDpExecutor() {}
    // Runs jobs[k] (slot fields are the executor's business) against snap; res[k] and ops[k]
    // (edit ops in traceback order) are filled.  Returns 0, LZGPU_NH_* or a negative error.
    virtual int run(const LzHostSnapshot& snap, std::vector<LzDpJob>& jobs,
                    std::vector<LzDpResult>& res, std::vector<std::vector<u32>>& ops) = 0;
};

// one problem's share of a multi-problem launch (HipDpExec::run_multi)
struct LzDpBatchItem {
    const LzHostSnapshot* snap; std::vector<LzDpJob>* jobs; std::vector<LzDpResult>* res; std::vector<std::vector<u32>>* ops;
    const u8* qdp; u32 qlen;               // the problem's query and target: DP class codes on the device (a window: offset pointers)
    const u8* tdp; u32 tlen;
};

struct LzGappedParams {
    const u8* t; u32 tlen;                 // host copies of the sequences (anchor reduction, rescoring)
    const u8* q; u32 qlen;
    const s32* sub;                        // [256][256] unmasked scoring
    s32 gap_open, gap_extend, ydrop, score_thresh;
    u32 window;                            // max anchors speculated per round
    const u32* sep1 = nullptr; u32 n_sep1 = 0;   // partition separators (lz_gapped_args), or none
    const u32* sep2 = nullptr; u32 n_sep2 = 0;
    bool strands_differ = false, inhibit_trivial = false;
    bool all_bounds = false;               // low-scoring alignments bound later extensions too (:1411-1429)
    u64 max_paired_bases = 0;              // :1441-1459 (0: no limit); exceeded -> LZGPU_NH_PAIRED_LIMIT
};

struct LzGappedStats { u64 anchors, anchors_extended, dp_runs, dp_cells, rounds, reruns, truncated, dp_rows; };

// The bounds and the masked cells a job will meet, as pieces of rows complete up to `horizon` (lz_dp_pieces.cpp; LzDpPiece in
// lz_dp_dev.hpp).  complete: no row beyond the horizon would add a piece (then the pieces hold for any number of rows).
struct LzDpPieces { std::vector<LzDpPiece> lb, rb, mk; bool complete = true; };
void lzh_dp_pieces(const LzHostSnapshot& S, const LzDpJob& J, u32 horizon, LzDpPieces& out);

void lzh_reduce_to_points(const u8* t, const u8* q, const s32* sub, lz_segment* segs, u32 n);

int lzh_gapped_extend(const LzGappedParams& G, LzDpExecutor& exec, lz_segment* anchors, u32 n_anchors,
                      std::vector<lz_align>& out, std::vector<u32>& out_ops, LzGappedStats& st);

// Test hook: the indexed neighbour search (msp_left_right with obi_maxend) against the reference's plain
// walk over every alignment starting at or before the anchor, on random snapshots.  Returns mismatches.
int lzh_selftest_neighbours(u32 seed, u32 n_aligns, u32 n_queries);

