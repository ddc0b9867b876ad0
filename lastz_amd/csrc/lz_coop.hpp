// lz_coop.hpp -- the X-drop extension of ONE hit by a whole wave (phase B's slow path: HSP candidates).
//
// xdrop_extend_seed_hit's two loops (src/seed_search.c:2623-2632, 2684-2693) look serial -- run += score;
// best = max(best, run); stop when run < best - xDrop -- but a block of bases acts on the scan's margin
// m = run - best + xDrop (>= 0 while the scan is alive) as a map of a closed, associative family:
//     f(m) = FAIL            if m < A
//          = min(m, B) + C   otherwise
// with, for prefix sums P1..Pn of the block's scores, A = max(0, -min Pj) (A = INF if the block drops more than
// xDrop below a maximum it set itself), B = xDrop - max(0, max Pj), C = Pn.  "g after f" is
//     A = (B_f + C_f < A_g) ? INF : max(A_f, A_g - C_f),   B = min(B_f, B_g - C_f),   C = C_f + C_g.
// So 32 lanes take 16 consecutive bases each (512 bases per step and direction, coalesced loads; the two halves of
// a wave run the left and the right scan side by side), an exclusive scan gives every lane the margin and the running score entering its block, the first lane whose block
// fails walks its 16 bases with the reference's own loop to find the base the scan stops on, and a wave
// reduction finds the best score and the FIRST position attaining it (the reference updates on "run > best").
// Results are bit-identical to the serial loops for any matrix and any xDrop >= 0.
//
// The per-lane pieces are LZ_HD (device + tests/emul); the cross-lane glue exists twice: wave intrinsics in
// seed_kernels.hip, plain loops over the lanes in lz_coop_scan_host() below (test infrastructure).
#pragma once
#include "lz_common.hpp"

#define LZ_COOP_INF   (1 << 29)
#define LZ_COOP_BLK   16
#define LZ_COOP_LANES 32               // lanes per scan direction: the two halves of a wave run the left and the right scan side by side

struct LzCoopMap { s32 A, B, C; };
LZ_HD LzCoopMap lz_coop_identity(s32 X) { LzCoopMap m; m.A = 0; m.B = X; m.C = 0; return m; }   // (B = X: the margin never exceeds xDrop)
LZ_HD LzCoopMap lz_coop_compose(const LzCoopMap& f, const LzCoopMap& g)       // g after f
{
    LzCoopMap h;
    const s32 t = g.A - f.C;
    h.A = (f.A >= LZ_COOP_INF || g.A >= LZ_COOP_INF || f.B + f.C < g.A) ? LZ_COOP_INF : (f.A > t ? f.A : t);
    const s32 u = g.B - f.C;
    h.B = f.B < u ? f.B : u;
    h.C = f.C + g.C;
    return h;
}
LZ_HD bool lz_coop_fails(const LzCoopMap& f, s32 m) { return m < f.A; }
LZ_HD s32  lz_coop_apply(const LzCoopMap& f, s32 m) { return (m < f.B ? m : f.B) + f.C; }

// one lane's block: scores sc[0..nb) in consumption order -> its map, its best prefix sum and the (1-based)
// count of bases after which that maximum is first attained (0: no base)
LZ_HD void lz_coop_block(const s32 sc[LZ_COOP_BLK], u32 nb, s32 X, LzCoopMap& f, s32& mx, u32& jmx)
{
    s32 p = 0, minp = 0; bool internal = false;
    mx = -LZ_COOP_INF; jmx = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (u32 j = 0; j < LZ_COOP_BLK; j++) {
        if (j < nb) {
            p += sc[j];
            if (p > mx) { mx = p; jmx = j + 1; }
            if (p < minp) minp = p;
            if (mx - p > X) internal = true;
        }
    }
    f.A = internal ? LZ_COOP_INF : -minp;
    f.B = X - (mx > 0 ? mx : 0);
    f.C = p;
}

// the block in which the scan stops, base by base from margin m / running score run: bases consumed (the one the
// scan stops on included), and the best prefix (value, 1-based count) among them
LZ_HD void lz_coop_resolve(const s32 sc[LZ_COOP_BLK], u32 nb, s32 X, s32 m, s32 run, u32& used, bool& stopped, s32& mx, u32& jmx)
{
    s32 best = run - m + X;
    mx = -LZ_COOP_INF; jmx = 0; used = 0; stopped = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (u32 j = 0; j < LZ_COOP_BLK; j++) {
        if (j < nb && !stopped) {
            run += sc[j]; used = j + 1;
            if (run > mx) { mx = run; jmx = j + 1; }
            if (run > best) best = run;
            if (run < best - X) stopped = true;
        }
    }
}

struct LzCoopSide { u32 stop_pos; u32 best_pos; s32 best; };   // where the scan stopped, where its best was first attained, the best

#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
// Host statement of the cooperative algorithm (64 "lanes" as plain loops), one scan direction.
// RIGHT: bases pos1, pos1+1, ... < stop;  LEFT: bases pos1-1, pos1-2, ... >= stop.  score(i) = score of target base i.
template <bool RIGHT, class Score>
inline LzCoopSide lz_coop_scan_host(u32 pos1, s32 stop, s32 X, Score&& score)
{
    LzCoopSide out; out.stop_pos = pos1; out.best_pos = pos1; out.best = 0;
    s32 run0 = 0, best0 = 0; u32 s = pos1;
    bool alive = RIGHT ? ((s32)s < stop) : ((s32)s > stop);
    if (X < 0) alive = false;
    while (alive) {
        s32 sc[LZ_COOP_LANES][LZ_COOP_BLK]; u32 nb[LZ_COOP_LANES];
        LzCoopMap f[LZ_COOP_LANES]; s32 mx[LZ_COOP_LANES]; u32 jmx[LZ_COOP_LANES];
        const u32 room = RIGHT ? (u32)(stop - (s32)s) : (u32)((s32)s - stop);
        for (u32 l = 0; l < LZ_COOP_LANES; l++) {
            const u32 off = LZ_COOP_BLK * l;
            nb[l] = room > off ? (room - off < LZ_COOP_BLK ? room - off : LZ_COOP_BLK) : 0;
            for (u32 j = 0; j < LZ_COOP_BLK; j++) sc[l][j] = j < nb[l] ? score(RIGHT ? s + off + j : s - off - j - 1) : 0;
            lz_coop_block(sc[l], nb[l], X, f[l], mx[l], jmx[l]);
        }
        // exclusive scan
        const s32 m0 = run0 - best0 + X;
        LzCoopMap ex = lz_coop_identity(X);
        int fl = -1; s32 m_in = 0, run_in = 0;
        s64 best_key = -1; bool any = false;
        for (u32 l = 0; l < LZ_COOP_LANES && fl < 0; l++) {
            const s32 mi = lz_coop_apply(ex, m0), ri = run0 + ex.C;
            if (nb[l] && lz_coop_fails(f[l], mi)) { fl = (int)l; m_in = mi; run_in = ri; break; }
            if (nb[l] && jmx[l]) {                               // candidate: first position attaining this lane's maximum
                const s64 v = (s64)ri + mx[l];
                const s64 key = (v << 16) | (s64)(0xFFFF - (LZ_COOP_BLK * l + jmx[l]));
                if (!any || key > best_key) { best_key = key; any = true; }
            }
            ex = lz_coop_compose(ex, f[l]);
        }
        u32 used_total;
        if (fl >= 0) {
            u32 used; bool stopped; s32 rmx; u32 rj;
            lz_coop_resolve(sc[fl], nb[fl], X, m_in, run_in, used, stopped, rmx, rj);
            if (rj) { const s64 key = ((s64)rmx << 16) | (s64)(0xFFFF - (LZ_COOP_BLK * (u32)fl + rj)); if (!any || key > best_key) { best_key = key; any = true; } }
            used_total = LZ_COOP_BLK * (u32)fl + used;
            alive = false;
        } else {
            used_total = room < LZ_COOP_BLK * LZ_COOP_LANES ? room : LZ_COOP_BLK * LZ_COOP_LANES;
            if (room <= LZ_COOP_BLK * LZ_COOP_LANES) alive = false;      // reached the end of a sequence / the left stop
        }
        if (any) {
            const s32 v = (s32)(best_key >> 16); const u32 idx = 0xFFFFu - (u32)(best_key & 0xFFFF);
            if (v > best0) { best0 = v; out.best_pos = RIGHT ? s + idx : s - idx; }
        }
        run0 += ex.C;                                            // (only meaningful when the scan goes on)
        s = RIGHT ? s + used_total : s - used_total;
    }
    out.stop_pos = s; out.best = best0;
    return out;
}
#endif
