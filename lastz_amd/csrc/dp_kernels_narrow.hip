// dp_kernels_narrow.hip -- the one-sided DP kernel on TWO waves per DP with the 16-bit sweep row (round 5).
//
// k_ydrop on four waves spends most of its vector instructions on what a row costs whatever its width -- three 256-lane scans, the
// hand-overs between the walks, the read-back of the row set-up -- for 1.6 cells per lane (876 wave-instructions per 400-cell row).  Half
// the lanes with four cells each pay that once per 256 cell slots instead of once per 128; what kept this from paying was LDS: a DP's
// 22 KiB allowed seven DPs = fourteen waves per CU, too few to hide the latency of a chain of short dependent steps.  With the C / D
// cells as 16-bit offsets (lz_dp_dev.hpp, LzDpCells16) a DP is 13.6 KiB (17.6 with mask stamps): eleven per CU.
// Same lz_dp_run, same executor, compiled with its own constants; the launcher (dp_kernels.hip) decides which DPs come here.
#ifndef LZ_DP_NARROW_LANES
#define LZ_DP_NARROW_LANES 128
#endif
#ifndef LZ_DP_NARROW_BATCH
#define LZ_DP_NARROW_BATCH 4
#endif
#define LZ_DP_LANES LZ_DP_NARROW_LANES
#define LZ_DP_BATCH LZ_DP_NARROW_BATCH
#define LZ_DP_K(name) name##_n
#define LZ_DP_ROW16_KERNEL 1
#define LZ_DP_WPE (LZ_DP_NARROW_LANES >= 128 ? 5 : 3)      // (waves per SIMD the register allocation must allow: LANES / 64 waves x 8-11 DPs per CU)
#define LZ_DP_WPE_FREE (LZ_DP_NARROW_LANES >= 128 ? 5 : 3)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include "lz_dp_dev.hpp"
#include "../../include/lzgpu.h"

#include "dp_kernels_dev.inc"

// one launch of the narrow kernel over job_ids[0..n) (declared in dp_kernels.hip)
int lzk_ydrop_narrow(bool no_trim, bool bounds, bool repl, unsigned n, size_t dyn_lds, hipStream_t st, const LzDpProblem* problems, const LzDpParams& P,
                     const LzDpJob* jobs, const u32* job_ids, const s32* tab, LzDpResult* res, u32 tab_rows)
{
    auto kern = bounds ? (no_trim ? (repl ? k_ydrop_n<true, true, true> : k_ydrop_n<true, true, false>) : (repl ? k_ydrop_n<false, true, true> : k_ydrop_n<false, true, false>))
                       : (no_trim ? (repl ? k_ydrop_n<true, false, true> : k_ydrop_n<true, false, false>) : (repl ? k_ydrop_n<false, false, true> : k_ydrop_n<false, false, false>));
    hipLaunchKernelGGL(kern, dim3(n), dim3(LZ_DP_LANES), dyn_lds, st, problems, P, jobs, job_ids, tab, res, tab_rows);
    return 0;                            // (a launch error is picked up by the caller's hipGetLastError behind the launches of the pass, like the four-wave kernel's)
}
// DPs of the narrow kernel a CU holds (the launcher's rule for one leading wave per DP against a copy of the row set-up in every wave)
unsigned lzk_ydrop_narrow_per_cu(bool bounds) { return bounds ? 8u : 11u; }
