#!/bin/bash
# memory-path counters of the seed kernels (counters only + --kernel-trace), one bench step of the 50 Mbp pair
set -u
O=gpurun_out/${1:-pmcmem}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { timeout 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/p_$2 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-gapped > /dev/null 2> $O/err_$2.txt; python tools/pmc_agg.py "$O/p_$2/**/*counter_collection.csv" | grep -v "rocprim\|__amd" | head -6 > $O/pmc_$2.txt; rm -rf $O/p_$2; cat $O/pmc_$2.txt; }
run "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" a
run "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" b
run "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" b2
run "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" c
run "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" d
run "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum TCC_STREAMING_REQ_sum" e
