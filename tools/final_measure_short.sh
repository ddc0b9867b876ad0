#!/bin/bash
# The short form of tools/final_measure.sh for a change that touches the gapped stage only: the driver's bench command, the GPU test suite, the smoke
# call, rocprofv3 kernel stats of the 50 Mbp bench and the DP kernels' instruction counts.  usage: bash tools/final_measure_short.sh <tag>
set -u
O=gpurun_out/${1:-final_short}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
LEGS="--no-cpu-baseline --no-cli --no-north-star --no-content --no-pmc"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats
bash tools/dp_pmc.sh $O/dp_pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" > $O/dp_pmc.txt 2>&1; tail -6 $O/dp_pmc.txt
