#!/bin/bash
# One GPU-box pass that produces everything profiles/ needs for a round: tests, the full bench line, rocprofv3
# kernel stats of the same bench command, the two PMC passes.  usage: bash tools/final_measure.sh <tag>; writes
# under gpurun_out/<tag>/.
set -u
O=gpurun_out/${1:-final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-cli > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > /dev/null 2>&1
python tools/pmc_fetch_write.py "$O/pmc_f/**/*counter_collection.csv" "$O/pmc_w/**/*counter_collection.csv" > $O/pmc_fetch_write.csv
head -12 $O/pmc_fetch_write.csv; head -12 $O/kernel_stats.csv
rm -rf $O/stats $O/pmc_f $O/pmc_w
LZGPU_OVERLAP=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-gapped > $O/bench_overlap.json 2> $O/bench_overlap.err
# the N > 1 code path of bench.py with two ranks on this one GPU (gloo stands in for RCCL, which refuses two ranks on one device): a smoke run, not a measurement
LZ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --tlen-multi 5000000 --q-units 3 --q-unit-len 2000000 > $O/bench_two_ranks_gloo_smoke.json 2> $O/bench_two_ranks.err; tail -c 700 $O/bench_two_ranks_gloo_smoke.json
