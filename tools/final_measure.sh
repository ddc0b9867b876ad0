#!/bin/bash
# One GPU-box pass that produces everything profiles/ needs for a round: tests, smoke, the driver's bench command, rocprofv3
# kernel stats of the seed-stage + gapped legs of the same command (50 Mbp and 200 Mbp), the two PMC passes, the N > 1 code
# path with B3 beside B2 (timeline).   usage: bash tools/final_measure.sh <tag>; writes under gpurun_out/<tag>/.
set -u
O=gpurun_out/${1:-final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
LEGS="--no-cpu-baseline --no-cli --no-north-star --no-content"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats200 -- python bench.py --north-star --steps 2 --warmup 1 --no-cpu-baseline --no-content > $O/bench_200m_under_rocprof.json 2> $O/rocprof_stats200.err
find $O/stats200 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_200m.csv \;
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
python tools/pmc_fetch_write.py "$O/pmc_f/**/*counter_collection.csv" "$O/pmc_w/**/*counter_collection.csv" > $O/pmc_fetch_write.csv
head -12 $O/pmc_fetch_write.csv; head -14 $O/kernel_stats.csv | cut -c1-60,400-
rm -rf $O/stats $O/stats200 $O/pmc_f $O/pmc_w
# the N > 1 code path of bench.py on this one GPU: every unit searched and gapped-extended, B3 of unit k beside B2 of unit k+1
# (one rank: the timeline of the overlap; two ranks over gloo: RCCL refuses two ranks on one device -- a smoke run, not a measurement)
LZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 1 --force-multi --steps 1 --warmup 1 --tlen-multi 50000000 --q-units 2 --q-unit-len 50000000 --no-cpu-baseline > $O/bench_multi_path_one_rank_50m_units.json 2> $O/bench_multi_one.err
LZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 1 --force-multi --steps 1 --warmup 1 --tlen-multi 50000000 --q-units 2 --q-unit-len 50000000 --no-cpu-baseline --no-gapped > $O/bench_multi_path_one_rank_50m_units_nogapped.json 2>> $O/bench_multi_one.err
LZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 1 --force-multi --steps 1 --warmup 1 --tlen-multi 200000000 --q-units 1 --q-unit-len 200000000 --no-cpu-baseline > $O/bench_multi_path_one_rank_200m_unit.json 2>> $O/bench_multi_one.err
LZ_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --tlen-multi 5000000 --q-units 3 --q-unit-len 2000000 --no-cpu-baseline > $O/bench_two_ranks_gloo_smoke.json 2> $O/bench_two_ranks.err
python - <<PY
import json
for f in ("bench_multi_path_one_rank_50m_units", "bench_multi_path_one_rank_50m_units_nogapped", "bench_multi_path_one_rank_200m_unit", "bench_two_ranks_gloo_smoke"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f).read().split("\n") if l.startswith("{")][-1])
        print(f, "ms/step", round(d["ms_per_step"], 1), "overlap", d.get("overlap"), "table share", round(d["table_build_and_broadcast_share_of_step"], 4), "alignments", d.get("alignments"))
    except Exception as e:
        print(f, "failed", e)
PY
