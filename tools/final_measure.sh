#!/bin/bash
# One GPU-box pass that produces everything profiles/ needs for a round.  usage: bash tools/final_measure.sh <tag>; writes under gpurun_out/<tag>/.
# Order matters: the driver's bench command runs FIRST, as the driver runs it -- the first process on a fresh box (k_fill_hits2 and
# k_count_hits are 30-40 % slower in the processes that follow another one on the same box: profiles/r05_fill_kernel_process_to_process.txt).
set -u
O=gpurun_out/${1:-final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
LEGS="--no-cpu-baseline --no-cli --no-north-star --no-content --no-pmc"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats200 -- python bench.py --north-star --steps 2 --warmup 1 --no-cpu-baseline --no-content --no-pmc > $O/bench_200m_under_rocprof.json 2> $O/rocprof_stats200.err
find $O/stats200 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_200m.csv \;
# the committed PMC table (what bench.py --no-pmc replays; the default run collects its own)
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
python tools/pmc_fetch_write.py "$O/pmc_f/**/*counter_collection.csv" "$O/pmc_w/**/*counter_collection.csv" > $O/pmc_fetch_write.csv
head -12 $O/pmc_fetch_write.csv; head -14 $O/kernel_stats.csv | cut -c1-60,400-
rm -rf $O/stats $O/stats200 $O/pmc_f $O/pmc_w
# instruction counts of k_ydrop (VALU / SALU / LDS wave-instructions per DP row)
bash tools/dp_pmc.sh $O/dp_pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" > $O/dp_pmc.txt 2>&1; tail -6 $O/dp_pmc.txt
# the N > 1 code path of bench.py on this one GPU.  (a) the WHOLE configs[3] job -- 200 Mbp target, 15 queries of 200 Mbp, every unit searched and
# gapped-extended, unit 0 = the north-star query with its SHA checked in the line: the N = 1 anchor of the curve; (b) configs[4]'s shape (--chain) and
# the B2-beside-B3 timeline on 50 Mbp units; (c) two ranks over gloo (RCCL refuses two ranks on one device): a smoke run, not a measurement
# (d, round 6) the same branch on its DEFAULT backend -- torch.distributed "nccl" = RCCL, world 1 -- small shape: init_process_group(nccl, device_id) and the
# zero-copy broadcast of the table's device buffers have run on this box before the driver's 8-GPU node runs them ("table_transport": "rccl" in the line)
timeout 600 python bench.py --gpus 1 --force-multi --steps 1 --warmup 0 --tlen-multi 50000000 --q-units 2 --q-unit-len 50000000 --no-cpu-baseline > $O/bench_multi_path_one_rank_rccl.json 2> $O/bench_multi_rccl.err; wc -l $O/bench_multi_path_one_rank_rccl.json; python -c "import json; d=json.load(open('$O/bench_multi_path_one_rank_rccl.json')); print('rccl one rank:', d['table_transport'], d['table_bytes'], 'ms/step', round(d['ms_per_step'],1), 'lpt', d['lpt_imbalance'], [r['units'] for r in d['per_rank']])"
LZ_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 1 --force-multi --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_multi_path_one_rank_configs3_full.json 2> $O/bench_multi_full.err; tail -c 400 $O/bench_multi_path_one_rank_configs3_full.json; echo
LZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 1 --force-multi --steps 1 --warmup 1 --tlen-multi 50000000 --q-units 2 --q-unit-len 50000000 --no-cpu-baseline > $O/bench_multi_path_one_rank_50m_units.json 2> $O/bench_multi_one.err
LZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 1 --force-multi --chain --steps 1 --warmup 1 --tlen-multi 50000000 --q-units 2 --q-unit-len 50000000 --no-cpu-baseline > $O/bench_multi_path_one_rank_50m_units_chain.json 2>> $O/bench_multi_one.err
LZ_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --tlen-multi 5000000 --q-units 3 --q-unit-len 2000000 --no-cpu-baseline > $O/bench_two_ranks_gloo_smoke.json 2> $O/bench_two_ranks.err
python - <<PY
import json
for f in ("bench_multi_path_one_rank_configs3_full", "bench_multi_path_one_rank_50m_units", "bench_multi_path_one_rank_50m_units_chain", "bench_two_ranks_gloo_smoke"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f).read().split("\n") if l.startswith("{")][-1])
        print(f, "ms/step", round(d["ms_per_step"], 1), "value", round(d["value"], 4), "overlap", d.get("overlap", {}).get("rank0_wall_over_sum"), "alignments", d.get("alignments"),
              "gcups", [r.get("gapped_gcups") for r in d.get("per_rank", [])], "north_star_unit", d.get("north_star_unit"), "selfcheck ok", d.get("device_selfcheck", {}).get("ok"))
    except Exception as e:
        print(f, "failed", e)
PY
# where the host time of the gapped stage goes (this build)
LZGPU_HOSTPROF=1 LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O/hostprof default 2>&1 | cut -c1-40,240-420 | tee $O/hostprof.txt
grep "hostprof\] gapped\|dpprof\] launch" $O/hostprof/bench_default.err | head -8 | cut -c1-300 | tee -a $O/hostprof.txt
