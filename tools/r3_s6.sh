#!/bin/bash
set -u
O=gpurun_out/${1:-s6}; mkdir -p $O/notes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1 LZGPU_NOTES_DIR=$GRAFT_REPO_ROOT/$O/notes
timeout 1500 python -m pytest tests/test_gpu_base_tests.py -m gpu -q 2>&1 | tail -60 | tee $O/pytest.txt
for f in $O/notes/notes_*.txt; do echo "$(basename $f) $(cat $f)"; done
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -s 2>&1 | tail -5
