#!/bin/bash
# The driver's command (python bench.py, defaults) with its wall time: OUT=gpurun_out/bench_x.json
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/bench_default.json}
S=$(date +%s)
timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT 2> ${OUT%.json}.err
echo "rc=$? secs=$(( $(date +%s) - S ))"
wc -c $OUT
tail -3 ${OUT%.json}.err
