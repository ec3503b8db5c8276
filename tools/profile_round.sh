#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh r01
# Writes under gpurun_out/profile_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# 1. per-kernel time of the very command the bench line comes from
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- \
    python bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
# 2. HBM traffic, separate PMC passes (no other tracing domains)
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- \
    python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- \
    python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_write.log 2>&1
# 3. instruction mix / VALU utilisation
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT -o sq -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_sq.log 2>&1
# 4. the un-profiled bench line (with the CPU legs)
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
