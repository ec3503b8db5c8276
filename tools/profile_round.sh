#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh ${TAG:-r04} "1200 rtty 12000 same"
# Writes under gpurun_out/profile_<tag>_<config>/ ; tools/summarize_profile.py copies the
# summaries into profiles/.  PMC counters are collected in their own passes with
# --kernel-trace only (never together with other tracing domains).
set -u
TAG=${1:-r04}
CONFIGS=${2:-"1200 rtty 12000 same"}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for C in $CONFIGS; do
  OUT=gpurun_out/profile_${TAG}_$C
  mkdir -p $OUT
  B="python bench.py --config $C --no-cpu --no-h2d"
  # 1. per-kernel time: the kernel launched serially on one stream -- what roofline.kernel_ms_avg
  #    is (bench.py takes it from K serial launches right after the timed passes) ...
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- \
      $B --steps 20 --warmup 3 --pipeline 1 > $OUT/bench_under_rocprof.log 2>&1
  #    ... and the very command the bench line comes from (three or four passes in flight: the timed
  #    launches overlap, so each one's own duration is longer while the passes per second go up)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o statspipe -- \
      $B --steps 20 --warmup 3 > $OUT/bench_under_rocprof_pipelined.log 2>&1
  # 2. HBM traffic, separate PMC passes
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- \
      $B --steps 4 --warmup 1 --preheat-ms 0 --pipeline 1 > $OUT/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- \
      $B --steps 4 --warmup 1 --preheat-ms 0 --pipeline 1 > $OUT/pmc_write.log 2>&1
  # 3. instruction mix / occupancy / clock
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
      --output-format csv -d $OUT -o sq -- $B --steps 4 --warmup 1 --preheat-ms 0 --pipeline 1 > $OUT/pmc_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --output-format csv -d $OUT -o clk -- $B --steps 4 --warmup 1 --preheat-ms 0 --pipeline 1 > $OUT/pmc_clk.log 2>&1
  # 4. the un-profiled bench line (with the CPU legs)
  timeout 900 python bench.py --config $C > $OUT/bench.json 2> $OUT/bench.err
  tail -c 600 $OUT/bench.json; echo
done
