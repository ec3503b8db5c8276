#!/bin/bash
# One bench workload under several settings of an experiment variable (MIFSK_EXPERIMENT knobs):
#   CONFIG=same VAR=MIFSK_LAT_FMIN VALUES="8 6 5 4" bash tools/gpu/envsweep.sh
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/envsweep; mkdir -p $O
for v in ${VALUES}; do
  env MIFSK_EXPERIMENT=1 ${VAR}=$v timeout -s KILL 300 python bench.py --config ${CONFIG:-same} --no-cpu --no-h2d ${BENCH_ARGS:-} > $O/$v.json 2> $O/$v.err
  python - $O/$v.json $v <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'kernel_ms %.4f frac %.3f ms/pass %.4f' % (l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['ms_per_step']), l['payload_roundtrip_ok_streams'], l['pipeline']['output_sets_equal_to_serial_launch'])
PY
done
