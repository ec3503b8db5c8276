# the four BASELINE configs without the CPU legs, one library (MIFSK_LIBRARY) -- what to run after a change
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bench4; mkdir -p $O
for c in ${CFGS:-1200 rtty 12000 same}; do
  timeout -s KILL 200 python bench.py --config $c --no-cpu --no-h2d ${BENCH_ARGS:-} > $O/b_$c.json 2>$O/b_$c.err
  python -c "
import json; l=json.loads(open('$O/b_$c.json').read().strip().splitlines()[-1]); print('$c', round(l['ms_per_step'],4), round(l['roofline']['kernel_ms_avg'],4), round(l['roofline']['kernel_ms_min'],4), round(l['roofline']['frac'],4), l['payload_roundtrip_ok_streams'])"
done
