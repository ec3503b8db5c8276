set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/eng; mkdir -p $O
for c in rtty same 12000; do
  timeout 200 python bench.py --config $c --no-cpu --engine workgroup > $O/b_$c.json 2>$O/b_$c.err
  python -c "
import json; l=json.loads(open('$O/b_$c.json').read().strip().splitlines()[-1]); print('$c', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['roofline'].get('launch'))"
done
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 300 python tools/counters.py --config rtty --engine workgroup
