set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -q --timeout 600 -k "rtty or tile or t03 or t50 or t04 or B1056" > $O/par.log 2>&1; echo "rc=$?" >> $O/par.log; tail -15 $O/par.log | cut -c1-300
timeout 300 python tools/gpu/segstat.py rtty 0.0 2>&1 | tail -3
timeout 300 python tools/gpu/segstat.py rtty 0.1 2>&1 | tail -3
timeout 300 python bench.py --no-cpu --config rtty > $O/rtty.json 2>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty.json').read().strip().splitlines()[-1]); print('rtty', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l['roofline']['launch'])"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q --timeout 600 -k rtty > $O/full.log 2>&1; tail -3 $O/full.log
