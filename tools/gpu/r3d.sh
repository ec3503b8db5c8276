set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slabs.py -q --timeout 600 -x > $O/slabs.log 2>&1; echo "rc=$?" >> $O/slabs.log; tail -30 $O/slabs.log | cut -c1-400
timeout 300 python bench.py --no-cpu --no-h2d --no-extra > $O/bench.json 2>$O/bench.err; python -c "
import json; l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'])"
timeout 200 python bench.py --no-cpu --config 12000 > $O/b12.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/b12.json').read().strip().splitlines()[-1]); print('12000', l['ms_per_step'], l['roofline']['kernel_ms_avg'])"
