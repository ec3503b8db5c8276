set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/verify; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests -m gpu -x -q --timeout 300 -p no:cacheprovider > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; tail -3 $O/gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > $O/bench.json 2>$O/bench.err; python -c "
import json; l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print({k:l[k] for k in ('value','ms_per_step','n_gpus')}, l['roofline']['frac'], l['roofline']['traffic'], l['cpu_baseline']['value'], l['cpu_baseline']['cores'], l['cpu_baseline']['mismatching_streams'])"
