set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p $O
for rep in 1 2 3; do for P in 2 3 4 5; do
timeout -s KILL 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-h2d --no-extra --pipeline $P > $O/p$P.json 2>$O/err
python - $O/p$P.json $P <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('P', sys.argv[2], 'ms_per_step %.4f' % l['ms_per_step'], 'serial %.4f' % l['roofline']['kernel_ms_avg'])
PY
done; done
