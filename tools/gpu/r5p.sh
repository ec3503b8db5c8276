set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p $O
for c in same rtty; do for ch in default 3,8 default 3,8; do
if [ $ch = default ]; then E=""; else E="MIFSK_EXPERIMENT=1 MIFSK_CHAIN=$ch"; fi
env $E timeout -s KILL 200 python bench.py --config $c --no-cpu --no-h2d --pipeline 1 > $O/$c.json 2>$O/err
python - $O/$c.json $c $ch <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
la=l['roofline']['launch']
print(sys.argv[2], 'chain', sys.argv[3], 'plan %sx%s' % (la.get('chain_groups'), la.get('chain_chunks')), 'ms_per_step %.4f' % l['ms_per_step'], l['payload_roundtrip_ok_streams'])
PY
done; done
