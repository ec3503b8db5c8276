set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
( time timeout 600 python bench.py --no-cpu --no-h2d --no-extra --config same --steps 5 > $O/same.json 2>$O/same.err ) 2>&1 | grep real
python -c "
import json; l=json.loads(open('$O/same.json').read().strip().splitlines()[-1]); print(l['roofline']['kernel_ms_avg'], l['payload_roundtrip_ok_streams']); print(json.dumps(l['payload_by_condition'], indent=1))"
