set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { python bench.py --config 1200 --no-cpu --steps 10 --streams $1 --engine $2 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1200 streams $1 $2', 'ms %.3f'%l['roofline']['kernel_ms_avg'], 'frac %.3f'%l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; }
for N in 512 1024 1536 2048 3072 4096; do run $N wave; run $N workgroup; done
