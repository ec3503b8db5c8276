set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { python bench.py --config $1 --no-cpu --steps 10 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'ms %.3f'%l['roofline']['kernel_ms_avg'], 'frac %.3f'%l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; }
run same occ4; run 12000 occ4; run rtty occ4
export MIFSK_LIBRARY=$GRAFT_REPO_ROOT/minimodem_amd/libmifsk_occ3.so
run same occ3; run 12000 occ3; run rtty occ3
