set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { python bench.py --config $1 --no-cpu --steps 6 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'ms %.3f'%l['roofline']['kernel_ms_avg'], 'frac %.3f'%l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; }
MIFSK_LDS_PAD=13000 run rtty 12perCU
MIFSK_LDS_PAD=20000 run rtty 8perCU
MIFSK_LDS_PAD=26000 run rtty 6perCU
MIFSK_LDS_PAD=40000 run rtty 4perCU
