set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { python bench.py --config $1 --no-cpu --steps 10 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'ms %.3f'%l['roofline']['kernel_ms_avg'], 'frac %.3f'%l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; }
run same default
MIFSK_WAVES_PER_CU=8 run same wpc8
MIFSK_WAVES_PER_CU=12 run same wpc12
MIFSK_SV=10 run same sv10
run rtty default
MIFSK_WAVES_PER_CU=8 run rtty wpc8
MIFSK_WAVES_PER_CU=4 run rtty wpc4
MIFSK_WAVES_PER_CU=3 run rtty wpc3
run 12000 default
MIFSK_WAVES_PER_CU=12 run 12000 wpc12
MIFSK_WAVES_PER_CU=8 run 12000 wpc8
