set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
b() { timeout 300 python bench.py --no-cpu --no-h2d --no-extra --config $1 --steps 6 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); la=l['roofline']['launch']; print('$1', '${MIFSK_SV:-}', '${MIFSK_WAVES_PER_CU:-}', l['roofline']['kernel_ms_avg'], round(l['roofline']['frac'],4), l['payload_roundtrip_ok_streams'], la.get('chain_groups'), la.get('chain_chunks'), la['workgroups_per_cu'], la['lds_bytes_per_workgroup'], la['kernel'])"; }
export MIFSK_EXPERIMENT=1
b same
MIFSK_SV=10 MIFSK_WAVES_PER_CU=8 b same
MIFSK_SV=10 b same
MIFSK_WAVES_PER_CU=12 b same
b 12000
MIFSK_SV=10 MIFSK_WAVES_PER_CU=8 b 12000
