set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
b() { timeout 300 python bench.py --no-cpu --no-h2d --no-extra --config $1 --streams $2 --steps 5 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); la=l['roofline']['launch']; print('$1', $2, '${MIFSK_LIBRARY:-}', '${MIFSK_CHAIN:-default}', l['roofline']['kernel_ms_avg'], round(l['roofline']['frac'],4), l['payload_roundtrip_ok_streams'], la.get('chain_groups'), la.get('chain_chunks'), la['workgroups_per_cu'], la['lds_bytes_per_workgroup'])"; }
export MIFSK_EXPERIMENT=1
for n in 2500 3000 4096 5000 7000; do
for gk in 0,0 2,8 2,16; do MIFSK_CHAIN=$gk b rtty $n; done
done
