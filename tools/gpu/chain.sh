set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
b() { timeout 300 python bench.py --no-cpu --no-h2d --no-extra --config $1 --steps 8 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); la=l['roofline']['launch']; print('$1', '${MIFSK_CHAIN:-default}', l['roofline']['kernel_ms_avg'], round(l['roofline']['frac'],4), l['payload_roundtrip_ok_streams'], la.get('chain_groups'), la.get('chain_chunks'), la['workgroups_per_cu'], la['kernel'])"; }
export MIFSK_EXPERIMENT=1
for r in 1 2; do for gk in 0,0 2,3 2,2 3,3 2,4; do MIFSK_CHAIN=$gk b 12000; done; done
unset MIFSK_CHAIN; b 12000
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fullsize.py -q --timeout 600 2>&1 | tail -3
