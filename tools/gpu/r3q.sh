set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_slabs.py -q --timeout 900 2>&1 | tail -5
b() { timeout 300 python bench.py --no-cpu --no-h2d --no-extra --config $1 --steps 6 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); la=l['roofline']['launch']; print('$1', '${MIFSK_CHAIN:-default}', l['roofline']['kernel_ms_avg'], round(l['roofline']['frac'],4), l['payload_roundtrip_ok_streams'], la.get('chain_groups'), la.get('chain_chunks'), la['workgroups_per_cu'])"; }
export MIFSK_EXPERIMENT=1
for c in rtty same; do
  unset MIFSK_CHAIN; b $c
  for gk in 0,0 2,8 3,8 3,6 2,6 2,4 3,4; do MIFSK_CHAIN=$gk b $c; done
done
