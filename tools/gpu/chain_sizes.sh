# Chained launches against one launch over batch sizes:
#   gpurun -- 'bash tools/gpu/chain_sizes.sh rtty "2500 3000 4096 5000 7000"'
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
CFG=${1:-rtty}; SIZES=${2:-"2500 3000 4096 5000 7000"}
O=gpurun_out/chain_sizes; mkdir -p $O
export MIFSK_EXPERIMENT=1
for n in $SIZES; do for gk in 0,0 2,8; do
MIFSK_CHAIN=$gk timeout 300 python bench.py --no-cpu --no-h2d --no-extra --config $CFG --streams $n --steps 5 > $O/x.json 2>>$O/bench.err
python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); la=l['roofline']['launch']
print('$CFG', $n, 'chain $gk', round(l['roofline']['kernel_ms_avg'],3), 'ms', round(l['roofline']['frac'],4), l['payload_roundtrip_ok_streams'], la['kernel'], la['workgroups_per_cu'], 'per CU')"
done; done
