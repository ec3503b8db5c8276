# same-box A/B of two builds of the library: ab.sh <config> <libA> <libB> [rounds]
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
CFG=${1:-rtty}; A=${2:-minimodem_amd/libmifsk_base.so}; B=${3:-minimodem_amd/libmifsk.so}; R=${4:-3}
O=gpurun_out/ab; mkdir -p $O
for i in $(seq $R); do for L in $A $B; do
MIFSK_LIBRARY=$PWD/$L timeout 300 python bench.py --no-cpu --no-h2d --no-extra --config $CFG --steps 8 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$CFG', '$L', l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; done; done
