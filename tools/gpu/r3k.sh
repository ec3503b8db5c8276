set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3k; mkdir -p $O
for pad in 0 14800 16384 18000 20480 27000; do
MIFSK_EXPERIMENT=1 MIFSK_LDS_PAD=$pad timeout 300 python bench.py --no-cpu --config rtty --steps 3 > $O/rtty_$pad.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty_$pad.json').read().strip().splitlines()[-1]); print('rtty lds pad $pad', l['roofline']['kernel_ms_avg'])"
done
