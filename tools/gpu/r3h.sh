set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h; mkdir -p $O
for d in 0 1; do
MIFSK_EXPERIMENT=1 MIFSK_SEG_DIRECT=$d MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 600 python tools/counters.py --config rtty > $O/ctr_rtty_$d.log 2>&1; grep -E "kernel|w_stage|w_correlate|w_barrier|cyc_total|resident" $O/ctr_rtty_$d.log
MIFSK_EXPERIMENT=1 MIFSK_SEG_DIRECT=$d timeout 300 python bench.py --no-cpu --config rtty --steps 5 > $O/rtty_$d.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty_$d.json').read().strip().splitlines()[-1]); print('rtty direct=$d', l['roofline']['kernel_ms_avg'], l['payload_roundtrip_ok_streams'])"
done
MIFSK_EXPERIMENT=1 MIFSK_SEG_DIRECT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "rtty or tile or t03 or t50 or t04 or B1056" 2>&1 | tail -2
