set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -q --timeout 600 -k "rtty or tile or t03 or t50 or t04 or B1056" 2>&1 | tail -3
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 600 python tools/counters.py --config rtty > $O/ctr_rtty.log 2>&1; grep -E "kernel|w_stage|w_correlate|w_barrier|cyc_total|resident" $O/ctr_rtty.log
timeout 300 python bench.py --no-cpu --config rtty --steps 5 > $O/rtty.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty.json').read().strip().splitlines()[-1]); print('rtty', l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l['roofline']['launch']['lds_bytes_per_workgroup'], l['roofline']['launch']['workgroups_per_cu'])"
