set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3f; mkdir -p $O
timeout 300 python bench.py --no-cpu --config rtty > $O/rtty.json 2>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty.json').read().strip().splitlines()[-1]); print('rtty', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l['roofline']['launch'])"
for w in 4 5 6; do MIFSK_EXPERIMENT=1 MIFSK_WAVES_PER_CU=$w timeout 300 python bench.py --no-cpu --config rtty --steps 3 > $O/rtty_w$w.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty_w$w.json').read().strip().splitlines()[-1]); print('rtty wpc $w', l['roofline']['kernel_ms_avg'], l['roofline']['launch']['lds_bytes_per_workgroup'], l['roofline']['launch']['workgroups_per_cu'])"; done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q --timeout 600 -k rtty > $O/full.log 2>&1; tail -3 $O/full.log
