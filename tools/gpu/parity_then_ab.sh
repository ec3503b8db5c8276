# the queue protocol of the workgroup engine: parity first (hard limits), then A/B against the build before it
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p $O
timeout -s KILL ${LIMIT:-240} python -m pytest tests/test_gpu_parity.py -x -q --timeout 40 --timeout-method=thread -k "workgroup" -p no:cacheprovider > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; grep -v amdgpu.ids $O/parity.log | tail -8
grep -q "rc=0" $O/parity.log || exit 1
for i in 1 2; do for L in minimodem_amd/libmifsk_base.so minimodem_amd/libmifsk.so; do
MIFSK_LIBRARY=$PWD/$L timeout -s KILL 120 python bench.py --no-cpu --no-h2d --no-extra --config 1200 --steps 10 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('1200', '$L', l['roofline']['kernel_ms_avg'], l['roofline']['kernel_ms_min'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; done; done
