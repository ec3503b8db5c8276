# a changed kernel: parity of build B first (hard limits), then A/B on configs CFGS against build A
#   A=minimodem_amd/libmifsk_base.so B=minimodem_amd/libmifsk_x.so K=workgroup CFGS="1200" bash tools/gpu/parity_then_ab.sh
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p $O
A=${A:-minimodem_amd/libmifsk_base.so}; B=${B:-minimodem_amd/libmifsk.so}
MIFSK_LIBRARY=$PWD/$B timeout -s KILL ${LIMIT:-240} python -m pytest tests/test_gpu_parity.py -x -q --timeout 40 --timeout-method=thread -k "${K:-workgroup}" -p no:cacheprovider > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; grep -v amdgpu.ids $O/parity.log | tail -8
grep -q "rc=0" $O/parity.log || exit 1
for c in ${CFGS:-1200}; do for i in 1 2 3; do for L in $A $B; do
MIFSK_LIBRARY=$PWD/$L timeout -s KILL 120 python bench.py --no-cpu --no-h2d --no-extra --config $c --steps 10 > $O/x.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$c', '$L', l['roofline']['kernel_ms_avg'], l['roofline']['kernel_ms_min'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; done; done; done
