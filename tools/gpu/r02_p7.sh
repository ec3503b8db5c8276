set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_p7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_fullsize.py -x -q --timeout 300 > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
grep -E "^FAILED|^E  |passed|failed|rc=" $O/parity.log | head
for C in 12000 same 1200; do python bench.py --config $C --no-cpu --steps 20 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', 'ms %.3f'%l['roofline']['kernel_ms_avg'], 'frac %.3f'%l['roofline']['frac'], l['payload_roundtrip_ok_streams'])"; done
