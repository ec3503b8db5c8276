set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_wave2; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_parity.py -x -v --timeout 40 -k "goldens" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
grep -E "PASS|FAIL|rc=|Error|assert" $O/parity.log | head -60
