set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_wave5; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 60 > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
tail -3 $O/parity.log
timeout 200 python tools/bench_configs.py > $O/bench_configs.log 2>&1; cat $O/bench_configs.log
MIFSK_LIBRARY=$GRAFT_REPO_ROOT/minimodem_amd/libmifsk_prof.so timeout 120 python tools/counters.py > $O/counters.log 2>&1; grep -E "mean|%" $O/counters.log | grep -v " 0.0  min" 
