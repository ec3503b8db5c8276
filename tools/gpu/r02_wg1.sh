set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_wg1; mkdir -p $O
MIFSK_ENGINE=workgroup timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 60 -k "not auto" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
tail -3 $O/parity.log
MIFSK_ENGINE=workgroup timeout 200 python tools/bench_configs.py > $O/bench_configs.log 2>&1; cat $O/bench_configs.log
