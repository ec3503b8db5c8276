set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_wave3; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
tail -15 $O/gpu.log
timeout 200 python tools/bench_configs.py > $O/bench_configs.log 2>&1; cat $O/bench_configs.log
MIFSK_ENGINE=workgroup timeout 200 python tools/bench_configs.py > $O/bench_configs_wg.log 2>&1; cat $O/bench_configs_wg.log
