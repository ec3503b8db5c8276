set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5all; mkdir -p $O
timeout -s KILL 2400 python -m pytest tests -q -m gpu --timeout 300 --timeout-method=thread -p no:cacheprovider -x > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log; grep -v amdgpu.ids $O/gpu_suite.log | tail -12
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
