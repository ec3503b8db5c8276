set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3s; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_chain.py -q -m gpu --timeout 900 -v 2>&1 | grep -v SKIPPED | tail -40
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x --deselect tests/test_gpu_chain.py 2>&1 | tail -4
