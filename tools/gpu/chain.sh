set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_slabs.py -q --timeout 600 -x 2>&1 | tail -3
bash tools/gpu/ab.sh same minimodem_amd/libmifsk_base.so minimodem_amd/libmifsk.so 3
