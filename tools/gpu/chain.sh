set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fullsize.py tests/test_gpu_slabs.py -q --timeout 900 -x 2>&1 | tail -3
for c in same rtty; do bash tools/gpu/ab.sh $c minimodem_amd/libmifsk_base.so minimodem_amd/libmifsk.so 2; done
