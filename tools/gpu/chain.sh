set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_slabs.py tests/test_gpu_fullsize.py -q --timeout 900 -x 2>&1 | tail -3
bash tools/gpu/ab.sh same minimodem_amd/libmifsk_base.so minimodem_amd/libmifsk.so 3
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 600 python tools/counters.py --config same 2>&1 | grep -E "mean " | head -30
