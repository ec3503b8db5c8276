set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_t7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q --timeout 300 -k "plan or bench_line" > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
grep -E "^FAILED|^E  |passed|failed|rc=" $O/gpu.log | head
python bench.py --no-cpu --steps 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['roofline']['launch'])"
python bench.py --no-cpu --steps 5 --config same 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['roofline']['launch'])"
