# parity of the demod path + the four bench configs without the CPU legs
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/quick; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py -x -q --timeout 300 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; tail -3 $O/gpu.log
for c in 1200 rtty 12000 same; do
  timeout 200 python bench.py --config $c --no-cpu ${BENCH_ARGS:-} > $O/b_$c.json 2>$O/b_$c.err
  python -c "
import json; l=json.loads(open('$O/b_$c.json').read().strip().splitlines()[-1]); print('$c', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'])"
done
