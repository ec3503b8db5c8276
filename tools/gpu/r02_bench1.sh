set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_bench1; mkdir -p $O
for C in 1200 rtty 12000 same; do
  timeout 400 python bench.py --config $C > $O/bench_$C.json 2> $O/bench_$C.err; echo "rc=$?"; tail -c 2500 $O/bench_$C.json | cut -c1-2500; tail -3 $O/bench_$C.err
done
