set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
export MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so
timeout -s KILL 200 python tools/counters.py --config 1200 --streams 256 > $O/ctr_1200_256.log 2>&1; cat $O/ctr_1200_256.log | grep -v amdgpu
timeout -s KILL 200 python tools/counters.py --config 1200noise > $O/ctr_1200noise.log 2>&1; cat $O/ctr_1200noise.log | grep -v amdgpu
timeout -s KILL 200 python tools/counters.py --config 1200noise --streams 256 > $O/ctr_1200noise_256.log 2>&1; cat $O/ctr_1200noise_256.log | grep -v amdgpu
unset MIFSK_LIBRARY
timeout -s KILL 300 python bench.py --config 1200noise --steps 10 > $O/bench_1200noise.json 2> $O/bench_1200noise.err; tail -c 3000 $O/bench_1200noise.json
