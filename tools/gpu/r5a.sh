# round 5, call A: r04 build vs the hoisted-scalars build on configs[1] (+ under impairments),
# parity of the workgroup engine, profile-build counters, stream-count sweep
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout -s KILL 400 python tools/gpu/abn.py --config 1200 --libs base,main --rounds 5 --steps 10 --counters > $O/ab_1200.log 2>&1; tail -8 $O/ab_1200.log
timeout -s KILL 300 python tools/gpu/abn.py --config 1200noise --libs base,main --rounds 3 --steps 10 --counters > $O/ab_1200noise.log 2>&1; tail -8 $O/ab_1200noise.log
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 40 --timeout-method=thread -k "workgroup" -p no:cacheprovider > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; grep -v amdgpu.ids $O/parity.log | tail -4
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout -s KILL 200 python tools/counters.py --config 1200 > $O/ctr_1200.log 2>&1; tail -45 $O/ctr_1200.log
for n in 256 512 768; do timeout -s KILL 200 python tools/gpu/abn.py --config 1200 --libs main --rounds 3 --steps 10 --streams $n 2>&1 | grep "ms/launch"; done | tee $O/sweep.log
