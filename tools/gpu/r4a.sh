# round 4, call 1: what do configs[1]'s workers wait for?  (profile build's split staging
# timers, cache-resident vs HBM-streaming batch sizes, the LDS-DMA probe)
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/glds_probe tools/ubench/glds_probe.hip 2>&1 | grep -E "error"
timeout 200 /tmp/glds_probe 1 > $O/glds_probe.log 2>&1; cat $O/glds_probe.log
timeout 300 python tools/gpu/size_sweep.py > $O/size_sweep.log 2>&1; grep -v amdgpu.ids $O/size_sweep.log
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 300 python tools/counters.py --config 1200 > $O/ctr_1200.log 2>&1; grep -v "amdgpu.ids\|XCD " $O/ctr_1200.log
timeout 200 python bench.py --config 1200 --no-cpu --no-h2d --no-extra > $O/b_1200.json 2>$O/b_1200.err; tail -c 600 $O/b_1200.json
