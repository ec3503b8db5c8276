# one micro-benchmark that includes the library's device headers:  UB=score_lat bash tools/gpu/ub2.sh
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ub
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iminimodem_amd/csrc -Iinclude -o /tmp/$UB tools/ubench/$UB.hip 2>&1 | grep -E "error"
timeout 300 /tmp/$UB ${UB_ARGS:-} > gpurun_out/ub/$UB${UB_TAG:-}.log 2>&1; cat gpurun_out/ub/$UB${UB_TAG:-}.log
