# compile and run micro-benchmarks from tools/ubench/:  UB="ta_rate longwin" UB_ARGS="4096 0" bash tools/gpu/ub.sh
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ub
for b in ${UB:-ta_rate}; do
  hipcc --offload-arch=gfx950 -O3 -o /tmp/$b tools/ubench/$b.hip 2>&1 | grep -E "error" 
  timeout 300 /tmp/$b ${UB_ARGS:-} 2>&1 | tee gpurun_out/ub/$b${UB_TAG:-}.log
done
