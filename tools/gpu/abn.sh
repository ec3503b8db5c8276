#!/bin/bash
# Same-box A/B of library builds over several workloads (tools/gpu/abn.py once per workload):
#   CONFIGS="1200 1200noise" LIBS=base,main ROUNDS=5 STEPS=10 bash tools/gpu/abn.sh
# Every build's whole output is compared with the first build's before anything is timed.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abn; mkdir -p $O
for c in ${CONFIGS:-1200 1200noise}; do
  timeout -s KILL ${LIMIT:-240} python tools/gpu/abn.py --config $c --libs ${LIBS:-base,main} \
      --rounds ${ROUNDS:-5} --steps ${STEPS:-10} ${ABN_ARGS:-} 2>&1 | grep -v "^$" | tee -a $O/$c.log | tail -${TAIL:-8}
done
