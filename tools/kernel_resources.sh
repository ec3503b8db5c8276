#!/bin/bash
# Per-kernel register / spill / scratch figures of the shipped device code, from the
# code objects' own metadata (llvm-readelf --notes): what the judge reads.
#   tools/kernel_resources.sh [extra hipcc flags] > profiles/rNN_kernel_resources.txt
set -e
cd "$(dirname "$0")/../minimodem_amd/csrc"
TMP=$(mktemp -d)
for f in mifsk_kernels mifsk_wave mifsk_ingest mifsk_txdev; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I. "$@" \
        --cuda-device-only --no-gpu-bundle-output -c $f.hip -o $TMP/$f.o
    /opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/$f.o | python3 -c '
import sys, re, subprocess
src = sys.argv[1]
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and line.lstrip().startswith("-"):
        if cur.get("symbol"): rows.append(cur)
        cur = {}
    if k in ("agpr_count","vgpr_count","vgpr_spill_count","sgpr_count","sgpr_spill_count",
             "private_segment_fixed_size","group_segment_fixed_size","symbol","max_flat_workgroup_size"):
        cur[k] = v
if cur.get("symbol"): rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["symbol"].replace(".kd","")],
                          capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print("%-14s %-40s vgpr %3s agpr %3s vgpr_spill %3s sgpr %3s sgpr_spill %3s scratch %4s B static_lds %5s wg %4s" % (
        src, name, r["vgpr_count"], r["agpr_count"], r["vgpr_spill_count"], r["sgpr_count"],
        r["sgpr_spill_count"], r["private_segment_fixed_size"], r["group_segment_fixed_size"],
        r["max_flat_workgroup_size"]))
' $f
done
rm -rf $TMP
