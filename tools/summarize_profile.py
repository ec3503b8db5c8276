#!/usr/bin/env python3
"""Turn what tools/profile_round.sh collected (gpurun_out/profile_<tag>/) into the
committed summaries under profiles/:  python tools/summarize_profile.py r01"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "demod_kernel<true>"


def counters(path):
    agg = collections.defaultdict(list)
    info = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                info = {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                        "workgroup": r.get("Workgroup_Size"), "grid": r.get("Grid_Size"),
                        "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size")}
    return {k: sum(v) / len(v) for k, v in agg.items()}, max((len(v) for v in agg.values()), default=0), info


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
    dst = os.path.join(ROOT, "profiles")
    for a, b in (("bench.json", "bench.json"), ("stats_kernel_stats.csv", "kernel_stats.csv"),
                 ("stats_domain_stats.csv", "domain_stats.csv"),
                 ("bench_under_rocprof.log", "bench_under_rocprof.log")):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
    bench = json.load(open(os.path.join(src, "bench.json")))
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    fetch, n, _ = counters(os.path.join(src, "fetch_counter_collection.csv"))
    write, _, _ = counters(os.path.join(src, "write_counter_collection.csv"))
    fkb, wkb = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
    hbm = fkb * 1024.0 * 2.0 + wkb * 1024.0
    json.dump({
        "workload": "bench.py default (1024 streams x 480000 samples, Bell202 1200 baud)",
        "kernel": "mifsk::" + KERNEL,
        "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
        "fetch_bytes_corrected": fkb * 2048.0, "write_bytes": wkb * 1024.0,
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide "
                "coalesced reads); WRITE_SIZE uncalibrated (small); separate rocprofv3 --pmc passes, "
                "--kernel-trace only",
        "dispatches_averaged": n, "traffic_over_algorithmic": hbm / alg,
    }, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
    sq, n, info = counters(os.path.join(src, "sq_counter_collection.csv"))
    samples = bench["config"]["streams_per_gpu"] * bench["config"]["samples_per_stream"]
    json.dump({
        "kernel": "mifsk::" + KERNEL, "workload": "bench.py default", "per_launch": sq,
        "dispatches_averaged": n, "launch": info,
        "derived": {"valu_insts_per_input_sample": sq["SQ_INSTS_VALU"] / samples,
                    "valu_wave_insts_per_launch": sq["SQ_INSTS_VALU"]},
    }, open(os.path.join(dst, tag + "_sq_counters.json"), "w"), indent=1)
    with open(os.path.join(src, "stats_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Name"]:
                print("rocprofv3 --stats: %s calls, average %.1f us (bench.py events: %.1f us)"
                      % (r["Calls"], float(r["AverageNs"]) / 1e3, bench["roofline"]["kernel_ms_avg"] * 1e3))
    print("HBM bytes/launch %.3e = %.2f x algorithmic; VALU wave-insts/launch %.3e"
          % (hbm, hbm / alg, sq["SQ_INSTS_VALU"]))


if __name__ == "__main__":
    main()
