#!/usr/bin/env python3
"""Turn what tools/profile_round.sh collected (gpurun_out/profile_<tag>_<config>/) into the
committed summaries under profiles/:  python tools/summarize_profile.py r02 1200 rtty 12000 same"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402   (kernel_source_id)

KERNELS = ("demod_kernel", "demod_wave_kernel")


def is_demod(name):
    return any(k in name for k in KERNELS)


def counters(path):
    agg = collections.defaultdict(list)
    info = {}
    dur = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if is_demod(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                info = {"kernel": r["Kernel_Name"].split("(")[0], "vgpr": r.get("VGPR_Count"),
                        "sgpr": r.get("SGPR_Count"), "workgroup": r.get("Workgroup_Size"),
                        "grid": r.get("Grid_Size"), "lds": r.get("LDS_Block_Size"),
                        "scratch": r.get("Scratch_Size")}
    return {k: sum(v) / len(v) for k, v in agg.items()}, max((len(v) for v in agg.values()), default=0), info


def kernel_durations(path):
    """average duration (ns) of the demod kernel in a --kernel-trace csv"""
    d = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if is_demod(r["Kernel_Name"]):
                d.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return sum(d) / max(1, len(d)), len(d)


def code_object_figures(kernel):
    """The kernel's register / spill / scratch figures from the code object's own metadata
    (profiles/<tag>_kernel_resources.txt, written by tools/kernel_resources.sh): rocprofv3's
    VGPR_Count / LDS_Block_Size columns are launch-packet fields (granules, static LDS only) and
    do not say what the kernel uses."""
    import glob
    import re
    want = re.sub(r"\s+", "", kernel or "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_resources.txt")), reverse=True):
        for line in open(path):
            m = re.match(r"\S+\s+(.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+vgpr_spill\s+(\d+)\s+sgpr\s+(\d+)"
                         r"\s+sgpr_spill\s+(\d+)\s+scratch\s+(\d+) B", line)
            name = re.sub(r"\s+", "", m.group(1)) if m else ""
            # (the plan names the instantiation without a trailing ", false"; chained launches
            # run the resumable twin, named in full)
            # (resource names carry every template argument: <SV, NQ, resumable, ring/auto-capable>)
            short = name.replace(",false,false>", ">").replace(",true,false>", ",true>")
            if m and want in (name, short, name.replace(",false>", ">")):
                return {"vgpr": int(m.group(2)), "agpr": int(m.group(3)), "vgpr_spill": int(m.group(4)),
                        "sgpr": int(m.group(5)), "sgpr_spill": int(m.group(6)), "scratch_bytes": int(m.group(7)),
                        "source": os.path.relpath(path, ROOT)}
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    configs = sys.argv[2:] or ["1200"]
    dst = os.path.join(ROOT, "profiles")
    ksid = bench.kernel_source_id()
    # the code objects' own register / spill / scratch figures, made NOW from the sources the
    # summaries are stamped with (kernel_source_id): a resources file left from before the last
    # kernel change would contradict the counters next to it
    import subprocess
    res = os.path.join(dst, "%s_kernel_resources.txt" % tag)
    with open(res, "w") as f:
        f.write("# kernel_source_id %s\n" % ksid)
        f.flush()
        subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh")], stdout=f, check=True)
    for c in configs:
        src = os.path.join(ROOT, "gpurun_out", "profile_%s_%s" % (tag, c))
        pre = "%s_%s" % (tag, c)
        for a, b in (("bench.json", "bench.json"), ("stats_kernel_stats.csv", "kernel_stats.csv"),
                     ("bench_under_rocprof.log", "bench_under_rocprof.log"),
                     ("statspipe_kernel_stats.csv", "kernel_stats_pipelined.csv")):
            if os.path.exists(os.path.join(src, a)) or not a.startswith("statspipe"):
                shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (pre, b)))
        b = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
        alg = b["roofline"]["algorithmic_bytes_per_launch"]
        # chained launches: one pass over the batch is groups x chunks dispatches
        # of the resumable kernel -- per-pass counters are the dispatch mean times that many
        plan = b["roofline"].get("launch") or {}
        groups, chunks = int(plan.get("chain_groups") or 0), int(plan.get("chain_chunks") or 0)
        per = groups * chunks if groups else 1
        fetch, n, info = counters(os.path.join(src, "fetch_counter_collection.csv"))
        write, _, _ = counters(os.path.join(src, "write_counter_collection.csv"))
        fkb, wkb = fetch["FETCH_SIZE"] * per, write["WRITE_SIZE"] * per
        hbm = fkb * 1024.0 * 2.0 + wkb * 1024.0
        json.dump({
            "workload": b["config"]["workload"], "kernel": info.get("kernel"),
            "kernel_source_id": ksid,
            "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
            "fetch_bytes_corrected": fkb * 2048.0, "write_bytes": wkb * 1024.0,
            "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide "
                    "coalesced reads); WRITE_SIZE uncalibrated (small); separate rocprofv3 --pmc passes, "
                    "--kernel-trace only",
            "dispatches_averaged": n, "dispatches_per_pass": per, "traffic_over_algorithmic": hbm / alg,
        }, open(os.path.join(dst, pre + "_hbm_traffic.json"), "w"), indent=1)
        sq, n, info = counters(os.path.join(src, "sq_counter_collection.csv"))
        clk, _, _ = counters(os.path.join(src, "clk_counter_collection.csv"))
        sq = {k: v * per for k, v in sq.items()}
        clk = {k: v * per for k, v in clk.items()}
        dur_ns, nd = kernel_durations(os.path.join(src, "clk_kernel_trace.csv"))
        samples = alg / 4.0
        derived = {"valu_insts_per_input_sample": sq["SQ_INSTS_VALU"] / samples,
                   "valu_wave_insts_per_launch": sq["SQ_INSTS_VALU"]}
        if "GRBM_GUI_ACTIVE" in clk and dur_ns:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            # (counter passes serialise the dispatches: a pass is `per` dispatches back to back)
            derived["shader_clock_ghz_during_kernel"] = clk["GRBM_GUI_ACTIVE"] / 8.0 / (dur_ns * per)
            derived["kernel_us_in_that_pass"] = dur_ns * per / 1e3
        json.dump({
            "kernel": info.get("kernel"), "workload": b["config"]["workload"],
            "kernel_source_id": ksid, "per_launch": dict(sq, **clk),
            "dispatches_averaged": n, "dispatches_per_pass": per,
            # what was launched: rocprofv3's packet fields (its VGPR count is in allocation granules
            # of the packet, its LDS the static part only), the code object's own metadata, and the
            # library's plan (dynamic LDS per workgroup, workgroups per CU)
            "launch": info, "code_object": code_object_figures(b["roofline"]["kernel"]),
            "plan": b["roofline"].get("launch"), "derived": derived,
        }, open(os.path.join(dst, pre + "_sq_counters.json"), "w"), indent=1)
        with open(os.path.join(src, "stats_kernel_stats.csv")) as f:
            for r in csv.DictReader(f):
                if is_demod(r["Name"]):
                    print("%s: rocprofv3 --stats: %s: %s calls, average %.1f us (bench.py events: %.1f us per pass%s)"
                          % (c, r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3, b["roofline"]["kernel_ms_avg"] * 1e3,
                             "; a pass is %d x %d dispatches, the %d groups' overlapping: average x %d = %.1f us"
                             % (groups, chunks, groups, chunks, float(r["AverageNs"]) / 1e3 * chunks) if groups else ""))
        if os.path.exists(os.path.join(src, "statspipe_kernel_stats.csv")):
            with open(os.path.join(src, "statspipe_kernel_stats.csv")) as f:
                for r in csv.DictReader(f):
                    if is_demod(r["Name"]):
                        print("%s: the default command (%s passes in flight): %s calls, average %.1f us each; %.1f us per pass"
                              % (c, (b.get("pipeline") or {}).get("passes_in_flight"), r["Calls"],
                                 float(r["AverageNs"]) / 1e3, b["ms_per_step"] * 1e3))
        print("%s: HBM bytes/launch %.3e = %.2f x algorithmic; VALU wave-insts/launch %.3e; frac %.3f"
              % (c, hbm, hbm / alg, sq["SQ_INSTS_VALU"], b["roofline"]["frac"]))


if __name__ == "__main__":
    main()
