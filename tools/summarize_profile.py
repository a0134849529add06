#!/usr/bin/env python
"""Turns the rocprofv3 CSVs of tools/profile_bench.sh (gpurun_out/prof_r01/) into the committed
summaries under profiles/:  <tag>_kernel_stats.csv (verbatim --stats output), <tag>_pmc.md
(per-kernel counter averages per launch) and traffic.json (HBM bytes per launch, used by
bench.py's roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced streaming read, other access widths are
uncalibrated; both the raw and the doubled figure are recorded, the doubled one is reported."""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r01")
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01_and2"
    workload = sys.argv[3] if len(sys.argv) > 3 else "and2"
    docs = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"),
                os.path.join(out, tag + "_kernel_stats.csv"))
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
    for p in sorted(os.listdir(src)):
        f = os.path.join(src, p, p + "_counter_collection.csv")
        if not os.path.exists(f):
            continue
        with open(f) as fh:
            for r in csv.DictReader(fh):
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# rocprofv3 PMC counters, average per launch (%s)" % tag, "",
             "Command: `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-queries 0` "
             "under `rocprofv3 --pmc ...` (one pass per counter group, tools/profile_bench.sh).", ""]
    traffic = {}
    fresh = {}
    tpath = os.path.join(out, "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f)
    for k in sorted(per):
        if "rocclr" in k:
            continue
        lines += ["## `%s`" % k, "", "| counter | launches | avg per launch |", "|---|---|---|"]
        for c in sorted(per[k]):
            v = per[k][c]
            lines.append("| %s | %d | %.6g |" % (c, len(v), sum(v) / len(v)))
        lines.append("")
        m = re.search(r"and_kernel<\d+, (true|false), (true|false)>", k)
        if m and "FETCH_SIZE" in per[k]:
            mode = "pruned" if m.group(1) == "true" else "exhaustive"
            fetch = sum(per[k]["FETCH_SIZE"]) / len(per[k]["FETCH_SIZE"]) * 1024
            wr = per[k].get("WRITE_SIZE", [0])
            write = sum(wr) / max(1, len(wr)) * 1024
            key = "%s_%s_%d" % (workload, mode, docs)
            if key not in fresh:  # the dense-only and the general instantiation add up
                fresh[key] = {"fetch_bytes_raw": 0, "write_bytes_raw": 0}
            fresh[key]["fetch_bytes_raw"] += int(fetch)
            fresh[key]["write_bytes_raw"] += int(write)
    for key, v in fresh.items():
        v["hbm_bytes_per_launch"] = 2 * v["fetch_bytes_raw"] + v["write_bytes_raw"]
        v["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB*1024, summed over "
                     "the batch's two and_kernel launches, FETCH doubled per the gfx950 correction in "
                     "MI355X_MICROARCH.md; this counts the L2's fabric-side requests, Infinity Cache "
                     "hits included (the index + tables fit the 256 MB cache)")
        traffic[key] = v
    with open(os.path.join(out, tag + "_pmc.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(tpath, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print(open(os.path.join(out, tag + "_kernel_stats.csv")).read())
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
