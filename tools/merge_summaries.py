#!/usr/bin/env python
"""Folds the summaries a profiling gpurun call left under gpurun_out/summaries/ into profiles/:

    python tools/merge_summaries.py <tag> [<tag> ...]

<tag>_kernel_stats.csv and <tag>_pmc.md are copied; the entries of <tag>_traffic.json are merged
into profiles/traffic.json in the order of the arguments — a workload's unpruned profile also
launches the pruned kernels a few times (the parity check), so name the tag whose entries should
stay LAST (tools/r3_profiles.sh: the *_exhaustive tags first, then the workloads' own)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, "gpurun_out", "summaries")
    out = os.path.join(ROOT, "profiles")
    tpath = os.path.join(out, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for tag in sys.argv[1:]:
        for name in (tag + "_kernel_stats.csv", tag + "_pmc.md"):
            shutil.copy(os.path.join(src, name), os.path.join(out, name))
        fresh = json.load(open(os.path.join(src, tag + "_traffic.json")))
        traffic.update(fresh)
        print(tag, sorted(fresh))
    with open(tpath, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
