#!/usr/bin/env python
"""Every bench workload at three vocabularies: the 256-term index of SURVEY.md
section 8d, 4096 and 65536 terms (same 10M docs, df_r = 0.5 N / r, ranks ~ Zipf(1)): QPS, step / kernel
/ host ms, algorithmic fraction, which kernel families ran, per-posting cost relative to the 256-term
run.  Writes gpurun_out/summaries/<tag>.json (tools/merge_summaries.py or a plain copy brings it to
profiles/).   python tools/vocab_sweep.py [tag] [workloads] [terms]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_vocab_sweep"
workloads = (sys.argv[2] if len(sys.argv) > 2 else "and2,or5,mixed,phrase3,bool").split(",")
terms = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "256,4096,65536").split(",")]
out = {"note": "bench.py --workload W --terms T --no-side --no-cpu-baseline --steps 10 (one 10M-doc segment; pruned "
               "mode timed, pruned == exhaustive and 64 oracle-checked queries per cell); ns_per_posting = kernel "
               "time / postings of the batch's lists (algorithmic bytes ~ 1.3 B per posting)", "cells": {}}
for w in workloads:
    for t in terms:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--terms", str(t), "--no-side",
               "--no-cpu-baseline", "--latency-queries", "0", "--steps", "10", "--warmup", "2"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        try:
            j = json.loads(line)
        except ValueError:
            out["cells"]["%s_t%d" % (w, t)] = {"error": (r.stderr or r.stdout)[-400:]}
            print(w, t, "FAILED", (r.stderr or r.stdout)[-300:])
            continue
        rf = j["roofline"]
        cell = {"qps": j["value"], "ms_per_step": j["ms_per_step"], "kernel_ms": rf["kernel_ms_avg"],
                "host_plan_ms": rf["host_plan_ms"], "frac": rf["frac"], "unique_frac": rf.get("unique_frac"),
                "kernels": rf["kernel"], "algorithmic_bytes": rf["algorithmic_bytes_per_launch"],
                "batch_unique_bytes": rf.get("batch_unique_bytes"), "distinct_queries": j["config"]["distinct_queries"],
                "exhaustive_kernel_ms": j["roofline_other_mode"]["kernel_ms_avg"],
                "pruned_equals_exhaustive": j["pruned_equals_exhaustive"], "parity_checked_queries": j["parity_checked_queries"],
                "index_bytes": j["config"]["index_bytes"]}
        cell["ns_per_algorithmic_kb"] = round(cell["kernel_ms"] * 1e6 / (cell["algorithmic_bytes"] / 1024.0), 4)
        out["cells"]["%s_t%d" % (w, t)] = cell
        print("%-8s T=%-6d qps %10.0f step %7.3f kernel %7.3f host %6.3f frac %6.3f ns/KB %7.4f %s" %
              (w, t, cell["qps"], cell["ms_per_step"], cell["kernel_ms"], cell["host_plan_ms"], cell["frac"],
               cell["ns_per_algorithmic_kb"], cell["kernels"]))
        sys.stdout.flush()
for w in workloads:  # cliffs: per-byte cost against the 256-term run
    base = out["cells"].get("%s_t256" % w, {}).get("ns_per_algorithmic_kb")
    for t in terms:
        c = out["cells"].get("%s_t%d" % (w, t), {})
        if base and "ns_per_algorithmic_kb" in c:
            c["cost_vs_t256"] = round(c["ns_per_algorithmic_kb"] / base, 3)
d = os.path.join(ROOT, "gpurun_out", "summaries")
os.makedirs(d, exist_ok=True)
with open(os.path.join(d, tag + ".json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote", os.path.join(d, tag + ".json"))
