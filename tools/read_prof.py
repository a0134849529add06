#!/usr/bin/env python
"""Summarise rocprofv3 rocpd sqlite outputs: kernel durations (kernel-trace) and PMC counters
per kernel.  usage: read_prof.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for f in sys.argv[1:]:
        c = sqlite3.connect(f)
        print("==", f)
        try:
            rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration), "
                             "sum(duration) from kernels group by name order by sum(duration) desc").fetchall()
            for r in rows:
                print("kernel %-70s calls=%d avg=%.1fus min=%.1fus max=%.1fus total=%.1fus" %
                      (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
        except Exception as e:  # noqa
            print("no kernels view:", e)
        try:
            cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
            rows = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from "
                             "counters_collection group by kernel_name, counter_name").fetchall()
            for r in rows:
                print("pmc %-40s %-28s n=%d avg=%.4g sum=%.4g" % (r[0][:40], r[1], r[2], r[3], r[4]))
        except Exception as e:  # noqa
            print("no counters:", e, cols if 'cols' in dir() else '')


if __name__ == "__main__":
    main()
