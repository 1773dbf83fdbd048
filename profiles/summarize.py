"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for pre in ("void gn::", "gn::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:70]


def main(out):
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for f in stats:
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        print("%-72s %8s %12s %10s %6s" % ("kernel", "calls", "avg_us", "total_ms", "pct"))
        for r in rows[:30]:
            print("%-72s %8s %12.2f %10.3f %6s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                     float(r["TotalDurationNs"]) / 1e6, r.get("Percentage", "")))
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
        agg = defaultdict(lambda: [0.0, 0])
        for f in files:
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr:
                    k = short(r["Kernel_Name"])
                    agg[k][0] += float(r["Counter_Value"])
                    agg[k][1] += 1
        print("\n== %s per launch (raw counter, KiB units; gfx950: double FETCH_SIZE for wide streaming reads) ==" % ctr)
        for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
            print("%-72s launches %5d  avg %14.1f KiB" % (k, n, v / max(n, 1)))


if __name__ == "__main__":
    main(sys.argv[1])
