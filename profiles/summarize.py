"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    for pre in ("void gn::fp0::", "gn::fp0::", "void gn::fp1::", "gn::fp1::", "void gn::fp2::", "gn::fp2::", "void gn::", "gn::"):
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
    # MFMA pipe counters (one pass): per-kernel averages per launch
    files = glob.glob(os.path.join(out, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    if agg:
        print("\n== MFMA pipe per launch (rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) ==")
        print("   issued bf16 FLOP = SQ_INSTS_MFMA x 32768 for v_mfma_f32_32x32x16_bf16; MFMA-busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs)")
        for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", [0, 1])[0])[:14]:
            print("%-72s " % k + "  ".join("%s=%.4g" % (n, v[0] / max(v[1], 1)) for n, v in sorted(c.items())))
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
