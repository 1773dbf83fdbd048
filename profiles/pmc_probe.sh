#!/bin/bash
# usage (on the GPU box): profiles/pmc_probe.sh <tag> "<counters>"   -> per-kernel averages
TAG=${1:-probe}; CTRS=${2:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void gn::", "")[:60]
    a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void gn::", "")[:60]
        dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])); dur[k][1] += 1
for k, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]][0]):
    d = dur[k][0] / max(dur[k][1], 1)
    print("%-62s avg %9.1f us  " % (k, d / 1e3) + "  ".join("%s=%.4g" % (n, v[0] / v[1]) for n, v in sorted(c.items())))
PY
