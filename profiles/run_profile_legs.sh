#!/bin/bash
# usage (on the GPU box): bash profiles/run_profile_legs.sh <tag> <bench.py args...>  -> gpurun_out/<tag>_kernel_stats.csv + <tag>_summary.txt
# rocprofv3 --kernel-trace --stats of one bench.py command (the program itself follows `--`).
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py "$@" > gpurun_out/${TAG}_bench.json 2> $OUT/err.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
python3 - <<PY > gpurun_out/${TAG}_summary.txt
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
print("rocprofv3 --kernel-trace --stats -- python3 bench.py $*")
print("%-78s %8s %12s %10s" % ("kernel", "calls", "avg us", "total %"))
for r in rows[:45]:
    print("%-78s %8s %12.2f %10s" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat gpurun_out/${TAG}_summary.txt
