#!/bin/bash
# usage (on the GPU box): bash profiles/run_profile_post.sh <tag>   -> gpurun_out/<tag>_post_kernel_stats.csv + summary
TAG=${1:-r4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_post_prof
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o post -- python3 $GRAFT_REPO_ROOT/profiles/post_probe.py > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_post_kernel_stats.csv
python3 - <<PY > gpurun_out/${TAG}_post_summary.txt
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_post_kernel_stats.csv")))
print("rocprofv3 --kernel-trace --stats -- python3 profiles/post_probe.py   (50 x postprocess.grasp_pipeline on one batch of 8 decoded scenes + one forward)")
print("%-70s %8s %12s %10s" % ("kernel", "calls", "avg us", "total %"))
for r in rows:
    print("%-70s %8s %12.2f %10s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat gpurun_out/${TAG}_post_summary.txt
