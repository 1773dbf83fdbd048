#!/bin/bash
# kernel-trace stats of the steps around the network (bench.py --legs post: voxel down-sample + collision detect + sort/NMS on
# the batch's 8 decoded scenes).  Run ON THE GPU BOX: gpurun -- bash profiles/run_profile_post.sh r3_post
TAG=${1:-r3_post}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --legs post"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
grep -i "voxel\|collision\|nms\|sort" $OUT/summary.txt
