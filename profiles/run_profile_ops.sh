#!/bin/bash
# PMC passes (HBM bytes) of the drop-in operator leg of the bench (group_points / ball_query / cylinder_query at SA sizes).
TAG=${1:-r2_ops}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --repeats 1 --legs ops,knn"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err || true
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
grep -i "group_points\|ball_grid\|grid_build\|query_kernel\|knn" $OUT/summary.txt
# the same at config[3]'s shapes (N = 50 000)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_n50000
mkdir -p $OUT
ARGS="profiles/ops_n50000.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err || true
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
grep -i "group_points\|ball_grid\|grid_build\|fps_cell" $OUT/summary.txt
