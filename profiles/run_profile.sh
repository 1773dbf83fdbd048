#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic.
# usage: profiles/run_profile.sh <tag>     -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA}"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err || true
find $OUT -name "*.csv" | head -20
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
