#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes (HBM traffic, MFMA pipe) of the bench command.
# usage: profiles/run_profile.sh <tag> [bench args]     -> gpurun_out/prof_<tag>/...
# The program itself follows `--` (python3 bench.py ...), never a wrapper (rocprofv3 preloads a library that initialises the GPU).
TAG=${1:-r2}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 2 --legs headline ${@:---streams 1}"
cd $GRAFT_REPO_ROOT
echo "command: python3 $ARGS" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err || true
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 $ARGS > $OUT/bench_mfma.json 2> $OUT/mfma.err || true
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt | head -60
