# headline leg over stream counts, three interleaved repetitions (run-to-run noise is ~1-2 %)
for rep in 1 2 3; do for s in 10 12 14; do
  v=$(python bench.py --steps 100 --legs headline --streams $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f %.4f' % (d['value'], d['ms_per_step']))")
  echo "rep=$rep streams=$s: $v"
done; done
v=$(GN_FORWARD_FORK=1 python bench.py --steps 40 --legs single,graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['single_stream_ms_per_step'], d['hipgraph_latency_ms'])")
echo "single/graph: $v"
