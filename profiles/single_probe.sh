for f in 1 0 1 0; do
  v=$(GN_FORWARD_FORK=$f python bench.py --steps 40 --legs single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['single_stream_ms_per_step'], d['value'])")
  echo "fork=$f single: $v"
done
