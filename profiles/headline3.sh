for rep in 1 2 3; do
  v=$(python bench.py --steps 100 --legs headline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('%.0f %.4f sa2 %.4f sa3 %.4f sa4 %.4f' % (d['value'], d['ms_per_step'], k['sa_hoisted_mlp_pool[2]'], k['sa_hoisted_mlp_pool[3]'], k['sa_hoisted_mlp_pool[4]']))")
  echo "rep=$rep: $v"
done
