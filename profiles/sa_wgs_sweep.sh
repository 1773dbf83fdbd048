# SA workgroups per CU (tiles per workgroup): fewer, longer workgroups load the resident weights less often
for rep in 1 2; do for w in 4 2 1 8; do
  v=$(GN_SA_WGS_PER_CU=$w python bench.py --steps 100 --legs headline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('%.0f %.4f sa1 %.4f sa2 %.4f sa3 %.4f sa4 %.4f' % (d['value'], d['ms_per_step'], k['sa_group_mlp_pool[1]'], k['sa_hoisted_mlp_pool[2]'], k['sa_hoisted_mlp_pool[3]'], k['sa_hoisted_mlp_pool[4]']))")
  echo "rep=$rep wgs_per_cu=$w: $v"
done; done
