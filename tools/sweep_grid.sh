cd "$GRAFT_REPO_ROOT"
for g in 1024 1536 2048 2560 3072 4096 6144; do
  echo -n "MRH_FUSED_GRID=$g : "; MRH_FUSED_GRID=$g python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'fps  k_ms', round(d['roofline']['kernel_ms_avg']*1000,2), 'us frac', round(d['roofline']['frac'],4))"
done
