cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/patch; mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_lidar_$ctr/pmc_fetch -o p -- python bench.py --pmc-inner-lidar > $OUT/pmc_lidar_$ctr.log 2>&1
  python tools/summarize_pmc.py $OUT/pmc_lidar_$ctr > $OUT/traffic_lidar_$ctr.txt 2>&1
  rm -rf $OUT/pmc_lidar_$ctr
done
grep -h -A1 "k_scan\|k_alloc3d" $OUT/traffic_lidar_*.txt | cut -c1-200
