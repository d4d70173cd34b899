set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 bash tools/profile_bench.sh r03 > gpurun_out/r03/profile_bench.log 2>&1
tail -5 gpurun_out/r03/profile_bench.log | cut -c1-200
timeout 300 python tools/trace_mc.py 25 > gpurun_out/r03/mc_phase_trace.txt 2>&1
( for m in base dummy_extract_first; do echo "== copy kernel, $m"; python tools/dbg_extract2.py $m 2>&1 | grep -E "call [345]|mesh post"; echo "== MRH_D2H_MEMCPY=1 (hipMemcpyAsync), $m"; MRH_D2H_MEMCPY=1 python tools/dbg_extract2.py $m 2>&1 | grep -E "call [345]|mesh post"; done ) > gpurun_out/r03/extract_second_context.txt 2>&1
cat gpurun_out/r03/extract_second_context.txt | cut -c1-250
timeout 300 python tools/host_path_breakdown.py > gpurun_out/r03/host_path_breakdown.txt 2>&1
tail -3 gpurun_out/r03/host_path_breakdown.txt
timeout 600 python tools/bench_cfg3.py 25 > gpurun_out/r03/cfg3_25_final.txt 2>&1
cat gpurun_out/r03/cfg3_25_final.txt
timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -1
