set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python tools/trace_mc.py 25 > gpurun_out/r03/trace_mc_25.txt 2>&1
cat gpurun_out/r03/trace_mc_25.txt
