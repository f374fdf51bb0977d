set -u
mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r02e/gtrace -o g -- python $REPO/tools/graph_step_trace.py 8192 > $REPO/gpurun_out/r02e/gtrace.log 2>&1
cd $REPO
grep "ms per step" gpurun_out/r02e/gtrace.log
python tools/trace_gaps.py $(find gpurun_out/r02e/gtrace -name "*kernel_trace.csv" | head -1) 3000 > gpurun_out/r02e/graph_step_gaps.json
cat gpurun_out/r02e/graph_step_gaps.json
rm -rf gpurun_out/r02e/gtrace
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02e/gpu_tests.txt; cat gpurun_out/r02e/gpu_tests.txt
