mkdir -p gpurun_out/r02i
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r02i/gpu_tests.txt; cat gpurun_out/r02i/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r02i/bench_line.json 2> gpurun_out/r02i/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r02i/bench.err
TDEQ_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 > gpurun_out/r02i/bench_n2_gloo.json 2>> gpurun_out/r02i/bench.err; echo "bench2 rc=$?"
