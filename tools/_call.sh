set -u
mkdir -p gpurun_out/r02g
timeout 600 python -m pytest tests/test_graph_mode_gpu.py tests/test_dist_gpu.py tests/test_accept_flip.py -m gpu -q -x 2>&1 | tail -4
bash tools/profile_gpu.sh r02 50 > gpurun_out/r02g/profile.log 2>&1; tail -30 gpurun_out/r02g/profile.log
timeout 600 python tools/config_times.py > gpurun_out/r02g/config_times.json 2> gpurun_out/r02g/config_times.err; echo "cfg rc=$?"; cat gpurun_out/r02g/config_times.json
timeout 600 python bench.py > gpurun_out/r02g/bench_line.json 2> gpurun_out/r02g/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --workload adjoint > gpurun_out/r02g/bench_adjoint_line.json 2>> gpurun_out/r02g/bench.err; echo "bench adj rc=$?"; cat gpurun_out/r02g/bench_adjoint_line.json
