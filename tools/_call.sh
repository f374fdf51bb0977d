mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_graph_mode_gpu.py tests/test_advice_r1.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python tools/adjoint_graph_bench.py > gpurun_out/r02h/adjoint_graph_bench.json 2> gpurun_out/r02h/agb.err; echo "rc=$?"; cat gpurun_out/r02h/adjoint_graph_bench.json; tail -5 gpurun_out/r02h/agb.err | cut -c1-300
