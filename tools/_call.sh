set -u
mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_graph_mode_gpu.py -m gpu -q -x 2>&1 | tail -6
timeout 600 python tools/shard_regime.py > gpurun_out/r02f/shard_regime.json 2> gpurun_out/r02f/shard.err; echo "shard rc=$?"; cat gpurun_out/r02f/shard_regime.json
