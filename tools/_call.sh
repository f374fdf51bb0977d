timeout 900 python -m pytest tests/test_many_segments.py tests/test_lookahead.py tests/test_graph_mode_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -8
