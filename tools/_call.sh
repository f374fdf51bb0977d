timeout 300 python tools/host_profile3.py 2>&1 | grep -v amdgpu | grep "backward s\|pack_fused\|pack_segments\|call_base"
timeout 900 python -m pytest tests/test_parity_golden.py tests/test_many_segments.py tests/test_graph_mode_gpu.py tests/test_fullsize_reference.py -m gpu -q -x 2>&1 | tail -3
