timeout 300 python tools/overlap_probe.py serial seq2 seq4 seq8 2>&1 | grep -v amdgpu
