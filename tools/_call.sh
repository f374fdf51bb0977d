timeout 300 python tools/host_profile3.py 2>&1 | grep -v amdgpu | head -60
