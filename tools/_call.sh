cd /root/repo
timeout 300 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids | head -2
