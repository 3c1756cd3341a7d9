VB2_SPLIT=0 VB2_B=48 python tools/stamps.py 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=0 VB2_B=48 python tools/stamps.py 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=1 VB2_B=48 python tools/stamps.py 2>&1 | grep -v amdgpu.ids
