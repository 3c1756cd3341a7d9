VB2_SPLIT=0 VB2_B=48 python tools/stamps.py 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=0 python tools/quick.py --batches 48,8,4,1 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=1 python tools/quick.py --batches 48 --no-optimize --no-parity 2>&1 | grep -v amdgpu.ids
