VB2_SPLIT=0 VB2_PAIR_MODE=single48 VB2_PAIR_TAG=base python tools/two_wg_pair.py 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=1 VB2_PAIR_MODE=single48 VB2_PAIR_TAG=split python tools/two_wg_pair.py 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=0 python tools/quick.py --batches 48,40,32,24,16,8 --no-optimize 2>&1 | grep -v amdgpu.ids
VB2_SPLIT=1 python tools/quick.py --batches 48,40,32,24,16,8 --no-optimize 2>&1 | grep -v amdgpu.ids
