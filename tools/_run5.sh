for i in 1 2; do
VB2_LIB_PATH=build_variants/base/libvb2.so python tools/quick.py --batches 48,4,1 --no-parity 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
VB2_SPLIT=0 python tools/quick.py --batches 48,4,1 --no-parity 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done
