set -x
VB2_PAIR_MODE=single48 VB2_PAIR_TAG=base python tools/two_wg_pair.py
VB2_PAIR_TAG=pair16 python tools/two_wg_pair.py
VB2_GEOM2=8,1 VB2_PAIR_TAG=pair8 python tools/two_wg_pair.py
VB2_GEOM2=10,1 VB2_PAIR_TAG=pair10_4wps python tools/two_wg_pair.py
VB2_LIB_PATH=build_variants/g2w10/libvb2.so VB2_PAIR_TAG=g2w10 python tools/two_wg_pair.py
VB2_LIB_PATH=build_variants/g2w12/libvb2.so VB2_PAIR_TAG=g2w12 python tools/two_wg_pair.py
VB2_PAIR_MODE=single48 VB2_PAIR_TAG=base2 python tools/two_wg_pair.py
