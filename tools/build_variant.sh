#!/bin/bash
# A/B builds of the library with extra -D flags: bash tools/build_variant.sh NAME -DMH_REFINE=0 ... [--only sketch_kernels.hip]
#   ->  mhap_amd/lib/variants/libmhaphip_NAME.so  (only the kernel files are recompiled with the flags; the other objects are shared)
# (run a variant with MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_NAME.so; mhap_amd/lib/ is git-ignored but travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m mhap_amd.build --variant "$name" "$@"
