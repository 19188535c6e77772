#!/bin/bash
# A/B builds of the library with extra -D flags: bash tools/build_variant.sh NAME -DMH_REFINE=0 ...  ->  mhap_amd/lib/variants/libmhaphip_NAME.so
# (run a variant with MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_NAME.so; mhap_amd/lib/ is git-ignored but travels with gpurun)
set -e
cd "$(dirname "$0")/../mhap_amd/csrc"
name=$1; shift
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-pass-failed -mllvm -amdgpu-sched-strategy=max-memory-clause "$@" \
  sketch_kernels.hip search_kernels.hip mhap_capi.hip mhap_dist.hip mhap_ingest.hip host_util.cpp -o ../lib/variants/libmhaphip_$name.so -lz -ldl
echo mhap_amd/lib/variants/libmhaphip_$name.so
