#!/bin/bash
# round 6: the final evidence run on the GPU box — the profile round, the GPU test suite, the fuzz campaign
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
ROUND=r06 bash tools/profile_round.sh > gpurun_out/r06/profile_round.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -30 ) > gpurun_out/r06/r06_gpu_pytest.log 2>&1
N1=300 NL1=300 NL2=200 NL3=150 NL4=100 N2=100 N3=60 N4=50 N5=80 N6=60 N7=60 N8=100 N9=60 ROUND=r06 bash tools/fuzz_campaign.sh > gpurun_out/r06/fuzz.log 2>&1
echo done > gpurun_out/r06/finished
