#!/bin/bash
# VGPR / SGPR / scratch / occupancy of the kernels in one .hip file (substring filter as 2nd argument): bash tools/vgpr_report.sh file.hip minhash
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -c "$1" -o /tmp/vgpr_report.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" | paste - - - - - \
  | sed 's/remark: [^ ]* //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | grep -E "${2:-.}" \
  | sed 's/[a-z_]*\.hip:[0-9]*:[0-9]*://g' | tr -s ' \t' ' '
