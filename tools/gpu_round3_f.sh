#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
bash tools/e2e_probe.sh c2 2>&1 | tee gpurun_out/r3f/e2e_c2.txt
df -h /tmp | tail -1
bash tools/e2e_probe.sh c4 2>&1 | grep -v "\[host\]" | tee gpurun_out/r3f/e2e_c4.txt
