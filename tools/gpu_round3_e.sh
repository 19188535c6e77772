#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed_ingest or group_of_ranks" 2>&1 | tail -15 | tee gpurun_out/r3e/pytest.txt
bash tools/e2e_probe.sh c2 2>&1 | tee gpurun_out/r3e/e2e_c2.txt
bash tools/e2e_probe.sh c2 --devices 0,0 2>&1 | grep -v "\[host\]" | tee gpurun_out/r3e/e2e_c2_2ranks.txt
