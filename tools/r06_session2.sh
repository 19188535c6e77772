#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s2; mkdir -p $O
timeout 300 tools/bin/line_gather_probe > $O/line_gather_probe.txt 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config3 or config4 or config5 or group or eager or rccl" 2>&1 | tail -30 ) > $O/pytest_a.log 2>&1
( timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_gpu_ranks.py -m gpu -q -x 2>&1 | tail -30 ) > $O/pytest_b.log 2>&1
echo done > $O/finished
