#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3l; rm -rf gpurun_out/r3l/*
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r3l/pytest.txt
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3l/bench_c4.json 2> gpurun_out/r3l/bench_c4.err
(bash tools/e2e_probe.sh c2; bash tools/e2e_probe.sh c4) > gpurun_out/r3l/e2e.txt 2>&1
timeout 600 python bench.py > gpurun_out/r3l/bench_c2_full.json 2> gpurun_out/r3l/bench_c2_full.err
