#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3n; rm -rf gpurun_out/r3n/*
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['records_per_step'])"; }
for cfg in c2 c2 c4slice c5slice c1; do echo "== $cfg"; run --config $cfg; done | tee gpurun_out/r3n/ab.txt
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee gpurun_out/r3n/pytest.txt
timeout 900 python tests/fuzz_parity.py 250 980000 2>&1 | grep -E "FAIL|failures" | head -5 | tee gpurun_out/r3n/fuzz.txt
