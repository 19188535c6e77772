#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3o; rm -rf gpurun_out/r3o/*
run() { python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['index_query'], d['records_per_step'], d['index_elements_per_step'])"; }
for cfg in c4 c2 c4slice; do echo "== $cfg" | tee -a gpurun_out/r3o/ab.txt; run --config $cfg | tee -a gpurun_out/r3o/ab.txt; done
MHAP_QUERY_CHUNK=4096 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "oracle or config2 or tiers or dense" 2>&1 | grep -E "passed|failed" | tail -1
MHAP_QUERY_CHUNK=128 MHAP_INDEX_MID=1 timeout 600 python tests/fuzz_parity.py 150 998000 2>&1 | grep -E "FAIL|failures" | head -3
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
