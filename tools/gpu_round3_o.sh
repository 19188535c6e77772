#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3o; rm -rf gpurun_out/r3o/*
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['index_query'], d['records_per_step'], d['index_elements_per_step'])"; }
for v in default dq512; do
for cfg in c5slice; do
  echo "== $cfg $v" | tee -a gpurun_out/r3o/ab.txt
  if [ $v = default ]; then run --config $cfg | tee -a gpurun_out/r3o/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run --config $cfg | tee -a gpurun_out/r3o/ab.txt; fi
done; done
MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_dq512.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or tiers or repeat or config5" 2>&1 | grep -E "passed|failed" | tail -1
