#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3o; rm -rf gpurun_out/r3o/*
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['ordered'], d['records_per_step'])"; }
for v in default ordt512 ordt128; do
  echo "== c2 $v" | tee -a gpurun_out/r3o/ab.txt
  if [ $v = default ]; then run --config c2 | tee -a gpurun_out/r3o/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run --config c2 | tee -a gpurun_out/r3o/ab.txt; fi
done
