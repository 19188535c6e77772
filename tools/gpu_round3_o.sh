#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3o; rm -rf gpurun_out/r3o/*
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['overlap'], d['records_per_step'])"; }
for v in default ojw8 ojw6 oju2 oju4; do
for cfg in c5slice c2; do
  echo "== $cfg $v" | tee -a gpurun_out/r3o/ab.txt
  if [ $v = default ]; then run --config $cfg | tee -a gpurun_out/r3o/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run --config $cfg | tee -a gpurun_out/r3o/ab.txt; fi
done; done
