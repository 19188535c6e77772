#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3m; rm -rf gpurun_out/r3m/*
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['candidates_per_step'], d['records_per_step'], d['overlap_slow_pairs_per_step'])"; }
for v in default ojpad1k ojpad3k; do
for cfg in c2; do
  echo "== $cfg $v" | tee -a gpurun_out/r3m/ab.txt
  if [ $v = default ]; then run --config $cfg 2>&1 | tee -a gpurun_out/r3m/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run --config $cfg 2>&1 | tee -a gpurun_out/r3m/ab.txt; fi
done
if [ $v = default ]; then python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3m/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3m/ab.txt; fi
done
