#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['candidates_per_step'], d['records_per_step'])"; }
for cfg in c2 c5slice c1; do
for v in default ojjoin ojnosearch; do
  echo "== $cfg $v" | tee -a gpurun_out/r3g/ab.txt
  if [ $v = default ]; then run --config $cfg 2>&1 | tee -a gpurun_out/r3g/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run --config $cfg 2>&1 | tee -a gpurun_out/r3g/ab.txt; fi
done; done
python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3g/ab.txt
