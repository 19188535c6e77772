#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
MHAP_GROUP_TRANSPORT_TEST=1 timeout 600 python tools/gpu_group_check.py 2>&1 | tail -15 | tee gpurun_out/r3b/group.txt
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['records_per_step'], d['records_sha256_sorted_lines'][:12])"; }
for v in default w4q128 w5q128; do
  echo "== c2 $v" | tee -a gpurun_out/r3b/ab.txt
  if [ $v = default ]; then run 2>&1 | tee -a gpurun_out/r3b/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run 2>&1 | tee -a gpurun_out/r3b/ab.txt; fi
done
