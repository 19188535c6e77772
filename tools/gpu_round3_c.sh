#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_config2" 2>&1 | tail -3 | tee gpurun_out/r3c/pytest.txt
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['records_per_step'], d['records_sha256_sorted_lines'][:12])"; }
for v in default w5 default w5; do
  echo "== c2 $v" | tee -a gpurun_out/r3c/ab.txt
  if [ $v = default ]; then run 2>&1 | tee -a gpurun_out/r3c/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run 2>&1 | tee -a gpurun_out/r3c/ab.txt; fi
done
echo "== c5slice default" | tee -a gpurun_out/r3c/ab.txt; run --config c5slice 2>&1 | tee -a gpurun_out/r3c/ab.txt
echo "== c5slice w5" | tee -a gpurun_out/r3c/ab.txt; MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_w5.so run --config c5slice 2>&1 | tee -a gpurun_out/r3c/ab.txt
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -v "^{" | tail -2 | tee -a gpurun_out/r3c/ab.txt
