#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
echo skip-pytest
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['records_per_step'], d['records_sha256_sorted_lines'][:12])"; }
for v in default w1e4 w1e6 classic default; do
  echo "== c2 $v" | tee -a gpurun_out/r3d/ab.txt
  if [ $v = default ]; then run 2>&1 | tee -a gpurun_out/r3d/ab.txt; elif [ $v = classic ]; then MHAP_MINHASH=classic run 2>&1 | tee -a gpurun_out/r3d/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run 2>&1 | tee -a gpurun_out/r3d/ab.txt; fi
done
echo "== c1 default" | tee -a gpurun_out/r3d/ab.txt; run --config c1 --steps 20 2>&1 | tee -a gpurun_out/r3d/ab.txt
echo "== c1 classic" | tee -a gpurun_out/r3d/ab.txt; MHAP_MINHASH=classic run --config c1 --steps 20 2>&1 | tee -a gpurun_out/r3d/ab.txt
echo "== c2 12500 reads (one rank of 8) default" | tee -a gpurun_out/r3d/ab.txt; run --reads 12500 --steps 5 2>&1 | tee -a gpurun_out/r3d/ab.txt; python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3d/ab.txt
echo "== c2 12500 reads classic" | tee -a gpurun_out/r3d/ab.txt; MHAP_MINHASH=classic run --reads 12500 --steps 5 2>&1 | tee -a gpurun_out/r3d/ab.txt; python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3d/ab.txt
echo "== c4slice default" | tee -a gpurun_out/r3d/ab.txt; run --config c4slice 2>&1 | tee -a gpurun_out/r3d/ab.txt
echo "== c5slice default" | tee -a gpurun_out/r3d/ab.txt; run --config c5slice 2>&1 | tee -a gpurun_out/r3d/ab.txt
