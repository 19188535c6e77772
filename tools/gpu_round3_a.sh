#!/bin/bash
# round-3 GPU check A: parity tests + A/B of the weight-kernel / MinHash-refine changes
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['candidates_per_step'], d['records_per_step'], d['records_sha256_sorted_lines'][:12])"; }
echo "== c2 default" | tee -a gpurun_out/r3a/ab.txt; run 2>&1 | tee -a gpurun_out/r3a/ab.txt
echo "== c2 refine0" | tee -a gpurun_out/r3a/ab.txt; MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_refine0.so run 2>&1 | tee -a gpurun_out/r3a/ab.txt
echo "== c2 default again" | tee -a gpurun_out/r3a/ab.txt; run 2>&1 | tee -a gpurun_out/r3a/ab.txt
echo "== c5slice default" | tee -a gpurun_out/r3a/ab.txt; run --config c5slice 2>&1 | tee -a gpurun_out/r3a/ab.txt
echo "== c5slice refine0" | tee -a gpurun_out/r3a/ab.txt; MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_refine0.so run --config c5slice 2>&1 | tee -a gpurun_out/r3a/ab.txt
echo "== c1" | tee -a gpurun_out/r3a/ab.txt; run --config c1 --steps 20 2>&1 | tee -a gpurun_out/r3a/ab.txt
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -v "^{" | tail -2 | tee -a gpurun_out/r3a/ab.txt
MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_refine0.so MHAP_MINHASH_PROF=1 timeout 300 python bench.py --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -v "^{" | tail -2 | tee -a gpurun_out/r3a/ab.txt
