#!/bin/bash
# compiler scheduling options on the whole library: MinHash and step time
export TMPDIR=/tmp
mkdir -p gpurun_out/r3o; rm -rf gpurun_out/r3o/*
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee gpurun_out/r3o/pytest.txt
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['records_per_step'])"; }
for v in default w1e5 norp default; do
  echo "== c2 $v" | tee -a gpurun_out/r3o/ab.txt
  if [ $v = default ]; then run --config c2 | tee -a gpurun_out/r3o/ab.txt; else MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_$v.so run --config c2 | tee -a gpurun_out/r3o/ab.txt; fi
done
for cfg in c1 c4slice c5slice; do echo "== $cfg" | tee -a gpurun_out/r3o/ab.txt; run --config $cfg | tee -a gpurun_out/r3o/ab.txt; done
python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3o/ab.txt
