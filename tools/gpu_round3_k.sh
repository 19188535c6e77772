#!/bin/bash
# counting-sort inverted index + dense second tier: parity, then the step on every config
export TMPDIR=/tmp
mkdir -p gpurun_out/r3k; rm -rf gpurun_out/r3k/*
echo skip-tests
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['candidates_per_step'], d['records_per_step'], d.get('index_elements_per_step'))"; }
for cfg in c2 c5slice c1 c4slice; do
  echo "== $cfg" | tee -a gpurun_out/r3k/ab.txt
  run --config $cfg 2>&1 | tee -a gpurun_out/r3k/ab.txt
done
python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3k/ab.txt
for i in 1 2; do timeout 400 python tools/check_elements.py c5slice 2>&1 | tail -1 | tee -a gpurun_out/r3k/ab.txt; done
