#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3m; rm -rf gpurun_out/r3m/*
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['candidates_per_step'], d['records_per_step'], d['overlap_slow_pairs_per_step'])"; }
for cfg in c2 c4slice c5slice c1 c4; do
  echo "== $cfg" | tee -a gpurun_out/r3m/ab.txt
  run --config $cfg 2>&1 | tee -a gpurun_out/r3m/ab.txt
done
python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3m/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_num_min or tiers_on_huge or dense_second or shared_repeat or config5 or config2" 2>&1 | tail -3
