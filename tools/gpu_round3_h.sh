#!/bin/bash
# join kernel shapes: parity under each, then each shape on every config
export TMPDIR=/tmp
mkdir -p gpurun_out/r3h; rm -f gpurun_out/r3h/*
for mode in alone pair team; do
  MHAP_JOIN_MODE=$mode timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "oracle or overlap or config5 or golden or group" 2>&1 | tail -2 | tee -a gpurun_out/r3h/pytest_$mode.txt
done
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['overlap'], d['candidates_per_step'], d['records_per_step'])"; }
for cfg in c2 c5slice c1 c4slice; do
for v in default alone pair team; do
  echo "== $cfg $v" | tee -a gpurun_out/r3h/ab.txt
  if [ $v = default ]; then run --config $cfg 2>&1 | tee -a gpurun_out/r3h/ab.txt; else MHAP_JOIN_MODE=$v run --config $cfg 2>&1 | tee -a gpurun_out/r3h/ab.txt; fi
done
done
for v in default alone pair; do if [ $v = default ]; then python tools/emulate_rank.py 8 2>&1 | tail -1; else MHAP_JOIN_MODE=$v python tools/emulate_rank.py 8 2>&1 | tail -1; fi; done | tee -a gpurun_out/r3h/ab.txt
