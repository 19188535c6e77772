#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s11; mkdir -p $O
for rep in 1 2; do
for tag in lines nolines; do
  echo "== $tag" >> $O/bench_c5slice.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 900 python bench.py --config c5slice --steps 4 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c5slice.txt
done
done
unset MHAP_INDEX_LINES
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "line_table or config5 or dense or tier or huge" 2>&1 | tail -8 ) > $O/pytest_a.log 2>&1
for tag in lines nolines; do
  echo "== $tag" >> $O/bench_c5rank.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 1500 python bench.py --config c5rank --steps 2 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c5rank.txt
done
echo done > $O/finished
