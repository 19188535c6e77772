#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s10; mkdir -p $O
run_emu() { f=$1; tag=$2; n=$3; c=$4; it=$5; shift 5; echo "== $tag" >> $O/$f; env "$@" timeout 900 python tools/emulate_rank.py $n $c $it 2>>$O/emu_err.txt | tail -1 >> $O/$f; }
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "line_table or candidate_paths or golden or config2 or config1 or config4 or config5 or index or tier or dense or num_min" 2>&1 | tail -8 ) > $O/pytest_a.log 2>&1
for rep in 1 2; do
  run_emu emu.txt n8_lines 8 c2 6 A=1
  run_emu emu.txt n8_nolines 8 c2 6 MHAP_INDEX_LINES=0
done
for rep in 1 2; do
for tag in nolines lines; do
  echo "== $tag" >> $O/bench_c2.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
done
done
unset MHAP_INDEX_LINES
run_emu emu.txt n8c4_lines 8 c4 3 MHAP_INDEX_LINES=1
for tag in nolines lines; do
  echo "== $tag" >> $O/bench_c1.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 600 python bench.py --config c1 --steps 200 --warmup 20 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c1.txt
done
for tag in lines nolines; do
  echo "== $tag" >> $O/bench_c4.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c4.txt
done
unset MHAP_INDEX_LINES
for tag in lines nolines; do
  echo "== $tag" >> $O/bench_c5slice.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 900 python bench.py --config c5slice --steps 4 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c5slice.txt
done
echo done > $O/finished
