#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s9; mkdir -p $O
run_emu() { f=$1; tag=$2; n=$3; c=$4; it=$5; shift 5; echo "== $tag" >> $O/$f; env "$@" timeout 900 python tools/emulate_rank.py $n $c $it 2>>$O/emu_err.txt | tail -1 >> $O/$f; }
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "line_table or candidate_paths or golden or config2 or config1 or config5 or index" 2>&1 | tail -8 ) > $O/pytest_a.log 2>&1
for rep in 1 2; do
  run_emu emu.txt n8_lines 8 c2 6 A=1
  run_emu emu.txt n8_nolines 8 c2 6 MHAP_INDEX_LINES=0
done
run_emu emu.txt n4_lines 4 c2 6 A=1
run_emu emu.txt n2_lines 2 c2 6 MHAP_INDEX_LINES=1
run_emu emu.txt n2_nolines 2 c2 6 MHAP_INDEX_LINES=0
for rep in 1 2; do
for tag in nolines lines; do
  echo "== $tag" >> $O/bench_c2.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
done
done
unset MHAP_INDEX_LINES
run_emu emu.txt n8c4_lines 8 c4 3 MHAP_INDEX_LINES=1
run_emu emu.txt n8c4_nolines 8 c4 3 MHAP_INDEX_LINES=0
for tag in nolines lines; do
  echo "== $tag" >> $O/bench_c1.txt
  if [ $tag = lines ]; then export MHAP_INDEX_LINES=1; else export MHAP_INDEX_LINES=0; fi
  timeout 600 python bench.py --config c1 --steps 200 --warmup 20 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c1.txt
done
echo done > $O/finished
