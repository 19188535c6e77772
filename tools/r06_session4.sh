#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s4; mkdir -p $O
run_emu() { # file, tag, N, cfg, iters, env...
  f=$1; tag=$2; n=$3; c=$4; it=$5; shift 5
  echo "== $tag" >> $O/$f
  env "$@" timeout 900 python tools/emulate_rank.py $n $c $it 2>>$O/emu_err.txt | tail -1 >> $O/$f
}
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "index or tier or golden or config2 or large_num_hashes or group or eager or rccl" 2>&1 | tail -15 ) > $O/pytest_a.log 2>&1
for rep in 1 2; do
  run_emu emu_c2.txt default 8 c2 6 A=1
  run_emu emu_c2.txt nolines 8 c2 6 MHAP_INDEX_LINES=0
  run_emu emu_c2.txt taildiv2 8 c2 6 MHAP_W1_TAIL_DIV=2
  run_emu emu_c2.txt taildiv4 8 c2 6 MHAP_W1_TAIL_DIV=4
  run_emu emu_c2.txt taildiv16 8 c2 6 MHAP_W1_TAIL_DIV=16
done
run_emu emu_c2.txt n4 4 c2 6 A=1
run_emu emu_c2.txt n2 2 c2 6 A=1
for tag in default nolines; do
  echo "== $tag" >> $O/bench_c2.txt
  if [ $tag = nolines ]; then export MHAP_INDEX_LINES=0; else unset MHAP_INDEX_LINES; fi
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
done
unset MHAP_INDEX_LINES
run_emu emu_c4.txt default 8 c4 3 A=1
run_emu emu_c4.txt nolines 8 c4 3 MHAP_INDEX_LINES=0
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 > $O/bench_c4.txt
echo done > $O/finished
