#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s6; mkdir -p $O
run_emu() { f=$1; tag=$2; n=$3; c=$4; it=$5; shift 5; echo "== $tag" >> $O/$f; env "$@" timeout 900 python tools/emulate_rank.py $n $c $it 2>>$O/emu_err.txt | tail -1 >> $O/$f; }
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -40 ) > $O/pytest.log 2>&1
for rep in 1 2; do
  for tag in default countown; do
    echo "== $tag" >> $O/bench_c2.txt
    unset MHAP_COUNT_OWN; [ $tag = countown ] && export MHAP_COUNT_OWN=1
    timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
  done
done
unset MHAP_COUNT_OWN
run_emu emu.txt n8 8 c2 6 A=1
run_emu emu.txt n4 4 c2 6 A=1
run_emu emu.txt n2 2 c2 6 A=1
run_emu emu.txt n8c4 8 c4 3 A=1
echo done > $O/finished
