#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s5; mkdir -p $O
V=mhap_amd/lib/variants
run_emu() { # file, tag, N, cfg, iters, env...
  f=$1; tag=$2; n=$3; c=$4; it=$5; shift 5
  echo "== $tag" >> $O/$f
  env "$@" timeout 900 python tools/emulate_rank.py $n $c $it 2>>$O/emu_err.txt | tail -1 >> $O/$f
}
for rep in 1 2; do
  run_emu emu_c2.txt nolines 8 c2 6 MHAP_INDEX_LINES=0
  run_emu emu_c2.txt perlane 8 c2 6 A=1
  run_emu emu_c2.txt quad16 8 c2 6 MHAP_LIB_PATH=$V/libmhaphip_quad16.so
  run_emu emu_c2.txt nolines_td8 8 c2 6 MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=8
  run_emu emu_c2.txt nolines_td32 8 c2 6 MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=32
  run_emu emu_c2.txt nolines_td128 8 c2 6 MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=128
  run_emu emu_c2.txt nolines_tdnone 8 c2 6 MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=1000000
done
run_emu emu_c2.txt n4_td16 4 c2 6 MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=16
run_emu emu_c2.txt n2_td16 2 c2 6 MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=16
for tag in nolines perlane nolines_td16; do
  echo "== $tag" >> $O/bench_c2.txt
  unset MHAP_INDEX_LINES MHAP_W1_TAIL_DIV
  case $tag in nolines) export MHAP_INDEX_LINES=0 ;; nolines_td16) export MHAP_INDEX_LINES=0 MHAP_W1_TAIL_DIV=16 ;; esac
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
done
unset MHAP_INDEX_LINES MHAP_W1_TAIL_DIV
run_emu emu_c4.txt nolines 8 c4 3 MHAP_INDEX_LINES=0
run_emu emu_c4.txt perlane 8 c4 3 A=1
MHAP_HOST_PROF=1 MHAP_INDEX_LINES=0 timeout 300 python tools/emulate_rank.py 8 c2 3 2>&1 | grep "host\]" | tail -40 > $O/host_prof_rank.txt
echo done > $O/finished
