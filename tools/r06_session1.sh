#!/bin/bash
# round 6, GPU session 1: the full -m gpu suite on the line-table index + seeded row items, then A/Bs of both on one box
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s1; mkdir -p $O
V=mhap_amd/lib/variants
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 -x 2>&1 | tail -40 ) > $O/pytest.log 2>&1
for rep in 1 2; do
  for tag in default nolines noseed lb2 lb8; do
    case $tag in
      default) env_="" ;;
      nolines) env_="MHAP_INDEX_LINES=0" ;;
      noseed) env_="MHAP_LIB_PATH=$V/libmhaphip_noseed.so" ;;
      lb2) env_="MHAP_LIB_PATH=$V/libmhaphip_lb2.so" ;;
      lb8) env_="MHAP_LIB_PATH=$V/libmhaphip_lb8.so" ;;
    esac
    echo "== $tag rep $rep" >> $O/emu_c2.txt
    env $env_ timeout 300 python tools/emulate_rank.py 8 c2 6 2>>$O/emu_err.txt | tail -1 >> $O/emu_c2.txt
  done
done
MHAP_MINHASH_PROF=1 timeout 300 python tools/emulate_rank.py 8 c2 3 > $O/emu_c2_prof.txt 2>&1
MHAP_MINHASH_PROF=1 MHAP_LIB_PATH=$V/libmhaphip_noseed.so timeout 300 python tools/emulate_rank.py 8 c2 3 > $O/emu_c2_prof_noseed.txt 2>&1
CONFIG=c2 STEPS=10 bash tools/ab_kernels.sh 2 mhap_amd/lib/libmhaphip.so $V/libmhaphip_noseed.so > $O/ab_c2.txt 2>&1
for tag in default nolines; do
  echo "== $tag" >> $O/bench_c2_lines.txt
  if [ $tag = nolines ]; then export MHAP_INDEX_LINES=0; else unset MHAP_INDEX_LINES; fi
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2_lines.txt
done
unset MHAP_INDEX_LINES
for tag in default nolines; do
  echo "== $tag" >> $O/emu_c4.txt
  if [ $tag = nolines ]; then export MHAP_INDEX_LINES=0; else unset MHAP_INDEX_LINES; fi
  timeout 900 python tools/emulate_rank.py 8 c4 3 2>>$O/emu_err.txt | tail -1 >> $O/emu_c4.txt
done
echo done > $O/finished
