#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s7; mkdir -p $O
V=mhap_amd/lib/variants
MHAP_HOST_PROF=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>&1 | grep "host\]" | tail -24 > $O/host_prof_c2.txt
for tag in default lines iqt2 iqt7 iqt2_lines iqt7_lines; do
  echo "== $tag" >> $O/bench_c2.txt
  unset MHAP_INDEX_LINES MHAP_LIB_PATH
  case $tag in lines) export MHAP_INDEX_LINES=1 ;; iqt2) export MHAP_LIB_PATH=$V/libmhaphip_iqt2.so ;; iqt7) export MHAP_LIB_PATH=$V/libmhaphip_iqt7.so ;;
    iqt2_lines) export MHAP_LIB_PATH=$V/libmhaphip_iqt2.so MHAP_INDEX_LINES=1 ;; iqt7_lines) export MHAP_LIB_PATH=$V/libmhaphip_iqt7.so MHAP_INDEX_LINES=1 ;; esac
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
done
echo done > $O/finished
