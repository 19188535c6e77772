#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s3; mkdir -p $O
V=mhap_amd/lib/variants
run_emu() { # tag, env...
  tag=$1; shift
  echo "== $tag" >> $O/emu_c2.txt
  env "$@" timeout 300 python tools/emulate_rank.py 8 c2 6 2>>$O/emu_err.txt | tail -1 >> $O/emu_c2.txt
}
for rep in 1 2; do
  run_emu default A=1
  run_emu nostagger MHAP_W1_STAGGER=0
  run_emu stagger2x MHAP_W1_STAGGER=340000
  run_emu stagger_half MHAP_W1_STAGGER=85000
  run_emu iqt1 MHAP_LIB_PATH=$V/libmhaphip_iqt1.so
  run_emu iqt2 MHAP_LIB_PATH=$V/libmhaphip_iqt2.so
  run_emu iqt7 MHAP_LIB_PATH=$V/libmhaphip_iqt7.so
done
MHAP_MINHASH_PROF=1 timeout 300 python tools/emulate_rank.py 8 c2 3 2>&1 | grep prof > $O/emu_c2_prof.txt
for tag in default nostagger; do
  echo "== $tag" >> $O/bench_c2.txt
  if [ $tag = nostagger ]; then export MHAP_W1_STAGGER=0; else unset MHAP_W1_STAGGER; fi
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 >> $O/bench_c2.txt
done
unset MHAP_W1_STAGGER
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config3 or config4 or config5 or group or eager or rccl" 2>&1 | tail -30 ) > $O/pytest_a.log 2>&1
( timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_gpu_ranks.py -m gpu -q -x 2>&1 | tail -30 ) > $O/pytest_b.log 2>&1
echo done > $O/finished
