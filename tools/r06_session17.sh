#!/bin/bash
# round 6, session 17: what gives the MinHash launch its higher clock behind the ordered kernel — the kernel before it, or no idle gap before it?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s17; mkdir -p $O
run() {  # label, env...
  echo "== $*" >> $O/prof_c2.txt
  env "$@" MHAP_MINHASH_PROF=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof\] launch" | tail -3 >> $O/prof_c2.txt
  echo "== $*" >> $O/bench_c2.txt
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_c2.txt
}
for rep in 1 2; do
  run MHAP_ORDERED_FIRST=0
  run MHAP_ORDERED_FIRST=1
  run MHAP_ORDERED_FIRST=2
  run MHAP_W1_PREFILL=5
  run MHAP_W1_PREFILL=25
  run MHAP_W1_PREFILL=100
done
echo done > $O/finished
