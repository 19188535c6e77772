#!/bin/bash
# round 6, session 21: the merge buffer's fill in front of the ordered part (default) or right in front of the MinHash launch (rounds 4-5); split sweep
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s21; mkdir -p $O
run() {
  echo "== $*" >> $O/prof_c2.txt
  env "$@" MHAP_MINHASH_PROF=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof\] launch" | tail -3 >> $O/prof_c2.txt
  echo "== $*" >> $O/bench_c2.txt
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_c2.txt
}
for rep in 1 2 3; do
  run MHAP_X=0
  run MHAP_W1_MERGE_FILL_LATE=1
  run MHAP_ORDERED_SPLIT=35
  run MHAP_ORDERED_SPLIT=75
  run MHAP_ORDERED_SPLIT=0
done
echo done > $O/finished
