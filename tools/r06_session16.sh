#!/bin/bash
# round 6, session 16: why the MinHash kernel is 4 ms faster behind the ordered kernel (clock or clocks per row?), and where the order pays
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s16; mkdir -p $O
for rep in 1 2; do
  for of in 0 1; do
    echo "== MHAP_ORDERED_FIRST=$of" >> $O/prof_c2.txt
    MHAP_ORDERED_FIRST=$of MHAP_MINHASH_PROF=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof" | tail -4 >> $O/prof_c2.txt
  done
done
for n in 2 4; do for of in 0 1; do
  echo "== N=$n MHAP_ORDERED_FIRST=$of" >> $O/emu_n.txt; MHAP_ORDERED_FIRST=$of timeout 600 python tools/emulate_rank.py $n c2 8 2>/dev/null | tail -1 >> $O/emu_n.txt
done; done
for c in c4slice c5slice; do for of in 0 1 0 1; do
  echo "== $c MHAP_ORDERED_FIRST=$of" >> $O/bench_other.txt
  MHAP_ORDERED_FIRST=$of timeout 900 python bench.py --config $c --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_other.txt
done; done
echo done > $O/finished
