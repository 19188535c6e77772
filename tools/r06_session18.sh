#!/bin/bash
# round 6, session 18: the ordered kernel in two parts (MHAP_ORDERED_SPLIT = per cent of the strands before the MinHash launch)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s18; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8 ) > $O/pytest.log 2>&1
run() {
  echo "== $*" >> $O/bench_c2.txt
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_c2.txt
}
for rep in 1 2; do for sp in 0 30 45 55 70 100; do run MHAP_ORDERED_SPLIT=$sp; done; done
for rep in 1 2; do for sp in 0 30 55 100; do
  echo "== N=8 MHAP_ORDERED_SPLIT=$sp" >> $O/emu8.txt; MHAP_ORDERED_SPLIT=$sp timeout 600 python tools/emulate_rank.py 8 c2 10 2>/dev/null | tail -1 >> $O/emu8.txt
done; done
for sp in 0 55; do for n in 2 4; do
  echo "== N=$n MHAP_ORDERED_SPLIT=$sp" >> $O/emu_n.txt; MHAP_ORDERED_SPLIT=$sp timeout 600 python tools/emulate_rank.py $n c2 8 2>/dev/null | tail -1 >> $O/emu_n.txt
done; done
for c in c1 c4slice c5slice; do for sp in 0 55 0 55; do
  echo "== $c MHAP_ORDERED_SPLIT=$sp" >> $O/bench_other.txt
  MHAP_ORDERED_SPLIT=$sp timeout 900 python bench.py --config $c --no-cpu-baseline --soak-seconds 0 $( [ $c = c1 ] && echo "--steps 200 --warmup 20" ) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_other.txt
done; done
echo done > $O/finished
