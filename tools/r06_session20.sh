#!/bin/bash
# round 6, session 20: the new default (55 % of the ordered kernel + the weighted MinHash launch in front of the weight-1 launch) against rounds 1-5's order
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s20; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8 ) > $O/pytest.log 2>&1
for rep in 1 2; do for sp in default 0; do
  E=""; [ $sp = 0 ] && E="MHAP_ORDERED_SPLIT=0"
  echo "== c2 $sp" >> $O/bench.txt
  env $E timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench.txt
  for n in 8 4 2; do
    echo "== N=$n $sp" >> $O/emu.txt; env $E timeout 600 python tools/emulate_rank.py $n c2 10 2>/dev/null | tail -1 >> $O/emu.txt
  done
  for c in c1 c4slice c5slice; do
    echo "== $c $sp" >> $O/bench.txt
    env $E timeout 900 python bench.py --config $c --no-cpu-baseline --soak-seconds 0 $( [ $c = c1 ] && echo "--steps 200 --warmup 20" ) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench.txt
  done
done; done
echo "== N=8 c4 default" >> $O/emu.txt; timeout 600 python tools/emulate_rank.py 8 c4 4 2>/dev/null | tail -1 >> $O/emu.txt
echo "== N=8 c4 0" >> $O/emu.txt; MHAP_ORDERED_SPLIT=0 timeout 600 python tools/emulate_rank.py 8 c4 4 2>/dev/null | tail -1 >> $O/emu.txt
echo done > $O/finished
