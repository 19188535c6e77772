#!/bin/bash
# round 6, session 14: small-index build shapes + device-made query list (A/B against the old behaviour through the env switches), the clock probe
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s14; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 ) > $O/pytest.log 2>&1
for rep in 1 2; do
  echo "== new" >> $O/emu8_ab.txt; timeout 600 python tools/emulate_rank.py 8 c2 6 2>/dev/null | tail -1 >> $O/emu8_ab.txt
  echo "== old (MHAP_INDEX_BINS_SHAPE=2 MHAP_INDEX_TILE=4096 MHAP_QUERY_LIST_HOST=1)" >> $O/emu8_ab.txt
  MHAP_INDEX_BINS_SHAPE=2 MHAP_INDEX_TILE=4096 MHAP_QUERY_LIST_HOST=1 timeout 600 python tools/emulate_rank.py 8 c2 6 2>/dev/null | tail -1 >> $O/emu8_ab.txt
done
MHAP_HOST_PROF=1 timeout 600 python tools/emulate_rank.py 8 c2 3 2>&1 >/dev/null | grep "host\]" | tail -24 > $O/emu8_hostprof.txt
for n in 2 4; do timeout 600 python tools/emulate_rank.py $n c2 4 2>/dev/null | tail -1 >> $O/emu_n.txt; done
timeout 600 python tools/emulate_rank.py 8 c4 3 2>/dev/null | tail -1 >> $O/emu_n.txt
for rep in 1 2; do
  echo "== new" >> $O/bench_ab.txt; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_ab.txt
  echo "== old bins shape" >> $O/bench_ab.txt; MHAP_INDEX_BINS_SHAPE=2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_ab.txt
done
MHAP_MINHASH_PROF=1 timeout 900 python tools/w1_clock_probe.py 8 c2 > $O/w1_clock_probe.json 2> $O/w1_clock_probe.err
timeout 900 python tools/w1_clock_probe.py 8 c2 > $O/w1_clock_probe_noprof.json 2>/dev/null
echo done > $O/finished
