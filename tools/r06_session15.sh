#!/bin/bash
# round 6, session 15: ordered kernel first (the eager add's order) against the default order; C1 host + kernel timeline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s15; mkdir -p $O
for rep in 1 2; do
  echo "== default" >> $O/emu8_ab.txt; timeout 600 python tools/emulate_rank.py 8 c2 10 2>/dev/null | tail -1 >> $O/emu8_ab.txt
  echo "== MHAP_ORDERED_FIRST=1" >> $O/emu8_ab.txt; MHAP_ORDERED_FIRST=1 timeout 600 python tools/emulate_rank.py 8 c2 10 2>/dev/null | tail -1 >> $O/emu8_ab.txt
  echo "== old (MHAP_INDEX_BINS_SHAPE=2 MHAP_INDEX_TILE=4096 MHAP_QUERY_LIST_HOST=1)" >> $O/emu8_ab.txt
  MHAP_INDEX_BINS_SHAPE=2 MHAP_INDEX_TILE=4096 MHAP_QUERY_LIST_HOST=1 timeout 600 python tools/emulate_rank.py 8 c2 10 2>/dev/null | tail -1 >> $O/emu8_ab.txt
done
for rep in 1 2; do
  echo "== default" >> $O/bench_ab.txt; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_ab.txt
  echo "== MHAP_ORDERED_FIRST=1" >> $O/bench_ab.txt; MHAP_ORDERED_FIRST=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" >> $O/bench_ab.txt
done
timeout 600 python bench.py --config c1 --steps 200 --warmup 20 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 > $O/bench_c1.json
MHAP_HOST_PROF=1 timeout 600 python bench.py --config c1 --steps 3 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "host\]" | tail -40 > $O/c1_hostprof.txt
mkdir -p $O/tl
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o tl --output-format csv -- python bench.py --config c1 --steps 3 --warmup 2 --no-cpu-baseline --soak-seconds 0 > $O/tl/log 2>&1
python - <<'PY' > gpurun_out/r06_s15/c1_kernel_timeline.txt
import csv, glob
f = glob.glob("gpurun_out/r06_s15/tl/**/tl_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-50:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e6 if prev_end else 0.0
    print("%8.3f  dur %8.3f  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, r["Kernel_Name"][:70]))
    prev_end = e
PY
rm -rf $O/tl
echo done > $O/finished
