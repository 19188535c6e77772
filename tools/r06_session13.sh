#!/bin/bash
# round 6, session 13: where a rank of eight's step goes (host timeline + kernel timeline), baseline bench line on this box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_s13; mkdir -p $O
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 > $O/bench_c2.json 2> $O/bench_c2.err
MHAP_HOST_PROF=1 timeout 600 python tools/emulate_rank.py 8 c2 3 > $O/emu8_hostprof.json 2> $O/emu8_hostprof.err
timeout 600 python tools/emulate_rank.py 8 c2 6 > $O/emu8.json 2> $O/emu8.err
mkdir -p $O/tl
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o tl --output-format csv -- python tools/emulate_rank.py 8 c2 2 > $O/tl/log 2>&1
python - <<'PY' > gpurun_out/r06_s13/emu8_kernel_timeline.txt
import csv, glob
f = glob.glob("gpurun_out/r06_s13/tl/**/tl_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e6 if prev_end else 0.0
    print("%8.3f  dur %8.3f  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, r["Kernel_Name"][:70]))
    prev_end = e
PY
rm -rf $O/tl
echo done > $O/finished
