# kernel timeline of the last bench step: start offset, duration and gap to the previous kernel (ms)
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/tl/bench.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/tl_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last kmer_weight/hash launch group: take the final 40 kernels
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e6 if prev_end else 0.0
    print("%8.3f  dur %8.3f  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, r["Kernel_Name"][:50]))
    prev_end = e
PY
