export TMPDIR=/tmp
mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/kt/bench.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/kt/**/kt_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
