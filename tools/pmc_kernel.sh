# instruction-mix counters of one kernel (substring match on the kernel name): [CONFIG=c5slice] bash tools/pmc_kernel.sh overlap_join_kernel
export TMPDIR=/tmp
K=${1:-overlap_join_kernel}
C=${CONFIG:-c2}
rm -rf gpurun_out/pmc_k
mkdir -p gpurun_out/pmc_k
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/pmc_k -o pmc --output-format csv -- python bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_k/bench.log 2>&1
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(float); dur = 0
f = glob.glob("gpurun_out/pmc_k/**/pmc_counter_collection.csv", recursive=True)[0]
seen=set()
for r in csv.DictReader(open(f)):
    if "$K" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
import json
try:
    b = json.loads([l for l in open("gpurun_out/pmc_k/bench.log").read().splitlines() if l.startswith("{")][-1]); pairs = b.get("candidates_per_step")
except Exception: pairs = None
print("$C", "candidate pairs per step", pairs, "(the counters are sums over every launch of the run: the timed step and the fenced extra one)")
print("$K", "dispatches", len(seen), "ns", dur, {k: "%.3e" % v for k, v in sorted(acc.items())})
PY
