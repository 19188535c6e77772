# instruction-mix counters of one kernel (substring match on the kernel name): bash tools/pmc_kernel.sh overlap_join_kernel
export TMPDIR=/tmp
K=${1:-overlap_join_kernel}
mkdir -p gpurun_out/pmc_k
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/pmc_k -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_k/bench.log 2>&1
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
print("$K", "dispatches", len(seen), "ns", dur, {k: "%.3e" % v for k, v in sorted(acc.items())})
PY
