# instruction-mix counters of the MinHash kernel: bit-sliced main loop only (variant 280) vs the default kernel
export TMPDIR=/tmp
for tag in default mainonly; do
  if [ $tag = mainonly ]; then export MHAP_MINHASH_VARIANT=280 MHAP_BS_SEED=0 MHAP_BS_MINREM=512; else unset MHAP_MINHASH_VARIANT MHAP_BS_SEED MHAP_BS_MINREM; fi
  mkdir -p gpurun_out/pmc_mh_$tag
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/pmc_mh_$tag -o pmc --output-format csv -- python bench.py --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_mh_$tag/bench.log 2>&1
  python - <<PY
import csv, collections
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open("gpurun_out/pmc_mh_$tag/pmc_counter_collection.csv")):
    if "minhash_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
print("$tag", {k: "%.3e" % v for k, v in sorted(acc.items())})
PY
done
