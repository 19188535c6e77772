# instruction-mix counters of the MinHash kernel for several library builds: bash tools/pmc_ab.sh lib1.so lib2.so ...
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pmc_ab; mkdir -p /tmp/pmc_ab
  MHAP_LIB_PATH=$v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d /tmp/pmc_ab -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --soak-seconds 0 > /tmp/pmc_ab/bench.log 2>&1
  echo "== $v"
  python tools/pmc_per_kernel.py $(find /tmp/pmc_ab -name "pmc_counter_collection.csv" | head -1) | grep "minhash_w1_kernel" | cut -c1-600
  MHAP_LIB_PATH=$v MHAP_MINHASH_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof" | tail -2
done
