# round 5, first look: (a) clock / power trace under the soak leg, (b) wave-clock attribution + mean shader clock of the MinHash kernel,
# (c) LDS / VALU counters for every kernel of a C2 step (ordered, weight, index build: VERDICT r04 item 3)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
O=gpurun_out/r05
python tools/smi_trace.py $O/smi_trace_c2_soak.txt 120 > $O/smi_summary.txt 2>&1 &
SMI=$!
sleep 3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --soak-seconds 15 > $O/bench_soak.json 2> $O/bench_soak.err
sleep 2
kill -TERM $SMI; wait $SMI
tail -1 $O/bench_soak.json | cut -c1-600
cat $O/smi_summary.txt
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --soak-seconds 0 > $O/bench_prof.json 2> $O/minhash_prof.txt
grep "w1 prof" $O/minhash_prof.txt | tail -4
rocprofv3 -L 2>/dev/null | grep -i -E "SQ_.*LDS|SQ_ACTIVE_INST|SQ_INST_CYCLES|GRBM_GUI|SQ_WAIT|SQ_INSTS_" | cut -c1-160 | sort -u > $O/counters_list.txt
wc -l $O/counters_list.txt
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag/bench.log 2>&1
  f=$(find gpurun_out/pmc_$tag -name "pmc_counter_collection.csv" | head -1)
  python tools/pmc_per_kernel.py $f > $O/pmc_per_kernel_$tag.txt 2>&1
  cat $O/pmc_per_kernel_$tag.txt | cut -c1-400
done
