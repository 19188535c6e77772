# the digest-stamped part of tools/profile_round.sh after a late kernel change: kernel stats, PMC traffic + pipe counters, the bench line and the
# MinHash attribution of the default workload (about seven minutes on the box): ROUND=r05 bash tools/profile_mini.sh
export TMPDIR=/tmp
R=${ROUND:-r06}
mkdir -p gpurun_out/prof_final gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_mix gpurun_out/$R
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o prof --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final/bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$c/bench.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/pmc_mix -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_mix/bench.log 2>&1
cp $(find gpurun_out/prof_final -name "prof_kernel_stats.csv" | head -1) gpurun_out/$R/${R}_rocprofv3_kernel_stats.csv
python tools/pmc_summary.py $(find gpurun_out/pmc_FETCH_SIZE -name "pmc_counter_collection.csv" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "pmc_counter_collection.csv" | head -1) > gpurun_out/$R/${R}_pmc_traffic.json
head -1 $(find gpurun_out/pmc_mix -name "pmc_counter_collection.csv" | head -1) > gpurun_out/$R/${R}_pmc_minhash_instmix.csv
grep -h "minhash_" $(find gpurun_out/pmc_mix -name "pmc_counter_collection.csv" | head -1) >> gpurun_out/$R/${R}_pmc_minhash_instmix.csv
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=pipes_$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag/bench.log 2>&1
done
python tools/pmc_pipes.py $(find gpurun_out/pmc_pipes_SQ_INSTS_LDS -name "pmc_counter_collection.csv" | head -1) $(find gpurun_out/pmc_pipes_GRBM_GUI_ACTIVE -name "pmc_counter_collection.csv" | head -1) > gpurun_out/$R/${R}_pmc_pipes.json
cp gpurun_out/$R/${R}_pmc_pipes.json profiles/${R}_pmc_pipes.json
cp gpurun_out/$R/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 --soak-seconds 20 > gpurun_out/$R/${R}_bench_final.json 2> gpurun_out/$R/bench_final.err
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof" | tail -4 > gpurun_out/$R/${R}_minhash_prof.txt
tail -1 gpurun_out/$R/${R}_bench_final.json | cut -c1-300
