# final profiles of a round: rocprofv3 kernel stats + PMC traffic on the default workload, the other configurations' bench lines, one rank of an
# N-GPU job, the native driver end to end (run on the GPU box); ROUND=r04 bash tools/profile_round.sh
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
# pipe counters of every kernel (LDS pipe, bank conflicts, VALU / SALU instructions): two more passes of eight counters
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=pipes_$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag/bench.log 2>&1
done
python tools/pmc_pipes.py $(find gpurun_out/pmc_pipes_SQ_INSTS_LDS -name "pmc_counter_collection.csv" | head -1) $(find gpurun_out/pmc_pipes_GRBM_GUI_ACTIVE -name "pmc_counter_collection.csv" | head -1) > gpurun_out/$R/${R}_pmc_pipes.json
cp gpurun_out/$R/${R}_pmc_pipes.json profiles/${R}_pmc_pipes.json
cp gpurun_out/$R/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json   # bench.py reads the byte counts from profiles/ (same source digest)
# the committed bench line, with a clock / power trace of the card under its soak leg (tools/smi_trace.py) and the MinHash kernel's own
# wave-clock attribution + mean shader clock beside it
# (round 6: the sampler is started by the bench line itself, on the card HIP reports for device 0 — by PCI address, not by "the busiest card",
#  which on a shared box was somebody else's — and a capture without clock or power samples is a failed capture: soak.smi_ok in the line)
timeout 900 python bench.py --steps 20 --warmup 5 --soak-seconds 20 --smi-trace gpurun_out/$R/${R}_smi_trace_c2.txt > gpurun_out/$R/${R}_bench_final.json 2> gpurun_out/$R/bench_final.err
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof" | tail -4 > gpurun_out/$R/${R}_minhash_prof.txt
tail -1 gpurun_out/$R/${R}_bench_final.json | cut -c1-400
MHAP_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$R/${R}_bench_forcedist_rccl_1rank.json
timeout 600 python bench.py --config c1 --steps 200 --warmup 20 > gpurun_out/$R/${R}_bench_c1.json 2>/dev/null   # (a 1.3-ms step: two steps are one hiccup away from any number)
for c in c4slice c5slice; do timeout 600 python bench.py --config $c > gpurun_out/$R/${R}_bench_$c.json 2>/dev/null; done
# configs[3] in full and one rank's share of configs[4] on one GPU, one rank of an N-GPU job (gathered rows of the other ranks in HBM), the native driver end to end
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 > gpurun_out/$R/${R}_bench_c4.json 2>/dev/null
timeout 1200 python bench.py --config c5rank --steps 2 --warmup 1 > gpurun_out/$R/${R}_bench_c5rank.json 2>/dev/null
# (medians over the iterations after the first; the last C2 line: the eager add's kernel order — the ordered kernel in front of the MinHash launch, what a real N-GPU run does for its exchange)
(for n in 2 4 8; do python tools/emulate_rank.py $n c2 10 2>/dev/null | tail -1; done; python tools/emulate_rank.py 8 c4 4 2>/dev/null | tail -1; python tools/emulate_rank.py 8 c5 1 count 2>/dev/null | tail -1
 echo "# N = 8 at C2 with MHAP_ORDERED_FIRST=1 (the eager add's kernel order)"; MHAP_ORDERED_FIRST=1 python tools/emulate_rank.py 8 c2 10 2>/dev/null | tail -1) > gpurun_out/$R/${R}_emulate_rank.txt
# round 6: the MinHash launch's clock by what runs in front of it (final source; alternating): rounds 1-5's order, the shipped split, all of the ordered kernel first
(for rep in 1 2 3; do for v in "MHAP_ORDERED_SPLIT=0" "MHAP_ORDERED_SPLIT=55" "MHAP_ORDERED_SPLIT=100"; do
   echo "== $v"; env $v MHAP_MINHASH_PROF=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof\] launch" | tail -3
   env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['kernel_ms_per_step'])"
 done; done) > gpurun_out/$R/${R}_minhash_clock_order_final.txt 2>&1
(echo "# tools/w1_clock_probe.py 8 c2 on the final source: [add wall ms, MinHash kernel ms] after 100 ms of idle / behind one add / behind three adds"; timeout 600 python tools/w1_clock_probe.py 8 c2 2>/dev/null | tail -1) > gpurun_out/$R/${R}_w1_clock_probe_final.txt
(bash tools/e2e_probe.sh c2; bash tools/e2e_probe.sh c4) > gpurun_out/$R/${R}_e2e_probe.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o p --output-format csv -- python bench.py --no-cpu-baseline --steps 1 --warmup 1 --config c5rank > /dev/null 2>&1
cp $(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1) gpurun_out/$R/${R}_rocprofv3_kernel_stats_c5rank.csv
# the planted family at 1 % and at 5 % divergence, one rank's size, first 40 000 queries (what the c5rank configuration was chosen from)
(python tools/c5_probe.py 625000 40000 0.01 2>/dev/null | tail -1; python tools/c5_probe.py 625000 40000 0.05 2>/dev/null | tail -1; python tools/c5_probe.py 160000 160000 0.01 2>/dev/null | tail -1) > gpurun_out/$R/${R}_c5_probe.txt
# where the pairs of the second stage end (diagnostic build of the join kernel: bash tools/build_variant.sh ojstats -DMH_OJ_STATS)
V=mhap_amd/lib/variants
if [ -f $V/libmhaphip_ojstats.so ]; then
  (MHAP_LIB_PATH=$V/libmhaphip_ojstats.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "oj stats" | tail -3 | sed 's/^/c2: /'
   MHAP_LIB_PATH=$V/libmhaphip_ojstats.so python bench.py --config c5slice --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "oj stats" | tail -3 | sed 's/^/c5slice: /') > gpurun_out/$R/${R}_join_exit_stats.txt
fi
# what the parts of the join kernel cost (timing builds, results wrong by construction): rows streamed only / + filter, lookups, groups collected /
# everything but the duplicated-hash groups / the shipped kernel
#   bash tools/build_variant.sh oj_stream -DMH_OJ_NO_SEARCH -DMH_OJ_JOIN_ONLY; bash tools/build_variant.sh oj_join -DMH_OJ_JOIN_ONLY; bash tools/build_variant.sh oj_nogroups -DMH_OJ_NO_GROUPS
if [ -f $V/libmhaphip_oj_stream.so ]; then
  (for c in c2 c5slice; do for v in $V/libmhaphip_oj_stream.so $V/libmhaphip_oj_join.so $V/libmhaphip_oj_nogroups.so mhap_amd/lib/libmhaphip.so; do
     MHAP_LIB_PATH=$v timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$c', '$v'.split('/')[-1], 'overlap kernel ms', d['kernel_ms_per_step'].get('overlap'))"
   done; done) > gpurun_out/$R/${R}_join_timing_builds.txt
fi
# instruction mix of the join kernel (PMC pass of its own)
(bash tools/pmc_kernel.sh overlap_join_kernel; CONFIG=c5slice bash tools/pmc_kernel.sh overlap_join_kernel) > gpurun_out/$R/${R}_pmc_join_instmix.txt 2>&1
# round 6: the first query tier's floor (random 64-byte lines) and what its parts cost (timing builds: bash tools/build_variant.sh iqt1 -DMH_IQ_TIMING=1 --only search_kernels.hip ...)
[ -x tools/bin/line_gather_probe ] && timeout 300 tools/bin/line_gather_probe > gpurun_out/$R/${R}_line_gather_probe.txt 2>&1
if [ -f $V/libmhaphip_iqt1.so ]; then
  (for v in mhap_amd/lib/libmhaphip.so $V/libmhaphip_iqt1.so $V/libmhaphip_iqt2.so $V/libmhaphip_iqt7.so; do echo "== $(basename $v)"; MHAP_LIB_PATH=$v timeout 300 python tools/emulate_rank.py 8 c2 4 2>/dev/null | tail -1; done) > gpurun_out/$R/${R}_iq_timing_builds.txt
fi
MHAP_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 1 --exchange-only --steps 3 2>/dev/null | tail -1 > gpurun_out/$R/${R}_exchange_only_1rank.json
timeout 400 python tools/check_elements.py c5slice 2>/dev/null | tail -1 > gpurun_out/$R/${R}_check_elements_c5slice.txt
ls -la gpurun_out/$R
