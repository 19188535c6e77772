# final profiles of a round: rocprofv3 kernel stats + PMC traffic on the default workload (run on the GPU box)
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_final gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o prof --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_final/bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$c/bench.log 2>&1
done
timeout 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -1 gpurun_out/bench_final.json | cut -c1-600
