#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3i; rm -rf gpurun_out/r3i/*
for cfg in c5slice c2; do
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o p --output-format csv -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --config $cfg > gpurun_out/r3i/bench_$cfg.json 2> gpurun_out/r3i/bench_$cfg.err
f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r3i/kernel_stats_$cfg.csv
done
