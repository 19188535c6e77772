#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3n; rm -rf gpurun_out/r3n/*
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee gpurun_out/r3n/pytest.txt
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'])"; }
for cfg in c2 c4slice c1 c5slice; do echo "== $cfg"; run --config $cfg; done | tee gpurun_out/r3n/ab.txt
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python tools/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "pmc_counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "pmc_counter_collection.csv" | head -1) | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d['kernels'].items(): print(k, 'fetch', round(v.get('fetch_bytes_raw',0)/1e9,3), 'write', round(v.get('write_bytes',0)/1e9,3), v['launches'])" | tee gpurun_out/r3n/fetch.txt
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --config c2 2>&1 | grep "w1 prof" | head -1 | tee gpurun_out/r3n/prof.txt
python tools/emulate_rank.py 8 2>&1 | tail -1 | tee -a gpurun_out/r3n/ab.txt
