#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3n; rm -rf gpurun_out/r3n/*
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r3n/pytest.txt
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'])"; }
for cfg in c2 c4slice c1; do echo "== $cfg"; run --config $cfg; done | tee gpurun_out/r3n/ab.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_summary.py $(find /tmp/pmc_f -name "pmc_counter_collection.csv" | head -1) | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d['kernels'].items(): print(k, round(v.get('fetch_bytes_raw',0)/1e9,3), v['launches'])" | tee gpurun_out/r3n/fetch.txt
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --config c2 2>&1 | grep "w1 prof" | head -2 | tee gpurun_out/r3n/prof.txt
