#!/bin/bash
# PMC passes on the C5 slice: what do index_query / overlap_join wait for?
export TMPDIR=/tmp
mkdir -p gpurun_out/r3j; rm -rf gpurun_out/r3j/*
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*" | sort -u > gpurun_out/r3j/counters.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --config c5slice > gpurun_out/r3j/bench_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" $i <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
with open(f'gpurun_out/r3j/pmc_{sys.argv[2]}.txt','w') as o:
    for k,v in acc.items():
        if 'index_query' in k or 'overlap' in k or 'minhash' in k:
            o.write(k+'\n')
            for c,x in sorted(v.items()): o.write(f'   {c} {x:.4g}\n')
PY
done
