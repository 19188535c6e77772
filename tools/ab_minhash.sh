# A/B of library builds on the C2 workload, alternating runs on one box: bash tools/ab_minhash.sh [rounds] lib1.so lib2.so ...
# prints the MinHash kernel's ms per step, the step, and the record checksum of every run
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    MHAP_LIB_PATH=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$v'.split('/')[-1], 'minhash %.2f' % k.get('minhash', 0), 'step %.2f' % d['ms_per_step'], 'records', d['records_per_step'], d.get('records_sha256_sorted_lines', '')[:16])"
  done
done
