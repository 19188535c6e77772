# A/B of library builds on a bench configuration, alternating runs on one box: CONFIG=c2 bash tools/ab_kernels.sh [rounds] lib1.so lib2.so ...
# prints every kernel's ms per step, the step and the record checksum of every run
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    MHAP_LIB_PATH=$v timeout 600 python bench.py --config ${CONFIG:-c2} --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$v'.split('/')[-1], ' '.join('%s %.2f' % (n, x) for n, x in k.items() if x > 0), 'step %.2f' % d['ms_per_step'], 'records', d['records_per_step'], d.get('records_checksum', '')[:16])"
  done
done
