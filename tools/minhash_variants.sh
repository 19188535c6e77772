for v in 64 65 66 68 70 71 128 134 135 39 102 103; do
  MHAP_MINHASH_VARIANT=$v python bench.py --reads 20000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('variant',$v,'U',$v//16,'VAR',$v%16,'minhash_ms',d['kernel_ms_per_step']['minhash'],'records',d['records_per_step'],'step_ms',d['ms_per_step'])"
done
MHAP_MINHASH_VARIANT=71 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
MHAP_MINHASH_VARIANT=135 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
