run() { timeout 200 python bench.py --reads 20000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', 'minhash_ms',d['kernel_ms_per_step']['minhash'],'weight_ms',d['kernel_ms_per_step']['kmer_weight'],'records',d['records_per_step'])"; }
run default
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
