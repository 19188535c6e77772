# A/B of the two MinHash kernels on 20k reads (run on the GPU box): bit-sliced (default) vs per-chain rows only
run() { timeout 200 python bench.py --reads 20000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', 'minhash_ms',d['kernel_ms_per_step']['minhash'],'records',d['records_per_step'])"; }
run bitsliced
MHAP_MINHASH=perchain run perchain
