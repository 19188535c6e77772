run() { python bench.py --reads 20000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', 'minhash_ms',d['kernel_ms_per_step']['minhash'],'records',d['records_per_step'])"; }
MHAP_MINHASH_VARIANT=64 run base_U4
MHAP_MINHASH_VARIANT=65 run xs32_U4
MHAP_MINHASH_VARIANT=71 run xs32_seed_single_U4
MHAP_MINHASH_VARIANT=135 run xs32_seed_single_U8
MHAP_MINHASH_VARIANT=103 run xs32_seed_single_U6
MHAP_MINHASH_VARIANT=39 run xs32_seed_single_U2
MHAP_MINHASH_VARIANT=65 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
