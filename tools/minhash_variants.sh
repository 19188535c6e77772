run() { python bench.py --reads 20000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', 'minhash_ms',d['kernel_ms_per_step']['minhash'],'records',d['records_per_step'])"; }
MHAP_MINHASH_VARIANT=256 run base_U4
MHAP_MINHASH_VARIANT=296 MHAP_BS_SEED=2048 MHAP_BS_MINREM=512 run bsq_seed2048
MHAP_MINHASH_VARIANT=296 MHAP_BS_SEED=0 MHAP_BS_MINREM=512 run bsq_argmin_minrem512
MHAP_MINHASH_VARIANT=296 MHAP_BS_SEED=0 MHAP_BS_MINREM=256 run bsq_argmin_minrem256
MHAP_MINHASH_VARIANT=296 MHAP_BS_SEED=0 MHAP_BS_MINREM=1024 run bsq_argmin_minrem1024
MHAP_MINHASH_VARIANT=296 MHAP_BS_SEED=0 MHAP_BS_MINREM=256 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
