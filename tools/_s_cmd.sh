L=mhap_amd/lib/ab
CONFIG=c2 STEPS=8 bash tools/ab_kernels.sh 3 $L/t5.so $L/t9.so
CONFIG=c5slice STEPS=3 bash tools/ab_kernels.sh 2 $L/t5.so $L/t9.so
CONFIG=c4slice STEPS=4 bash tools/ab_kernels.sh 1 $L/t5.so $L/t9.so
echo "default: $(timeout 900 python tests/fuzz_parity.py 150 410000 2>/dev/null | tail -1)"
echo "qcap64: $(MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_qcap64.so timeout 900 python tests/fuzz_parity.py 60 420000 2>/dev/null | tail -1)"
echo "classic: $(MHAP_MINHASH=classic timeout 900 python tests/fuzz_parity.py 40 430000 2>/dev/null | tail -1)"
MHAP_MINHASH_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --soak-seconds 0 2>&1 >/dev/null | grep "w1 prof" | tail -2
