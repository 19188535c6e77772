import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import mhap_amd, numpy as np
fa = mhap_amd.synth_reads(1, 3000, seed=2026, error_rate=0.05)
p = mhap_amd.MhapParams(num_hashes=128, ordered_sketch_size=512, device=0)
with mhap_amd.MinHashSearch(p) as ms:
    sk = ms.sketch(fa)
print("status", sk["status"])
import oracle_lib as O
for i in range(1):
    rc1, mh = O.minhash(fa.sequence(i), 16, 128)
    print(i, (sk["minhash"][2*i]==mh).all())
    rc2, od, _ = O.ordered(fa.sequence(i), 12, 512)
    print(i, (sk["ordered"][2*i,:od.shape[0]]==od).all())
