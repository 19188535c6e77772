import sys, time, ctypes
sys.path.insert(0, "/root/repo")
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
def probe(tag, gb=30):
    p = ctypes.c_void_p(); t = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); dt = time.time() - t
    t = time.time(); hip.hipFree(p); df = time.time() - t
    print(f"{tag}: hipMalloc({gb} GB) {dt*1e3:.1f} ms rc {rc}, hipFree {df*1e3:.1f} ms", flush=True)
probe("torch only")
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch
probe("library imported")
ms = MinHashSearch(MhapParams())
probe("handle created")
fa = mhap_amd.synth_reads(2000, 3000, seed=11, error_rate=0.12)
ms.add_data(fa); ms.synchronize()
probe("after a small add")
r = ms.find_matches()
probe("after a small search")
probe("again", 57)
