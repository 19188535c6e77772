"""Build libmhaphip.so (gfx950) and the mhap-hip CLI in-tree with hipcc.

No CMake: four translation units, one hipcc command.  The built artefacts live under
mhap_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmhaphip.so")
CLI = os.path.join(LIBDIR, "mhap-hip")
SOURCES = ["sketch_kernels.hip", "search_kernels.hip", "mhap_capi.hip", "host_util.cpp"]
HEADERS = ["device_common.hpp", "kernels.hpp", "overlap_lane.hpp", os.path.join("..", "..", "include", "mhap_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 kernels)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    if force or _stale(LIB, deps):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
               "-Wno-pass-failed", *srcs, "-o", LIB, "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
    cli_src = os.path.join(CSRC, "mhap_cli.cpp")
    if os.path.exists(cli_src) and (force or _stale(CLI, [cli_src, LIB])):
        cmd = [hipcc, "-O2", "-std=c++17", "-pthread", cli_src, "-o", CLI, f"-L{LIBDIR}", "-lmhaphip",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
