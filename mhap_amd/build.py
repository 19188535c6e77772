"""Build libmhaphip.so (gfx950) and the mhap-hip CLI in-tree with hipcc.

No CMake: four translation units, one hipcc command.  The built artefacts live under
mhap_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).  An artefact is rebuilt whenever the SHA-256 of its
sources + build command differs from the stamp written next to it.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmhaphip.so")
CLI = os.path.join(LIBDIR, "mhap-hip")
SOURCES = ["sketch_kernels.hip", "search_kernels.hip", "mhap_capi.hip", "mhap_dist.hip", "mhap_ingest.hip", "host_util.cpp"]
HEADERS = ["device_common.hpp", "kernels.hpp", "mhap_internal.hpp", "overlap_lane.hpp", os.path.join("..", "..", "include", "mhap_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 kernels)")


def _digest(deps, cmd):
    """SHA-256 over the build command and the bytes of every source/header: an artefact is reused only when the stamp
    written next to it matches (mtimes do not survive a snapshot copy to the GPU box and say nothing about content)."""
    h = hashlib.sha256(" ".join(cmd).encode())
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def source_digest():
    """SHA-256 over the kernel / library sources and headers (no build command): what a profile of the kernels is stamped with
    (tools/pmc_summary.py) and what bench.py compares before it quotes that profile's byte counts."""
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in HEADERS]
    return _digest(deps, ["sources"])


def _stale(target, digest):
    stamp = target + ".sha256"
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != digest


# -amdgpu-sched-strategy=max-memory-clause: the machine scheduler variant that keeps memory instructions clustered.  A/B on the
# whole library at C2 (tools/minhash_ab.sh-style alternating runs on one box): MinHash 88.8 -> 86.9 ms, join 4.84 -> 4.71, step 111.2 ->
# 109.1; max-ilp, the AMDGPU register-pressure trackers, an occupancy-only metric bias and no high-RP reschedule stage: +-0.5 ms.
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-pass-failed", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]


def _stamp(target, digest):
    with open(target + ".sha256", "w") as fh:
        fh.write(digest + "\n")


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    try:
        hipcc = _hipcc()
    except RuntimeError:
        if os.path.exists(LIB) and not force:   # a box without a compiler: the shipped artefact is all there is — but say so if it is stale
            cmd = ["hipcc", f"--offload-arch={ARCH}", *CFLAGS, *srcs, "-o", LIB, "-lz", "-ldl"]
            if _stale(LIB, _digest(deps, cmd[1:])):
                print(f"warning: {LIB} does not match the sources next to it (no hipcc here to rebuild it); "
                      "mhap_amd.load_library() checks its ABI version and struct sizes", file=sys.stderr)
            return LIB
        raise
    cmd = [hipcc, f"--offload-arch={ARCH}", *CFLAGS, *srcs, "-o", LIB, "-lz", "-ldl"]
    dg = _digest(deps, cmd[1:])
    if force or _stale(LIB, dg):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        _stamp(LIB, dg)
    cli_src = os.path.join(CSRC, "mhap_cli.cpp")
    if os.path.exists(cli_src):
        cmd = [hipcc, "-O2", "-std=c++17", "-pthread", cli_src, "-o", CLI, f"-L{LIBDIR}", "-lmhaphip",
               "-Wl,-rpath,$ORIGIN"]
        dg_cli = _digest([cli_src, os.path.join(CSRC, HEADERS[-1])], cmd[1:] + [dg])
        if force or _stale(CLI, dg_cli):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True, cwd=CSRC)
            _stamp(CLI, dg_cli)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
