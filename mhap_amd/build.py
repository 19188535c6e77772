"""Build libmhaphip.so (gfx950) and the mhap-hip CLI in-tree with hipcc.

No CMake: six translation units, each compiled to an object of its own (side by side) and linked with one hipcc command.  The built
artefacts live under mhap_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).  An object is rebuilt whenever the SHA-256
of its source + the headers + its command differs from the stamp written next to it, the library whenever an object changed.
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
SOURCES = ["sketch_kernels.hip", "search_kernels.hip", "search_kernels_wide.hip", "search_kernels_wide2.hip", "mhap_capi.hip", "mhap_dist.hip", "mhap_ingest.hip", "host_util.cpp"]
HEADERS = ["device_common.hpp", "kernels.hpp", "mhap_internal.hpp", "overlap_lane.hpp", os.path.join("..", "..", "include", "mhap_hip.h")]
ARCH = "gfx950"
EXTRA_DEPS = {"search_kernels_wide.hip": ["search_kernels.hip"], "search_kernels_wide2.hip": ["search_kernels.hip"]}   # (a translation unit that includes another one)


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


def stale_objects():
    """Translation units whose shipped object was built from other bytes than the sources here (no compiler needed: the digest of
    (source, headers, command) is recomputed with the recorded compiler path left out of the comparison where it cannot be known)."""
    out = []
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    for s in SOURCES:
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        cmd = [f"--offload-arch={ARCH}", *[f for f in CFLAGS if f != "-shared"], "-c", os.path.join(CSRC, s), "-o", obj]
        dg = _digest([os.path.join(CSRC, s)] + [os.path.join(CSRC, d) for d in EXTRA_DEPS.get(s, [])] + hdrs, cmd)
        if _stale(obj, dg):
            out.append(s)
    return out


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


OBJDIR = os.path.join(LIBDIR, "obj")
# variant builds that tests need next to the shipped library: NAME -> (extra flags, the translation units they change)
VARIANTS = {
    # the weight-1 MinHash kernel with a 31-entry candidate queue per wave: every row overflows it, so the exact redo path
    # (sketch_kernels.hip, `if (!ok)` in minhash_w1_kernel) is EXECUTED against the oracle (tests/test_gpu_parity.py)
    "qcap64": (["-DMH_QCAP=64"], ["sketch_kernels.hip"]),
}


def variant_path(name):
    return os.path.join(LIBDIR, "variants", f"libmhaphip_{name}.so")


def _compile_objects(hipcc, flags, only, tag, force, verbose):
    """One object per translation unit (compiled side by side), reused while the digest of (source, headers, command) holds."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    # a translation unit that textually includes a listed one is a variant too (-DMH_OJ_GCAP on search_kernels.hip alone left
    # the two wider passes of the join kernel at the default and the A/B silently mixed configurations)
    only = list(only) + [s for s, deps in EXTRA_DEPS.items() if s not in only and any(d in only for d in deps)]
    jobs, objs = [], []
    for s in SOURCES:
        variant = bool(flags) and s in only
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + (f".{tag}" if variant else "") + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", *[f for f in CFLAGS if f != "-shared"], *(flags if variant else []), "-c", os.path.join(CSRC, s), "-o", obj]
        dg = _digest([os.path.join(CSRC, s)] + [os.path.join(CSRC, d) for d in EXTRA_DEPS.get(s, [])] + hdrs, cmd[1:])
        objs.append(obj)
        if force or _stale(obj, dg):
            jobs.append((cmd, obj, dg))

    def run(job):
        cmd, obj, dg = job
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        _stamp(obj, dg)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1, 6))) as ex:
        list(ex.map(run, jobs))
    return objs, bool(jobs)


def _link(hipcc, objs, target, force, verbose):
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-pthread", *objs, "-o", target, "-lz", "-ldl"]
    h = hashlib.sha256(" ".join(cmd[1:]).encode())
    for o in objs:
        with open(o + ".sha256") as fh:
            h.update(fh.read().encode())
    dg = h.hexdigest()
    if force or _stale(target, dg):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        os.makedirs(os.path.dirname(target), exist_ok=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        _stamp(target, dg)
    return dg


def build(force=False, verbose=False, variants=()):
    """libmhaphip.so + mhap-hip; `variants`: names from VARIANTS to build as mhap_amd/lib/variants/libmhaphip_NAME.so as well."""
    os.makedirs(LIBDIR, exist_ok=True)
    try:
        hipcc = _hipcc()
    except RuntimeError:
        if os.path.exists(LIB) and not force:   # a box without a compiler: the shipped artefact is all there is
            print(f"note: no hipcc here; using the shipped {LIB} (mhap_amd.load_library() checks its ABI version and struct sizes)", file=sys.stderr)
            st = stale_objects()
            if st:
                print(f"warning: the shipped library is STALE against the sources here: {', '.join(st)} changed since their objects were built", file=sys.stderr)
            return LIB
        raise
    objs, _ = _compile_objects(hipcc, [], [], "", force, verbose)
    dg = _link(hipcc, objs, LIB, force, verbose)
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
    for name in variants:
        flags, only = VARIANTS[name]
        vobjs, _ = _compile_objects(hipcc, flags, only, name, force, verbose)
        _link(hipcc, vobjs, variant_path(name), force, verbose)
    return LIB


def build_variant(name, flags, only=None, verbose=False):
    """An ad-hoc variant (tools/build_variant.sh): extra -D flags on the given translation units (default: the two kernel files)."""
    hipcc = _hipcc()
    vobjs, _ = _compile_objects(hipcc, list(flags), only or ["sketch_kernels.hip", "search_kernels.hip"], name, False, verbose)
    _link(hipcc, vobjs, variant_path(name), False, verbose)
    return variant_path(name)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # python -m mhap_amd.build --variant NAME -DX=1 ... [--only file.hip,...]
        rest = sys.argv[3:]
        only = None
        if "--only" in rest:
            i = rest.index("--only")
            only = rest[i + 1].split(",")
            rest = rest[:i] + rest[i + 2:]
        print(build_variant(sys.argv[2], rest, only, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, variants=list(VARIANTS) if "--variants" in sys.argv else ()))
