"""Basic-block instruction budget of one kernel from the compiler's ISA listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -mllvm -amdgpu-sched-strategy=max-memory-clause -S --cuda-device-only \
        -o /tmp/sketch.s mhap_amd/csrc/sketch_kernels.hip
  python tools/isa_budget.py /tmp/sketch.s minhash_w1_kernelILb0 [--blocks] [--scratch]

Prints, per basic block: wave instructions by kind (VALU / v_bitop3 among them / SALU / LDS / VMEM / scratch / branch), whether the block
ends in a backward branch (a loop latch) and to where, and the loop nest it belongs to (innermost latch range).  --scratch lists every
scratch_load / scratch_store with its block.  The counts are static: multiply by the executions of the region (rows, steps, triggers,
drain trips — MHAP_MINHASH_PROF prints them) to get the dynamic budget DESIGN.md quotes against SQ_INSTS_VALU."""
import re
import sys


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        if op in ("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_code_end"):
            return "wait"
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    want_blocks, want_scratch = "--blocks" in sys.argv, "--scratch" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "line": start, "ins": []}
    for i in range(start + 1, end):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "line": i, "ins": []}
            continue
        if not l or l.startswith((";", ".", "//")):
            continue
        op = l.split()[0]
        cur["ins"].append((op, l, i))
    blocks.append(cur)
    index = {b["label"]: n for n, b in enumerate(blocks)}
    tot = {}
    rows = []
    for n, b in enumerate(blocks):
        c = {}
        for op, l, _ in b["ins"]:
            k = kind(op)
            c[k] = c.get(k, 0) + 1
            if op == "v_bitop3_b32":
                c["bitop3"] = c.get("bitop3", 0) + 1
            if k == "valu" and re.search(r"\bs\d+|\bs\[\d+|vcc|exec", l.split(None, 1)[1] if " " in l else "") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_writelane")):
                c["valu_sgpr_src"] = c.get("valu_sgpr_src", 0) + 1
        back = None
        for op, l, _ in b["ins"]:
            if kind(op) == "branch":
                t = l.split()[-1]
                if t in index and index[t] <= n:
                    back = t
        b["counts"], b["back"] = c, back
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
    # loops: latch block n with back edge to header h covers blocks h..n
    loops = sorted(((index[b["back"]], n) for n, b in enumerate(blocks) if b["back"]), key=lambda t: (t[1] - t[0]))
    print("kernel %s: %d blocks, static instructions %s" % (name, len(blocks), tot))
    print("loops (header..latch, static VALU / SALU / LDS / VMEM / scratch inside):")
    for h, n in sorted(loops):
        s = {}
        for b in blocks[h:n + 1]:
            for k, v in b["counts"].items():
                s[k] = s.get(k, 0) + v
        inner = [(a, z) for a, z in loops if a >= h and z <= n and (a, z) != (h, n)]
        print("  %-10s .. %-10s (lines %d-%d) blocks %3d  valu %5d (bitop3 %4d, sgpr-src %3d) salu %4d lds %3d vmem %3d scratch %3d branch %3d  nested loops %d" % (
            blocks[h]["label"], blocks[n]["label"], blocks[h]["line"] + 1, blocks[n]["line"] + 1, n - h + 1, s.get("valu", 0), s.get("bitop3", 0), s.get("valu_sgpr_src", 0),
            s.get("salu", 0), s.get("lds", 0), s.get("vmem", 0), s.get("scratch", 0), s.get("branch", 0), len(inner)))
    if want_blocks:
        for n, b in enumerate(blocks):
            c = b["counts"]
            print("%-10s line %6d  valu %4d bitop3 %3d salu %3d lds %2d vmem %2d scratch %2d br %d %s" % (
                b["label"], b["line"] + 1, c.get("valu", 0), c.get("bitop3", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0), c.get("scratch", 0),
                c.get("branch", 0), ("<- latch of " + b["back"]) if b["back"] else ""))
    if want_scratch:
        for b in blocks:
            for op, l, i in b["ins"]:
                if kind(op) == "scratch":
                    print("%-10s line %6d  %s" % (b["label"], i + 1, l))


if __name__ == "__main__":
    main()
