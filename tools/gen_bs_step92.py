#!/usr/bin/env python3
"""bs_step() of mhap_amd/csrc/sketch_kernels.hip with 92 vector operations instead of 107, in place.

x ^= x << 21; x ^= x >> 35; x ^= x << 4 on 64 bit-planes.  With A[b] = x[b] ^ x[b-21] (b >= 21; A[b] = x[b] below) the result is
    C[b] = A[b] ^ A[b-4]                 b = 33..63      (two inputs: ONE FREE SLOT of a three-input xor)
    C[b] = A[b] ^ A[b-4] ^ A[b+31]       b = 29..32
    C[b] = A[b] ^ A[b-4] ^ C[b+35]       b =  4..28
    C[b] = A[b] ^ A[b+35]                b =  0..3       (one free slot)
The 107-operation form (tools/gen_bs_step.py) forms every A[21..63] (43 two-input xors) and then spends one operation per output (64).
But an A[b] whose readers all have a free slot need not exist: its reader takes x[b] and x[b-21] instead.  The readers of A[b] are C[b],
C[b+4] (and C[b-35] for b = 35..38, C[b-31] for b = 60..63), and a reader can absorb one expanded A: along each chain b, b+4, b+8, ...
inside 33..59 every other A is dropped, 4 + 4 + 4 + 3 = 15 of the 43.  28 + 64 = 92 operations, 61 of them three-input.

In place: plane b holds x[b], then A[b] (if formed), then C[b]; a write must follow every other reader of what it overwrites.  That
relation has cycles.  Forming an A[b] in a TEMPORARY instead of its plane removes its plane's constraints at no cost in operations;
SAVED below is a set of five such A's that leaves the relation acyclic (found by `--search`: greedy cycle breaking over samples of the
cycles, preferring A's; a saved x[] would cost a copy, and none is needed).  The script orders the operations, runs them in place on
random values against the 64-bit step, and prints the C++ statements.   (Written as single-assignment values instead, the compiler's own
placement needed ten copies per step: 102 instructions.)"""
import itertools
import random
import sys
import networkx as nx

SKIP_START = (0, 0, 0, 0)                                   # per residue b % 4: which of the chain's alternations is dropped
SAVED = {("a", 32), ("a", 37), ("a", 38), ("a", 53), ("a", 62)}


def skips(start):
    s = set()
    for r, st in zip(range(4), start):
        s.update([b for b in range(33, 60) if b % 4 == r][st::2])
    return s


def build(skip):
    """op name -> (plane it belongs to, operands); operands: ('x', b) a plane's old content, ('a', b), ('c', b)"""
    def A(b):
        if b < 21:
            return [("x", b)]
        return [("x", b), ("x", b - 21)] if b in skip else [("a", b)]
    ops = {("a", b): (b, [("x", b), ("x", b - 21)]) for b in range(21, 64) if b not in skip}
    for b in range(64):
        if b >= 33:
            ts = A(b) + A(b - 4)
        elif b >= 29:
            ts = A(b) + A(b - 4) + A(b + 31)
        elif b >= 4:
            ts = A(b) + A(b - 4) + [("c", b + 35)]
        else:
            ts = A(b) + A(b + 35)
        assert len(ts) <= 3
        ops[("c", b)] = (b, ts)
    return ops


def graph(ops, saved):
    g = nx.DiGraph()
    g.add_nodes_from(ops)
    readers = {}
    for name, (_, ts) in ops.items():
        for t in ts:
            readers.setdefault(t, []).append(name)
            if t[0] in "ac":
                g.add_edge(t, name, content=None)            # producer first
    for b in range(64):
        seq = [("x", b)] + ([("a", b)] if ("a", b) in ops else []) + [("c", b)]
        for prev, w in zip(seq, seq[1:]):
            if prev in saved:
                continue                                     # (lives in a temporary: its plane is not overwritten by w's predecessor)
            for r in readers.get(prev, []):
                if r != w and not g.has_edge(r, w):
                    g.add_edge(r, w, content=prev)           # reader of the old content before the writer
    return g


def search(seeds=12):
    best = None
    for start in itertools.product((0, 1), repeat=4):
        skip = skips(start)
        if len(skip) < 15:
            continue
        for seed in range(seeds):
            rnd = random.Random(seed)
            ops, saved = build(skip), set()
            while True:
                g = graph(ops, saved)
                try:
                    nx.find_cycle(g)
                except nx.NetworkXNoCycle:
                    break
                cnt = {}
                for cy in itertools.islice(nx.simple_cycles(g), 200):
                    for u, v in zip(cy, cy[1:] + cy[:1]):
                        c = g.edges[u, v]["content"]
                        if c is not None:
                            cnt.setdefault(c, set()).add(id(cy))
                saved.add(max(cnt, key=lambda c: len(cnt[c]) * (3.0 if c[0] == "a" else 1.0) * (0.7 + 0.6 * rnd.random())))
            cost = (len(ops) + sum(1 for c in saved if c[0] == "x"), len(saved))
            if best is None or cost < best[0]:
                best = (cost, start, sorted(saved))
                print(best, flush=True)
    return best


def statements():
    ops = build(skips(SKIP_START))
    g = graph(ops, SAVED)
    order = list(nx.lexicographical_topological_sort(g, key=lambda n: (-n[1], n[0])))

    def operand(t):
        return f"t{t[1]}" if t in SAVED else f"P[{t[1]}]"
    lines = []
    for n in order:
        dst, ts = ops[n]
        o = [operand(t) for t in ts]
        if n[0] == "a" and n in SAVED:
            lines.append(f"const uint32_t t{dst} = {o[0]} ^ {o[1]};")
        elif len(o) == 3:
            lines.append(f"P[{dst}] = bs_xor3({o[0]}, {o[1]}, {o[2]});")
        elif o[0] == f"P[{dst}]":
            lines.append(f"P[{dst}] ^= {o[1]};")
        else:
            lines.append(f"P[{dst}] = {o[0]} ^ {o[1]};")
    return lines


def step_ref(x):
    m = (1 << 64) - 1
    x ^= (x << 21) & m
    x ^= x >> 35
    x ^= (x << 4) & m
    return x


def check(lines):
    vals = [random.getrandbits(64) for _ in range(32)]
    P = [sum(((v >> b) & 1) << j for j, v in enumerate(vals)) for b in range(64)]
    env = {"P": P, "bs_xor3": lambda a, b, c: a ^ b ^ c}
    for ln in lines:
        exec(ln.replace("const uint32_t ", "").rstrip(";"), env)
    want = [step_ref(v) for v in vals]
    for b in range(64):
        assert P[b] == sum(((w >> b) & 1) << j for j, w in enumerate(want)), b


if __name__ == "__main__":
    if "--search" in sys.argv:
        search()
        sys.exit(0)
    lines = statements()
    for _ in range(50):
        check(lines)
    n3 = sum("bs_xor3" in ln for ln in lines)
    print(f"  // {len(lines)} operations ({n3} three-input), in place, five temporaries and no copy: tools/gen_bs_step92.py")
    for ln in lines:
        print("  " + ln)
