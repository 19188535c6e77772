#!/usr/bin/env python3
"""Generator of bs_step() in mhap_amd/csrc/sketch_kernels.hip: an evaluation order of the bit-sliced xorshift64 step that works
IN PLACE with a single saved plane.

With A = x ^ (x << 21) (in place, planes 63..21 descending) the step's result is
    C[b] = A[b] ^ A[b-4]                 b = 33..63
    C[b] = A[b] ^ A[b-4] ^ A[b+31]       b = 29..32
    C[b] = A[b] ^ A[b-4] ^ C[b+35]       b =  4..28
    C[b] = A[b] ^ A[b+35]                b =  0..3
Writing C[b] over A[b] needs every other reader of A[b] done first; that relation has one cycle through all 64 planes, and it
is broken by keeping a copy of one plane (any of 29..38).  This script finds the order (topological sort) for the copy of
plane 35, checks it against the 64-bit xorshift on random values, and prints the C++ statements."""
import random
import networkx as nx

SAVED = 35


def ins(b):
    if b >= 33:
        return [("A", b), ("A", b - 4)]
    if b >= 29:
        return [("A", b), ("A", b - 4), ("A", b + 31)]
    if b >= 4:
        return [("A", b), ("A", b - 4), ("C", b + 35)]
    return [("A", b), ("A", b + 35)]


def order():
    g = nx.DiGraph()
    g.add_nodes_from(range(64))
    for b in range(64):
        for t, x in ins(b):
            if t == "C":
                g.add_edge(x, b)
            elif x != b and x != SAVED:
                g.add_edge(b, x)
    return list(nx.lexicographical_topological_sort(g, key=lambda b: -b))


def step_ref(x):
    m = (1 << 64) - 1
    x ^= (x << 21) & m
    x ^= x >> 35
    x ^= (x << 4) & m
    return x


def simulate(seq):
    vals = [random.getrandbits(64) for _ in range(32)]
    P = [sum(((v >> b) & 1) << j for j, v in enumerate(vals)) for b in range(64)]
    for b in range(63, 20, -1):
        P[b] ^= P[b - 21]
    T = P[SAVED]
    done = set()
    for b in seq:
        acc = 0
        for t, x in ins(b):
            if t == "C":
                assert x in done
                acc ^= P[x]
            else:
                assert x == SAVED or x not in done or x == b
                acc ^= T if (x == SAVED and x != b and SAVED in done) else P[x]
        P[b] = acc
        done.add(b)
    got = [sum(((P[b] >> j) & 1) << b for b in range(64)) for j in range(32)]
    assert got == [step_ref(v) for v in vals]


if __name__ == "__main__":
    seq = order()
    for _ in range(20):
        simulate(seq)
    pos = {b: i for i, b in enumerate(seq)}
    print(f"  const uint32_t T = P[{SAVED}];")
    for b in seq:
        terms = []
        for t, x in ins(b):
            if x == b:
                continue
            terms.append("T" if (t == "A" and x == SAVED and pos[SAVED] < pos[b]) else f"P[{x}]")
        if len(terms) == 1:
            print(f"  P[{b}] ^= {terms[0]};")
        else:
            print(f"  P[{b}] = bs_xor3(P[{b}], {terms[0]}, {terms[1]});")
