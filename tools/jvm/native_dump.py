"""The native side of tools/jvm/ScoreTableDump.java: the same three dumps from this repository's restatement (the CPU oracle, whose
arithmetic the library shares: identity table with glibc's log/exp, the hand-written Java "%.6f", Guava's BloomFilter sizing and
MURMUR128_MITZ_64 probes).  tests/golden/verify_against_jar.sh diffs the two line by line wherever a JVM exists.
  python tools/jvm/native_dump.py score 12 1536 | fmt6 | bloom
  python tools/jvm/native_dump.py dat reads.fasta out.dat [k=16 H=64 k2=12 S=256 min_olap=116]
The `dat` mode is a second parity channel that needs no record formatting: the MinHash + ordered sketches of every strand of a FASTA
file in MHAP's own binary sketch format (J/impl/SequenceSketchStreamer.java:322-395, J/impl/SequenceSketch.java:123-148: big-endian,
per entry u8 isFwd, i32 size, then u8 isFwd, i64 id, UTF header, i32 length, i32 H, H x i32, i32 ordered length, i32 k2, i32 size,
size x (i32 hash, i32 pos)) — what `java -jar mhap.jar -p <dir> -q <outdir>` writes for the same file, byte for byte, if the restatement
is right (tests/golden/verify_against_jar.sh compares them with cmp; `mhap-hip -p` must write the same bytes too: tests/test_cli_gpu.py)."""
import ctypes as C
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402

M = (1 << 64) - 1


def lcg(x):
    return (x * 6364136223846793005 + 1442695040888963407) & M


def signed(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def bits(v):
    return "%x" % struct.unpack("<Q", struct.pack("<d", v))[0]


def read_fasta(path):
    """FastaData.enqueueNextSequenceInFile (J/impl/FastaData.java:125-204): lines concatenated, upper-cased, empty records skipped."""
    seqs, cur = [], None
    with open(path, "rb") as fh:
        for ln in fh.read().split(b"\n"):
            ln = ln.rstrip(b"\r")
            if ln.startswith(b">"):
                if cur:
                    seqs.append(cur)
                cur = ""
            elif cur is not None:
                cur += ln.decode("latin-1")
    if cur:
        seqs.append(cur)
    return [q.upper() for q in seqs if q]


def dump_dat(fasta, outpath, k=16, H=64, k2=12, S=256, min_olap=116):
    out = bytearray()
    for i, seq in enumerate(read_fasta(fasta)):
        rid = i + 1                                              # ids: running count of non-empty records, 1-based (FastaData.java:180-181)
        if len(seq) < min_olap:                                  # SequenceSketchStreamer.java:129-133
            continue
        both = []
        for fwd, sq in ((1, seq), (0, O.rc(seq))):
            rc1, mh = O.minhash(sq, k, H)
            rc2, od, olen = O.ordered(sq, k2, S)
            if rc1 != 0 or rc2 != 0:                             # ZeroNGramsFoundException: the read is dropped entirely (:235-238)
                both = []
                break
            both.append((fwd, mh, od, olen))
        for fwd, mh, od, olen in both:
            hdr = str(rid).encode()
            pay = struct.pack(">bqH", fwd, rid, len(hdr)) + hdr + struct.pack(">ii", len(seq), H) + struct.pack(">%di" % H, *[int(v) for v in mh[:H]])
            pay += struct.pack(">iii", olen, k2, len(od)) + b"".join(struct.pack(">ii", int(h), int(p)) for h, p in od)
            out += struct.pack(">bi", fwd, len(pay)) + pay
    with open(outpath, "wb") as fh:
        fh.write(bytes(out))
    return len(out)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "score"
    out = sys.stdout
    if what == "dat":
        a = sys.argv[2:]
        kw = dict(zip(("k", "H", "k2", "S", "min_olap"), (int(x) for x in a[2:])))
        print(dump_dat(a[0], a[1], **kw), "bytes")
        return
    if what == "score":
        k2, S = int(sys.argv[2]), int(sys.argv[3])
        f = O.lib().orc_jaccard_to_identity
        f.restype = C.c_double
        for k in range(S + 1):
            for inter in range(k + 1):
                j = 0.0 if k == 0 else inter / k
                out.write(bits(f(C.c_double(j), C.c_int(k2))) + "\n")
    elif what == "fmt6":
        x = 0x4D484150
        for i in range(20000):
            x = lcg(x)
            if i % 4 == 0:
                v = (x >> 11) / 9007199254740992.0
            elif i % 4 == 1:
                v = ((x >> 40) % 2000000) / 1.0e6 + 5.0e-7
            elif i % 4 == 2:
                v = float((x >> 44) % 3000)
            else:
                v = 1.0 - (x >> 11) / 9007199254740992.0 * 0.3
            out.write(bits(v) + " " + O.java_fmt6(v) + "\n")
    else:
        for n in (1, 2, 10, 1000, 65536, 1000000, 123456789):
            b, k = O.bloom_params(n)
            out.write(f"size {n} {b} {k}\n")
        x, vals = 7, []
        for _ in range(1000):
            x = lcg(x)
            vals.append(signed(x))
        vals = np.array(vals, dtype=np.int64)
        flt = O.Filter(vals, np.full(len(vals), 1e-3), 1e-5, 0.0, 3.0, False, remove_unique=1, whitelist=vals, size_bloom=1000)
        y = 7
        for i in range(64):
            y = lcg(y)
            probe = y if i % 2 == 0 else y ^ 0x5555555555555555
            out.write(f"probe {signed(probe)} {1 if flt.might_contain(signed(probe)) else 0}\n")


if __name__ == "__main__":
    main()
