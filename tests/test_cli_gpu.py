"""The native driver (mhap_amd/lib/mhap-hip) keeps MHAP's command line, record format and `.dat` files."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import mhap_amd
from mhap_amd import FastaData, MhapParams, MinHashSearch
from mhap_amd import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "mhap_amd", "lib", "mhap-hip")
GFLAGS = ["-k", "16", "--num-hashes", "64", "--ordered-kmer-size", "12", "--ordered-sketch-size", "256", "--min-olap-length", "116"]


def _run(args):
    r = subprocess.run([CLI] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return sorted(l for l in r.stdout.split("\n") if l), r.stderr


def test_cli_self_overlap_matches_golden():
    g = json.load(open(os.path.join(GOLD, "small_reads.json")))
    lines, err = _run(["-s", os.path.join(GOLD, "small_reads.fasta")] + GFLAGS)
    assert lines == g["sorted_records"]
    assert "Time (s) to read and hash from file:" in err and "Total matches found: %d" % len(lines) in err


def test_cli_matches_the_extended_golden_fixture():
    """The fixture's other cases (what tests/golden/verify_against_jar.sh diffs against a real mhap.jar when a JVM exists): -q with
    reads below --min-olap-length in the query file, -f with a filter file, --supress-noise 1 and 2."""
    g = json.load(open(os.path.join(GOLD, "small_reads.json")))
    s, q, f = (os.path.join(GOLD, x) for x in ("small_reads.fasta", "small_queries.fasta", "small_kmers.txt"))
    lines, err = _run(["-s", s, "-q", q, "--no-self"] + GFLAGS)
    assert lines == g["query_records_no_self"] and len(lines) > 50
    assert "Processed 10 to sequences." in err                    # 12 query reads, two below --min-olap-length
    thr = ["--filter-threshold", repr(g["filter_threshold"])]
    assert _run(["-s", s, "-f", f] + thr + GFLAGS)[0] == g["filter_records"]
    assert _run(["-s", s, "-f", f, "--supress-noise", "1"] + thr + GFLAGS)[0] == g["supress_noise_1_records"]
    assert _run(["-s", s, "-f", f, "--supress-noise", "2"] + thr + GFLAGS)[0] == g["supress_noise_2_records"]
    assert g["filter_records"] != g["sorted_records"] and g["supress_noise_1_records"] != g["filter_records"]


def test_cli_dat_roundtrip_and_format(tmp_path):
    """-p writes MHAP's `.dat` (big-endian framing, Appendix B); -s x.dat reproduces the FASTA run."""
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir(); outdir.mkdir()
    fasta = indir / "small_reads.fasta"
    fasta.write_bytes(open(os.path.join(GOLD, "small_reads.fasta"), "rb").read())
    _run(["-p", str(indir), "-q", str(outdir)] + GFLAGS)
    dat = outdir / "small_reads.dat"
    blob = dat.read_bytes()
    # first record: u8 isFwd, i32 size, payload = u8 isFwd, i64 id, UTF header, i32 seqlen, i32 H, H*i32, i32 n, i32 k2, i32 size, ...
    is_fwd, size = struct.unpack(">bi", blob[:5])
    pay = blob[5:5 + size]
    f2, rid, hl = struct.unpack(">bqH", pay[:11])
    hdr = pay[11:11 + hl].decode()
    seqlen, H = struct.unpack(">ii", pay[11 + hl:19 + hl])
    assert (is_fwd, f2, rid, hdr, H) == (1, 1, 1, "1", 64)
    fa = FastaData.from_file(str(fasta))
    assert seqlen == fa.lengths[0]
    mh = struct.unpack(">64i", pay[19 + hl:19 + hl + 256])
    assert list(mh) == O.minhash(fa.sequence(0), 16, 64)[1].tolist()
    n, k2, osz = struct.unpack(">iii", pay[19 + hl + 256:19 + hl + 268])
    assert (n, k2, osz) == (seqlen - 11, 12, 256)
    g = json.load(open(os.path.join(GOLD, "small_reads.json")))
    lines, _ = _run(["-s", str(dat)] + GFLAGS)
    assert lines == g["sorted_records"]
    # the whole file, byte for byte, against the restatement's writer (tools/jvm/native_dump.py dat: what verify_against_jar.sh
    # compares with `java -jar mhap.jar -p` on a JVM box — GPU sketches == oracle sketches == (there) the JVM's)
    import importlib.util
    spec = importlib.util.spec_from_file_location("native_dump", os.path.join(ROOT, "tools", "jvm", "native_dump.py"))
    nd = importlib.util.module_from_spec(spec); spec.loader.exec_module(nd)
    ref = tmp_path / "native.dat"
    nd.dump_dat(str(fasta), str(ref), 16, 64, 12, 256, 116)
    assert ref.read_bytes() == blob


def test_cli_query_mode_and_full_ids(tmp_path):
    base = mhap_amd.synth_reads(60, 2500, seed=91, error_rate=0.05)

    def write(path, idx, prefix):
        with open(path, "w") as fh:
            for i in idx:
                fh.write(f">{prefix}{i} extra,words\n{base.sequence(i)}\n")
    sfile, qfile = tmp_path / "index.fasta", tmp_path / "query.fa"
    write(sfile, range(0, 40), "s")
    write(qfile, range(30, 60), "q")
    flags = ["--num-hashes", "64", "--ordered-sketch-size", "400"]
    lines, err = _run(["-s", str(sfile), "-q", str(qfile)] + flags)
    index = FastaData.from_file(str(sfile))
    queries = FastaData.from_file(str(qfile), id_offset=40)
    p = MhapParams(num_hashes=64, ordered_sketch_size=400)
    with MinHashSearch(p) as ms:
        ms.add_data(index)
        want = mhap_amd.records_to_lines(ms.find_matches()) + mhap_amd.records_to_lines(ms.find_matches_stream(queries))
    assert lines == sorted(want) and len(lines) > 30
    noself, _ = _run(["-s", str(sfile), "-q", str(qfile), "--no-self"] + flags)
    with MinHashSearch(p) as ms:
        ms.add_data(index)
        assert noself == sorted(mhap_amd.records_to_lines(ms.find_matches_stream(queries)))
    # the same queries as a precomputed .dat (-p), then -q queries.dat: forward entries only, same records
    qdir, ddir = tmp_path / "qfa", tmp_path / "qdat"
    qdir.mkdir(); ddir.mkdir()
    (qdir / "query.fa").write_bytes(qfile.read_bytes())
    _run(["-p", str(qdir), "-q", str(ddir)] + flags)
    viadat, _ = _run(["-s", str(sfile), "-q", str(ddir / "query.dat"), "--no-self"] + flags)
    assert viadat == noself
    full, _ = _run(["-s", str(sfile), "-q", str(qfile), "--store-full-id"] + flags)
    hdr = {i + 1: f"s{i}" for i in range(40)}
    hdr.update({41 + j: f"q{30 + j}" for j in range(30)})
    expect = sorted(" ".join([hdr[int(l.split()[0])], hdr[int(l.split()[1])]] + l.split()[2:]) for l in lines)
    assert full == expect
    # names also come through for compressed input (they are parsed by mhap_fasta_read, not by a second pass over the file)
    import gzip
    sgz, qgz = tmp_path / "index.fasta.gz", tmp_path / "query.fa.gz"
    sgz.write_bytes(gzip.compress(sfile.read_bytes())); qgz.write_bytes(gzip.compress(qfile.read_bytes()))
    fullgz, _ = _run(["-s", str(sgz), "-q", str(qgz), "--store-full-id"] + flags)
    assert fullgz == expect


def test_cli_query_id_offsets_skip_unsketched_reads(tmp_path):
    """seqNumberProcessed advances by the sketches PRODUCED (MhapMain.java:462,537; SequenceSketchStreamer.java:129-133,
    268-271): reads below --min-olap-length get a FASTA id but do not move the id offset of the following -q files."""
    base = mhap_amd.synth_reads(50, 2000, seed=17, error_rate=0.05)
    short = "ACGTTGCA" * 10                                     # 80 bp < 116

    def write(path, items):
        with open(path, "w") as fh:
            for j, s in enumerate(items):
                fh.write(f">x{j}\n{s}\n")
    sfile = tmp_path / "index.fasta"
    qdir = tmp_path / "q"
    qdir.mkdir()
    s_items = [base.sequence(i) for i in range(0, 12)] + [short, short] + [base.sequence(i) for i in range(12, 30)]
    q1 = [base.sequence(20), short, base.sequence(25), base.sequence(31), short, base.sequence(5)]
    q2 = [base.sequence(i) for i in range(28, 45)]
    write(sfile, s_items); write(qdir / "a.fasta", q1); write(qdir / "b.fasta", q2)
    flags = ["--num-hashes", "64", "--ordered-sketch-size", "400"]
    lines, err = _run(["-s", str(sfile), "-q", str(qdir)] + flags)
    index = FastaData.from_file(str(sfile))
    assert len(index) == 32
    off1 = 30                                                   # 30 of the 32 -s reads were sketched
    off2 = off1 + 4                                             # 4 of a.fasta's 6 reads were sketched
    qa = FastaData.from_file(str(qdir / "a.fasta"), id_offset=off1)
    qb = FastaData.from_file(str(qdir / "b.fasta"), id_offset=off2)
    p = MhapParams(num_hashes=64, ordered_sketch_size=400)
    with MinHashSearch(p) as ms:
        ms.add_data(index)
        want = (mhap_amd.records_to_lines(ms.find_matches()) + mhap_amd.records_to_lines(ms.find_matches_stream(qa))
                + mhap_amd.records_to_lines(ms.find_matches_stream(qb)))
    assert lines == sorted(want) and len(lines) > 20
    assert "Processed 4 to sequences." in err and "Processed 17 to sequences." in err
    assert "Processed 60 unique sequences (fwd and rev)." in err


def test_cli_filter_file_and_presets(tmp_path):
    fa = mhap_amd.synth_reads(120, 3000, seed=31, error_rate=0.05)
    fasta = tmp_path / "r.fasta"
    with open(fasta, "w") as fh:
        for i in range(len(fa)):
            fh.write(f">r{i}\n{fa.sequence(i)}\n")
    counts = {}
    for i in range(len(fa)):
        s = fa.sequence(i)
        for j in range(len(s) - 15):
            counts[s[j:j + 16]] = counts.get(s[j:j + 16], 0) + 1
    total = sum(counts.values())
    top = sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))[:300]
    ffile = tmp_path / "kmers.txt"
    with open(ffile, "w") as fh:
        fh.write(f"{len(counts)} {len(top)}\n")
        for kmer, c in top:
            fh.write(f"{kmer}\t{c / total:.10f}\n")
    flags = ["--num-hashes", "128", "--ordered-sketch-size", "512", "--filter-threshold", "1e-5"]
    lines, err = _run(["-s", str(fasta), "-f", str(ffile)] + flags)
    flt = mhap_amd.FrequencyCounts.from_file(str(ffile), filter_cutoff=1e-5, repeat_weight=0.9)
    oflt = O.Filter(flt.hashes, flt.fractions, 1e-5, 0.9, 3.0, False)
    want = O.record_lines(O.run_self(FastaData.from_file(str(fasta)), H=128, S=512, nthreads=8, flt=oflt)["records"])
    assert lines == want and len(lines) > 50
    assert "Read in k-mer filter for sizes: [16]" in err
    for mode in ("1", "2"):                                          # --supress-noise: the file's k-mers as a Bloom-filter whitelist
        sup, _ = _run(["-s", str(fasta), "-f", str(ffile), "--supress-noise", mode] + flags)
        f2 = mhap_amd.FrequencyCounts.from_file(str(ffile), filter_cutoff=1e-5, repeat_weight=0.9, supress_noise=int(mode))
        o2 = O.Filter(f2.hashes, f2.fractions, 1e-5, 0.9, 3.0, False, remove_unique=int(mode), whitelist=f2.whitelist, size_bloom=f2.size_bloom)
        wants = O.record_lines(O.run_self(FastaData.from_file(str(fasta)), H=128, S=512, nthreads=8, flt=o2)["records"])
        assert sup == wants, mode
    fast, _ = _run(["-s", str(fasta), "--settings", "2"])          # fast preset: H 256, thr 0.80, S 1000, k2 14
    wantf = O.record_lines(O.run_self(FastaData.from_file(str(fasta)), H=256, S=1000, k2=14, threshold=0.80, nthreads=8)["records"])
    assert fast == wantf
    bad = subprocess.run([CLI, "-s", str(fasta), "--threshold", "2"], capture_output=True, text=True)
    assert bad.returncode == 1 and "0<=threshold<=1.0" in bad.stdout


def test_cli_filter_files_without_fractions(tmp_path):
    """Legal but unusual -f files (J/sketch/FrequencyCounts.java:150-200): k-mer-only lines (no fraction: they only feed the
    --supress-noise whitelist), a header-only file (the filter is still installed: every k-mer gets idf = --repeat-idf-scale), and a
    line whose fraction does not parse (Double.parseDouble throws: the whole line is dropped, whitelist included).  The native
    parser (mhap_set_filter_file) and the Python mirror (FrequencyCounts.from_file) must agree with the oracle on all of them."""
    fa = mhap_amd.synth_reads(120, 2500, seed=41, error_rate=0.05)
    fasta = tmp_path / "reads.fasta"
    W.write_fasta(fa, str(fasta))
    flags = ["--num-hashes", "128", "--ordered-sketch-size", "512"]
    u, cnt, total = W.count_kmers(fa, 16, True, None)
    order = np.argsort(-cnt)[:300]
    lines = [W.kmer_string(u[i], 16) for i in order]
    reads = FastaData.from_file(str(fasta))

    def check(name, body, mode, expect_bloom):
        f = tmp_path / name
        f.write_text(body)
        got, _ = _run(["-s", str(fasta), "-f", str(f)] + (["--supress-noise", str(mode)] if mode else []) + flags)
        fc = mhap_amd.FrequencyCounts.from_file(str(f), filter_cutoff=1e-5, repeat_weight=0.9, supress_noise=mode)
        assert fc.size_bloom == expect_bloom
        oflt = O.Filter(fc.hashes, fc.fractions, 1e-5, 0.9, 3.0, False, remove_unique=mode, whitelist=fc.whitelist, size_bloom=fc.size_bloom)
        want = O.record_lines(O.run_self(reads, H=128, S=512, nthreads=8, flt=oflt)["records"])
        assert got == want and len(want) > 20, name
        with MinHashSearch(MhapParams(num_hashes=128, ordered_sketch_size=512), kmer_filter=fc) as ms:     # the Python path installs the same filter
            ms.add_data(reads)
            assert sorted(mhap_amd.records_to_lines(ms.find_matches())) == want, name
        return fc

    plain, _ = _run(["-s", str(fasta)] + flags)
    # whitelist only: no line carries a fraction
    fc = check("white.txt", "300 300\n" + "".join(k + "\n" for k in lines), 1, 300)
    assert len(fc.hashes) == 0 and len(fc.whitelist) == 300
    check("white2.txt", "300 300\n" + "".join(k + "\n" for k in lines), 2, 300)
    # header only, sizeBloom 0 (counts as 1): idf = range for every k-mer, so the weights are round(3 tf), not tf
    fc = check("header.txt", "0 0\n", 0, 1)
    assert len(fc.hashes) == 0
    # a malformed fraction drops its line from the table AND from the whitelist
    body = "300 300\n" + "".join(f"{k}\t{'oops' if i % 7 == 3 else '%.6e' % (cnt[order[i]] / total)}\n" for i, k in enumerate(lines))
    fc = check("bad.txt", body, 1, 300)
    assert len(fc.whitelist) == 300 - len([i for i in range(300) if i % 7 == 3])
    assert plain   # (sanity: the unfiltered run has records too)


def test_bench_distributed_path_through_the_library_one_rccl_rank():
    """bench.py's N > 1 code path on the one GPU this box has: torchrun with one rank, RCCL communicator created inside the library
    from the id torch.distributed broadcasts (mhap_dist_init), collective search (mhap_dist_find_matches_self).  Record count and
    record checksum must equal the plain single-GPU run's."""
    import sys
    args = ["--reads", "3000", "--length", "3000", "--steps", "1", "--warmup", "0", "--error-rate", "0.05", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().split("\n")[-1])
    dist = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args,
                          capture_output=True, text=True, timeout=900, env=dict(os.environ, MHAP_BENCH_FORCE_DIST="1"))
    assert dist.returncode == 0, dist.stderr[-3000:]
    r2 = json.loads([l for l in dist.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert r2["records_per_step"] == r1["records_per_step"] and r1["records_per_step"] > 1000
    assert r2["records_checksum"] == r1["records_checksum"] and r1["records_sha256_sorted_lines"]
    assert r2["phase_wall_ms"]["exchange"] >= 0.0
    # the per-rank view the driver can check an N-GPU run against: N ranks, each counting N, distinct devices
    assert r2["ranks"]["consistent"] and r2["ranks"]["world"] == 1 and r2["ranks"]["per_rank"][0]["comm_count"] == 1 and r1["ranks"] is None


def test_bench_preflight_says_what_an_n_gpu_run_needs():
    """bench.py --preflight: one GPU is ready (devices, the library's RCCL entry points, the IPC mode); eight on this one-GPU box are
    not, and the report names the missing devices instead of a collective hanging."""
    import sys
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight", "--gpus", "1"], capture_output=True, text=True, timeout=300)
    r1 = json.loads(one.stdout.strip().split("\n")[-1])
    assert one.returncode == 0 and r1["ready"] and r1["checks"]["library_rccl_entry_points"]["ok"], (one.stdout, one.stderr[-1000:])
    import torch
    if torch.cuda.device_count() < 8:
        many = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight", "--gpus", "8"], capture_output=True, text=True, timeout=300)
        r8 = json.loads(many.stdout.strip().split("\n")[-1])
        assert many.returncode == 1 and not r8["ready"] and not r8["checks"]["devices"]["ok"] and "torch.distributed.run" in r8["launch"]


def test_bench_dry_collective_and_rank_views():
    """bench.py --gpus 1 --dry-collective: the communicator the timed run would form (one rank here), a checked 1 MB and 64 MB all-gather
    through the library's transport, and the per-rank view — RCCL's count / user rank / device and the PCI bus id — in ONE JSON line,
    without any compute step; a world size that does not match the launch is refused before any collective."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-collective"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    d = json.loads(r.stdout.strip().split("\n")[-1])
    assert d["dry_collective"] and d["ready"] and d["n_gpus"] == 1
    v = d["ranks"]
    assert v["consistent"] and v["world"] == 1 and v["per_rank"][0]["comm_count"] == 1 and v["per_rank"][0]["comm_user_rank"] == 0
    assert v["per_rank"][0]["pci_bus_id"] and v["per_rank"][0]["rccl_version"]
    res = d["per_rank_results"][0]
    assert res["error"] is None and len(res["allgather_1MB_ms"]) == 3 and res["allgather_64MB_ms"] > 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-collective"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "torch.distributed.run" in (bad.stdout + bad.stderr)


def test_bench_exchange_only_separates_fabric_time_from_the_time_under_the_add():
    """bench.py --gpus 1 --exchange-only (one RCCL rank here): the real row volumes of a (reduced) C2 job gathered alone and by the eager
    add under its kernels, the exposed part the search waited for, every rank's numbers in one JSON line."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--exchange-only", "--config", "c2", "--reads", "4000", "--steps", "2"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    d = json.loads(r.stdout.strip().split("\n")[-1])
    assert d["exchange_only"] and d["ok"] and d["n_gpus"] == 1
    m = d["per_rank"][0]
    assert m["error"] is None and m["reads"] == 4000 and m["ordered_bytes_sent"] == 4000 * 1536 * 8
    assert len(m["alone"]["ordered_gather_ms"]) == 3 and min(m["alone"]["ordered_gather_ms"]) > 0
    assert len(m["under_the_add"]) == 2
    for u in m["under_the_add"]:
        assert u["ordered_gather_ms"] > 0 and u["small_gather_ms"] > 0 and u["records"] > 0 and u["exposed_ms"] >= 0
    assert m["under_the_add"][-1]["eager_searches"] >= 2          # the searches found their rows gathered by the add


def test_cli_gpus_flag_shards_the_index_and_keeps_the_records(tmp_path):
    """mhap-hip --devices 0,0[,0]: two and three ranks (sharing this box's one GPU) — reads dealt round-robin, one index shard
    per rank, the forward query rows gathered inside the library — print exactly the records of the one-GPU run, in self mode
    and with -q (toSelf = false); --gpus N with a device that does not exist fails loudly."""
    fa = mhap_amd.synth_reads(1200, 2500, seed=31, error_rate=0.06)
    fasta = tmp_path / "reads.fasta"
    W.write_fasta(fa, str(fasta))
    q = mhap_amd.synth_reads(1200, 2500, seed=31, error_rate=0.06, shard=5, nshards=12)     # reads of the same genome as queries
    qf = tmp_path / "queries.fasta"
    W.write_fasta(q, str(qf))
    flags = ["--num-hashes", "128", "--ordered-sketch-size", "512"]
    one, _ = _run(["-s", str(fasta), "-q", str(qf)] + flags)
    assert len(one) > 500
    for devs in ("0,0", "0,0,0"):
        many, err = _run(["-s", str(fasta), "-q", str(qf), "--devices", devs] + flags)
        assert many == one, devs
        assert f"Using {len(devs.split(','))} GPU ranks" in err
    ndev = 1
    try:
        import torch
        ndev = torch.cuda.device_count()
    except Exception:
        pass
    bad = subprocess.run([CLI, "-s", str(fasta), "--gpus", str(ndev + 1)] + flags, capture_output=True, text=True)
    assert bad.returncode != 0 and "device ordinal out of range" in bad.stderr
