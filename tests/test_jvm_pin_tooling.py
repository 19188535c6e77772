"""The JVM-pin tooling, exercised without a JVM.

tests/golden/verify_against_jar.sh is the one command a box with a JVM and mhap.jar has to run to pin the oracle
(parity unpinned).  No such box has been seen, so the script, tools/jvm/native_dump.py and the way the two are
glued had never executed end to end.  Here a stand-in `java` / `javac` on PATH REPLAYS canned outputs — the records of the
committed fixture for each of the five mhap.jar runs (J/impl/MatchResult.java:98-113 lines), the `.dat` file of the `-p`
run (J/impl/SequenceSketchStreamer.java:322-395) and the three ScoreTableDump tables — once faithfully and once with a single
flipped byte per channel.  It pins NOTHING about the reference (the canned outputs come from the restatement itself); it proves
that the plumbing reports `ok` on equal inputs, names the channel and exits non-zero on a one-byte difference."""
import json
import os
import stat
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

STUB_JAVA = r'''#!/usr/bin/env python3
# stand-in for `java`: replays what mhap.jar / ScoreTableDump would print for the golden fixture (tests/test_jvm_pin_tooling.py)
import json, os, subprocess, sys
ROOT, GOLD = os.environ["STUB_ROOT"], os.environ["STUB_GOLD"]
flip = os.environ.get("STUB_FLIP", "")
a = sys.argv[1:]
log = os.environ.get("STUB_LOG")
if log:
    open(log, "a").write(" ".join(a) + "\n")
def native(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "jvm", "native_dump.py"), *args], check=True, capture_output=True).stdout
if "-jar" in a and "-p" in a:                       # mhap.jar -p <dir> -q <outdir>: one .dat per FASTA file of <dir>
    src, dst = a[a.index("-p") + 1], a[a.index("-q") + 1]
    out = os.path.join(dst, "small_reads.dat")
    native("dat", os.path.join(src, "small_reads.fasta"), out, "16", "64", "12", "256", "116")
    if flip == "dat":
        b = bytearray(open(out, "rb").read()); b[len(b) // 2] ^= 1; open(out, "wb").write(bytes(b))
    sys.exit(0)
if "-jar" in a:                                     # the five record runs
    g = json.load(open(os.path.join(GOLD, "small_reads.json")))
    if "--supress-noise" in a:
        key = "supress_noise_%s_records" % a[a.index("--supress-noise") + 1]
    elif "-f" in a:
        key = "filter_records"
    elif "--no-self" in a:
        key = "query_records_no_self"
    else:
        key = "sorted_records"
    lines = list(reversed(g[key]))                  # (mhap's output order is unspecified: the script must sort)
    if flip == key:
        l = lines[len(lines) // 2]; d = "1" if l[-1] != "1" else "2"
        lines[len(lines) // 2] = l[:-1] + d
    sys.stderr.write("Running with these settings: (stub)\n")
    sys.stdout.write("\n".join(lines) + "\n")
    sys.exit(0)
if "ScoreTableDump" in a:                           # java -cp jar:dir ScoreTableDump score 12 1536 | fmt6 | bloom
    what = a[a.index("ScoreTableDump") + 1:]
    out = native(*what)
    if flip == what[0]:
        b = bytearray(out); i = len(b) // 2
        while b[i] in b"\n ": i += 1
        b[i] = ord("0") if b[i] != ord("0") else ord("1")
        out = bytes(b)
    sys.stdout.buffer.write(out)
    sys.exit(0)
sys.stderr.write("stub java: unexpected arguments %r\n" % (a,))
sys.exit(64)
'''

STUB_JAVAC = r'''#!/bin/sh
# stand-in for `javac`: the source must be there and name the class the script runs afterwards
for last in "$@"; do :; done
grep -q "class ScoreTableDump" "$last" || { echo "stub javac: $last does not define ScoreTableDump" >&2; exit 1; }
exit 0
'''


def _stubs(tmp_path, with_javac=True):
    bindir = tmp_path / "bin"
    bindir.mkdir()
    for name, text in (("java", STUB_JAVA),) + ((("javac", STUB_JAVAC),) if with_javac else ()):
        p = bindir / name
        p.write_text(text)
        p.chmod(p.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    return str(bindir)


def _run(tmp_path, bindir, flip="", jar="stub-mhap.jar", path_extra=True):
    env = dict(os.environ)
    env["PATH"] = (bindir + os.pathsep if path_extra else "") + env["PATH"]
    env.update(STUB_ROOT=ROOT, STUB_GOLD=GOLD, STUB_FLIP=flip, STUB_LOG=str(tmp_path / "java_calls.log"), TMPDIR=str(tmp_path))
    if jar is None:
        env.pop("MHAP_JAR", None)
    else:
        env["MHAP_JAR"] = str(tmp_path / jar)
    return subprocess.run(["sh", os.path.join(GOLD, "verify_against_jar.sh")], env=env, capture_output=True, text=True, timeout=600)


def test_verify_script_reports_ok_on_a_faithful_replay(tmp_path):
    r = _run(tmp_path, _stubs(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    g = json.load(open(os.path.join(GOLD, "small_reads.json")))
    for name, key in (("self", "sorted_records"), ("query", "query_records_no_self"), ("filter", "filter_records"),
                      ("sup1", "supress_noise_1_records"), ("sup2", "supress_noise_2_records")):
        assert f"ok       {name} {len(g[key])} records from mhap.jar, {len(g[key])} in the fixture" in out, out
    assert "fixture matches mhap.jar" in out
    assert "ok       sketches: small_reads.dat" in out and "identical to mhap.jar -p" in out
    for name, nlines in (("score", (1536 + 1) * (1536 + 2) // 2), ("fmt6", 20000), ("bloom", 71)):
        assert f"ok       {name}: {nlines} lines identical to the JVM's" in out, out
    assert "JDK / Guava arithmetic matches the restatement" in out
    assert "MISMATCH" not in out
    # the flags every mhap.jar run gets are the fixture's (J/main/MhapMain.java:67-125 names)
    calls = open(tmp_path / "java_calls.log").read().splitlines()
    assert len(calls) == 5 + 1 + 3, calls
    p = g["params"]
    for c in calls[:6]:
        assert f"-k {p['k']} --num-hashes {p['H']} --ordered-kmer-size {p['k2']} --ordered-sketch-size {p['S']} --min-olap-length {p['min_olap_length']} --num-threads 1" in c, c
    assert f"--filter-threshold {g['filter_threshold']}" in calls[2]


@pytest.mark.parametrize("flip,needle", [
    ("sorted_records", "MISMATCH self"),
    ("query_records_no_self", "MISMATCH query"),
    ("supress_noise_2_records", "MISMATCH sup2"),
    ("dat", "MISMATCH sketches"),
    ("score", "MISMATCH score"),
    ("fmt6", "MISMATCH fmt6"),
    ("bloom", "MISMATCH bloom"),
])
def test_verify_script_fails_on_one_flipped_byte(tmp_path, flip, needle):
    r = _run(tmp_path, _stubs(tmp_path), flip=flip)
    assert r.returncode != 0, r.stdout + r.stderr
    assert needle in r.stdout + r.stderr, r.stdout + r.stderr
    if flip in ("dat", "score", "fmt6", "bloom"):     # the record channel was fine and said so before the later channel failed
        assert "fixture matches mhap.jar" in r.stdout


def test_verify_script_without_a_jvm_or_a_jar_says_so(tmp_path):
    bindir = _stubs(tmp_path)
    r = _run(tmp_path, bindir, jar=None)
    assert r.returncode == 2 and "set MHAP_JAR" in r.stdout
    # no java on PATH at all (the build container and every GPU box seen so far)
    env = {"PATH": "/nonexistent", "HOME": str(tmp_path)}
    r = subprocess.run(["/bin/sh", os.path.join(GOLD, "verify_against_jar.sh")], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "no java on PATH" in r.stdout


def test_verify_script_without_javac_compares_records_and_sketches_only(tmp_path):
    import shutil
    if shutil.which("javac"):
        pytest.skip("a real javac is on PATH")
    r = _run(tmp_path, _stubs(tmp_path, with_javac=False))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fixture matches mhap.jar" in r.stdout and "ok       sketches" in r.stdout
    assert "no javac on PATH" in r.stdout
