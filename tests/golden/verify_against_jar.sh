#!/bin/sh
# Re-derives the golden records with a real MHAP jar when a JVM is available and diffs them against the
# committed oracle-derived fixture.  Usage: MHAP_JAR=/path/mhap-2.1.3.jar sh tests/golden/verify_against_jar.sh
set -e
cd "$(dirname "$0")"
command -v java >/dev/null 2>&1 || { echo "no java on PATH: cannot verify (fixture stays oracle-derived)"; exit 2; }
[ -n "$MHAP_JAR" ] || { echo "set MHAP_JAR"; exit 2; }
java -jar "$MHAP_JAR" -s small_reads.fasta -k 16 --num-hashes 64 --ordered-kmer-size 12 --ordered-sketch-size 256 \
     --min-olap-length 116 --num-threads 1 2>/dev/null | sort > /tmp/jar_records.txt
python3 - <<'PY'
import json
want = json.load(open("small_reads.json"))["sorted_records"]
got = [l.rstrip("\n") for l in open("/tmp/jar_records.txt")]
assert sorted(got) == sorted(want), "MISMATCH between mhap.jar and the oracle-derived fixture"
print("fixture matches mhap.jar:", len(got), "records")
PY
