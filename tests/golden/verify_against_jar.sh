#!/bin/sh
# Re-derives the golden records with a real MHAP jar when a JVM is available and diffs them against the
# committed oracle-derived fixture: the self overlap, the -q index-vs-stream run (queries below --min-olap-length included:
# id offsets), the -f filter run and both --supress-noise modes.
# Then (needs javac too) compiles tools/jvm/ScoreTableDump.java against the jar's Guava and diffs, bit for bit, the three pieces of
# JDK / Guava arithmetic the restatement depends on with tools/jvm/native_dump.py: the (inter, k) identity table (Math.log / Math.exp
# against glibc), String.format("%.6f") on 20 000 doubles incl. ties, and BloomFilter sizing + membership.
# Also: the MinHash + ordered sketches of the fixture as a `.dat` file, mhap.jar -p against tools/jvm/native_dump.py dat, byte for byte.
# Runtime on a laptop-class JVM: under a minute (six runs of mhap.jar on 60 short reads + three small dumps).
# Usage: MHAP_JAR=/path/mhap-2.1.3.jar sh tests/golden/verify_against_jar.sh
set -e
cd "$(dirname "$0")"
command -v java >/dev/null 2>&1 || { echo "no java on PATH: cannot verify (fixture stays oracle-derived)"; exit 2; }
[ -n "$MHAP_JAR" ] || { echo "set MHAP_JAR"; exit 2; }
FLAGS="-k 16 --num-hashes 64 --ordered-kmer-size 12 --ordered-sketch-size 256 --min-olap-length 116 --num-threads 1"
T=$(mktemp -d)
java -jar "$MHAP_JAR" -s small_reads.fasta $FLAGS 2>/dev/null | sort > $T/self.txt
java -jar "$MHAP_JAR" -s small_reads.fasta -q small_queries.fasta --no-self $FLAGS 2>/dev/null | sort > $T/query.txt
THR=$(python3 -c 'import json; print(json.load(open("small_reads.json"))["filter_threshold"])')
java -jar "$MHAP_JAR" -s small_reads.fasta -f small_kmers.txt --filter-threshold $THR $FLAGS 2>/dev/null | sort > $T/filter.txt
java -jar "$MHAP_JAR" -s small_reads.fasta -f small_kmers.txt --filter-threshold $THR --supress-noise 1 $FLAGS 2>/dev/null | sort > $T/sup1.txt
java -jar "$MHAP_JAR" -s small_reads.fasta -f small_kmers.txt --filter-threshold $THR --supress-noise 2 $FLAGS 2>/dev/null | sort > $T/sup2.txt
python3 - "$T" <<'PY'
import json, sys
T = sys.argv[1]
g = json.load(open("small_reads.json"))
bad = 0
for name, key in (("self", "sorted_records"), ("query", "query_records_no_self"), ("filter", "filter_records"),
                  ("sup1", "supress_noise_1_records"), ("sup2", "supress_noise_2_records")):
    got = sorted(l.rstrip("\n") for l in open(f"{T}/{name}.txt") if l.strip())
    ok = got == sorted(g[key])
    print(("ok      " if ok else "MISMATCH"), name, len(got), "records from mhap.jar,", len(g[key]), "in the fixture")
    bad += not ok
assert bad == 0, "MISMATCH between mhap.jar and the oracle-derived fixture"
print("fixture matches mhap.jar")
PY
# Second channel, no record formatting involved: the sketches themselves.  `-p` makes mhap.jar write the MinHash + ordered sketches of
# every strand as small_reads.dat; tools/jvm/native_dump.py writes the same file from the restatement (no GPU needed); cmp, byte for byte.
ROOT=$(cd ../.. && pwd)
mkdir -p "$T/in" "$T/jdat"
cp small_reads.fasta "$T/in/"
java -jar "$MHAP_JAR" -p "$T/in" -q "$T/jdat" $FLAGS >/dev/null 2>&1
python3 "$ROOT/tools/jvm/native_dump.py" dat small_reads.fasta "$T/native.dat" 16 64 12 256 116 >/dev/null
if cmp -s "$T/jdat/small_reads.dat" "$T/native.dat"; then echo "ok       sketches: small_reads.dat ($(wc -c < "$T/native.dat") bytes) identical to mhap.jar -p"
else echo "MISMATCH sketches: mhap.jar -p and the restatement write different .dat files"; cmp "$T/jdat/small_reads.dat" "$T/native.dat" | head -2; exit 1; fi
if command -v javac >/dev/null 2>&1; then
  javac -cp "$MHAP_JAR" -d "$T" "$ROOT/tools/jvm/ScoreTableDump.java"
  for what in "score 12 1536" "fmt6" "bloom"; do
    name=$(echo $what | cut -d' ' -f1)
    java -cp "$MHAP_JAR:$T" ScoreTableDump $what > "$T/$name.jvm"
    python3 "$ROOT/tools/jvm/native_dump.py" $what > "$T/$name.native"
    if cmp -s "$T/$name.jvm" "$T/$name.native"; then echo "ok       $name: $(wc -l < "$T/$name.jvm") lines identical to the JVM's"
    else echo "MISMATCH $name: first differing lines:"; diff "$T/$name.jvm" "$T/$name.native" | head -6; exit 1; fi
  done
  echo "JDK / Guava arithmetic matches the restatement"
else
  echo "no javac on PATH: the arithmetic dumps (tools/jvm) were not compared"
fi
