# tests/fuzz_parity.py against the round's final kernels on the GPU box: ROUND=r06 bash tools/fuzz_campaign.sh  ->  gpurun_out/$R/${R}_fuzz_summary.txt
# (random flags, read mixes, repeat families, -f filters; sorted record lines GPU vs oracle; every line: draws, setting, failures)
R=${ROUND:-r06}
mkdir -p gpurun_out/$R
OUT=gpurun_out/$R/${R}_fuzz_${FUZZ_TAG:-summary}.txt
echo "# tests/fuzz_parity.py against the round-6 kernels (line table of the first and middle query tiers with queued hits, wave-reduced elements counter, seeded MinHash row items, join kernel's wide units compiled alone, tagged rendezvous, small-shard index build shapes, the ordered kernel's two launches), run on an MI355X box:" > $OUT
run() {  # draws seed label env...
  n=$1; seed=$(( $2 + ${SEED_OFFSET:-0} )); label=$3; shift 3   # (SEED_OFFSET: a second campaign with draws of its own)
  f=$(env "$@" timeout 3000 python tests/fuzz_parity.py $n $seed 2>/dev/null | tail -1)
  echo "$n draws, $label: $f" >> $OUT
}
run ${N1:-400} 50000 "default"
run ${NL1:-300} 60000 "line table forced on these small indexes (MHAP_INDEX_LINES=1)" MHAP_INDEX_LINES=1
run ${NL2:-200} 61000 "line table, packed lines (MHAP_INDEX_LINE_LOAD=14: partner spills, 'see the buckets' lines)" MHAP_INDEX_LINES=1 MHAP_INDEX_LINE_LOAD=14
run ${NL3:-150} 62000 "line table + middle tier forced" MHAP_INDEX_LINES=1 MHAP_INDEX_MID=1
run ${NL4:-100} 63000 "line table, sparse lines, 128-query chunks" MHAP_INDEX_LINES=1 MHAP_INDEX_LINE_LOAD=1 MHAP_QUERY_CHUNK=128
run ${N2:-150} 51000 "queue-overflow variant of the MinHash kernel (qcap64: every full row takes the exact redo)" MHAP_LIB_PATH=mhap_amd/lib/variants/libmhaphip_qcap64.so
run ${N3:-100} 52000 "MHAP_MINHASH=classic (the general kernel for the weight-1 strands too)" MHAP_MINHASH=classic
run ${N4:-80} 53000 "MHAP_MINHASH=perchain" MHAP_MINHASH=perchain
run ${N5:-100} 54000 "TEAM shape pinned + prune forced" MHAP_JOIN_MODE=team MHAP_OVERLAP_PRUNE=1
run ${N6:-80} 55000 "128-query chunks (post stage on the worker thread)" MHAP_QUERY_CHUNK=128
run ${N7:-80} 56000 "dense tier, class-ordered, 64-entry passes" MHAP_INDEX_DENSE=1 MHAP_INDEX_GROUP=1 MHAP_INDEX_GROUP_T=4 MHAP_INDEX_CLASS_LOG=6 MHAP_DENSE_RANGE_LOG=6

run ${N8:-150} 70000 "FUZZ_WIDE corners (--num-hashes 700 .. 4096, --ordered-sketch-size 1 .. 8192, numMinMatches to 200)" FUZZ_WIDE=1
run ${NS1:-150} 92000 "ordered kernel in two launches forced on these small jobs (MHAP_ORDERED_SPLIT=55, the weighted MinHash launch beside the first)" MHAP_ORDERED_SPLIT=55
run ${NS2:-100} 93000 "the same with the weighted launch waiting, odd split (MHAP_ORDERED_SPLIT=33 MHAP_ORDERED_NOWAIT=0)" MHAP_ORDERED_SPLIT=33 MHAP_ORDERED_NOWAIT=0
run ${NS3:-100} 94000 "index build: 256-entry tiles, the 8-KB bins shape (MHAP_INDEX_TILE=256 MHAP_INDEX_BINS_SHAPE=1)" MHAP_INDEX_TILE=256 MHAP_INDEX_BINS_SHAPE=1
run ${NS4:-60} 95000 "index build: 4096-entry tiles, the 32-KB bins shape (rounds 1-5)" MHAP_INDEX_TILE=4096 MHAP_INDEX_BINS_SHAPE=2
run ${N9:-100} 91000 "wide join passes off (MHAP_JOIN_WIDE=0: the lane kernel with its bounded selection takes every pair of more than 128 joined k-mers)" MHAP_JOIN_WIDE=0
cat $OUT
