# end-to-end probe of the native driver (run on the GPU box): wall time and the host-side timeline of `mhap-hip -s reads.fasta`
#   bash tools/e2e_probe.sh [c2|c4] [extra mhap-hip flags, e.g. --devices 0,0]
cfg=${1:-c2}; shift
python - <<PY
import sys, time; sys.path.insert(0,'.')
from mhap_amd import workloads as W
t=time.time(); fa=W.config_reads('$cfg'); W.write_fasta(fa,'/tmp/$cfg.fasta'); print("# wrote /tmp/$cfg.fasta in %.1f s" % (time.time()-t))
PY
ls -la /tmp/$cfg.fasta | awk '{print "# FASTA bytes:", $5}'
for i in 1 2 3; do s=$(date +%s.%N); mhap_amd/lib/mhap-hip -s /tmp/$cfg.fasta "$@" > /tmp/out.txt 2> /tmp/err.txt; e=$(date +%s.%N); python3 -c "print(\"wall %.3f s\" % ($e - $s))"; wc -l < /tmp/out.txt; done
sort /tmp/out.txt | sha256sum | cut -c1-16
grep -E "Time|Stored|Using" /tmp/err.txt | head -12
MHAP_HOST_PROF=1 mhap_amd/lib/mhap-hip -s /tmp/$cfg.fasta "$@" 2>&1 >/dev/null | grep "\[host\]\|\[cli\]\|\[ingest\]" | grep -v "ids built\|ids h2d\|meta mirrored\|finish_add" | head -90
rm -f /tmp/$cfg.fasta
