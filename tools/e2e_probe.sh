# end-to-end probe of the native driver on the C2 FASTA (run on the GPU box): wall time and the host-side timeline
python - <<'PY'
import sys; sys.path.insert(0,'.')
from mhap_amd import workloads as W
fa=W.config_reads('c2')
W.write_fasta(fa,'/tmp/c2.fasta')
PY
for i in 1 2; do s=$(date +%s.%N); mhap_amd/lib/mhap-hip -s /tmp/c2.fasta > /tmp/out.txt 2> /tmp/err.txt; e=$(date +%s.%N); python3 -c "print(\"wall %.3f s\" % ($e - $s))"; wc -l < /tmp/out.txt; done
cat /tmp/err.txt | head -30
MHAP_HOST_PROF=1 mhap_amd/lib/mhap-hip -s /tmp/c2.fasta 2>&1 >/dev/null | grep "\[host\]\|\[cli\]" | head -70
