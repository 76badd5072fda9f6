#!/bin/bash
# which configuration of the 64-stream lock-step run differs from the single-stream runs?  digests per (mode, cluster size)
O=gpurun_out/dbg64; mkdir -p $O
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
for k in range(4):
    K.write_stream('$O/s%d.bin' % k, K.make_stream(${1:-150}, seed=2 + k, stops=()))
PY
E=lmono_amd/host/estimator_seq
export LMONO_BA_REPORT_RETRIES=1
for k in 0 1 2 3; do $E $O/s$k.bin - sync | grep "^DIG" | awk '{print $3}' > $O/single$k.txt; done
echo "single: $(cat $O/single0.txt) $(cat $O/single1.txt) $(cat $O/single2.txt) $(cat $O/single3.txt)"
for N in ${2:-64}; do
for cfg in "sync 0" "async 0" "sync 1" "sync 2" "sync 4" "async 2"; do
  set -- $cfg
  LMONO_BA_CLUSTER=$2 $E $O/s0.bin - $1 streams=$N digest $O/s1.bin $O/s2.bin $O/s3.bin > $O/out_$1_$2.txt 2> $O/err_$1_$2.txt
  python3 - <<PY
dig = {int(l.split()[1]): l.split()[2] for l in open("$O/out_$1_$2.txt") if l.startswith("DIG")}
single = [open("$O/single%d.txt" % k).read().strip() for k in range(4)]
bad = [s for s in sorted(dig) if dig[s] != single[s % 4]]
print("N=$N $1 cluster $2: %d streams, %d differ %s | %s" % (len(dig), len(bad), bad[:12], open("$O/err_$1_$2.txt").read().strip()[-200:]))
PY
done
done
