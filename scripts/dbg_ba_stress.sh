#!/bin/bash
# stress: many fresh processes of the BA solve (batch of seven, every K) and of the Estimator loop; prints anything that differs or fails
mkdir -p gpurun_out/stress
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from workloads import s2 as K
st = K.make_stream(150, seed=2)
K.write_stream('gpurun_out/stress/stream150.bin', st)
PY
for i in 1 2 3 4 5 6; do
  for k in 1 8; do
    timeout -k 5 100 python scripts/dbg_ba_cluster.py seven $k 3 2>&1 | grep -v amdgpu.ids | grep -v "same as run 0 \[30, 30, 30, 30, 4, 30, 30\] \[1, 1, 1, 1, 0, 1, 1\]"
    LMONO_BA_CLUSTER=$k timeout -k 5 100 lmono_amd/host/estimator_seq gpurun_out/stress/stream150.bin gpurun_out/stress/odo_${k}_$i.txt sync > gpurun_out/stress/out_${k}_$i.txt 2> gpurun_out/stress/err_${k}_$i.txt || { echo "estimator_seq K=$k run $i FAILED"; tail -2 gpurun_out/stress/err_${k}_$i.txt; }
  done
done
md5sum gpurun_out/stress/odo_*.txt | awk '{print $1}' | sort | uniq -c
rm -f gpurun_out/stress/stream150.bin
