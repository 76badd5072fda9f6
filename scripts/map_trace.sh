#!/bin/bash
# rocprofv3 kernel-trace summary of the single-stream laserMapping bench (run through gpurun).  usage: bash scripts/map_trace.sh <tag>
O=$PWD/gpurun_out/${1:-map_tr}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --workload map --scans 64 --streams 1 --cpu-sample 0 --no-extras > $O/b.json 2> $O/err.txt
find $O/t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats.csv
rm -rf $O/t
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats.csv")))
fr=[r for r in rows if "k_map_commit" in r["Name"]]
n=int(fr[0]["Calls"]) if fr else 1
tot=0
for r in rows:
    if "lmono::k_map" in r["Name"] or "k_vox" in r["Name"] or "k_grid" in r["Name"] or "k_copy" in r["Name"] or "rocclr" in r["Name"]:
        per=float(r["TotalDurationNs"])/n/1e3; tot+=per
        print("%-28s %5.1f calls/frame  %6.1f us avg  %6.1f us/frame" % (r["Name"].split("(")[0].replace("lmono::",""), int(r["Calls"])/n, float(r["AverageNs"])/1e3, per))
print("kernel time per frame %.1f us over %d frames" % (tot, n))
PY
