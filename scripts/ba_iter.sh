#!/bin/bash
# One GPU iteration on the BA half (run through gpurun): BA parity tests, the single-window and Estimator-loop bench lines, phase cycles.
# usage: bash scripts/ba_iter.sh <tag> [quick]
TAG=${1:-ba}; MODE=${2:-full}
O=gpurun_out/$TAG; mkdir -p $O
if [ "$MODE" = full ]; then TESTS="tests/test_ba_solve_gpu.py tests/test_estimator_loop_gpu.py tests/test_config0_gpu.py tests/test_ba_factors_gpu.py tests/test_marg_gpu.py tests/test_ba_feat.py"; else TESTS="tests/test_ba_solve_gpu.py"; fi
timeout -k 10 500 python -m pytest $TESTS -m gpu -x -q > $O/tests.log 2>&1
rc=$?
tail -4 $O/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python bench.py --workload ba --windows 1 > $O/ba1.json 2> $O/ba1.err && python - <<PY
import json
d=json.loads(open("$O/ba1.json").read().strip().splitlines()[-1]); print("ba --windows 1: ms_per_step", d["ms_per_step"], "rel diff vs cpu", d.get("final_cost_rel_diff_vs_cpu"))
PY
timeout -k 10 120 python bench.py --workload ba > $O/ba1024.json 2> $O/ba1024.err && python - <<PY
import json
d=json.loads(open("$O/ba1024.json").read().strip().splitlines()[-1]); print("ba --windows 1024: windows/s", d["value"])
PY
timeout -k 10 200 python bench.py --workload ba-seq --frames-seq 1000 --cpu-frames 0 > $O/baseq.json 2> $O/baseq.err && python - <<PY
import json
d=json.loads(open("$O/baseq.json").read().strip().splitlines()[-1]); print("ba-seq 1000 frames: frames/s", d["value"], "inline", d["config"].get("inline_marginalisation"))
PY
bash scripts/prof_ba_phases.sh 1 > $O/phases_w1.txt 2>&1; tail -1 $O/phases_w1.txt
bash scripts/prof_ba_seq_phases.sh > $O/phases_seq.txt 2>&1
python - <<PY
import re
L=[l for l in open("$O/phases_seq.txt") if l.startswith("PROF")]
def nums(l): return [int(x) for x in re.findall(r"-?\d+", l)]
if len(L)>=2:
    a,b=nums(L[-2]),nums(L[-1]); print("seq window (last - previous):", [y-x for x,y in zip(a,b)])
PY
