#!/bin/bash
# Round-6 evidence (run on the GPU box through gpurun, AFTER the last kernel change of the round):
#   headline  kernel-trace summaries of the default bench command (with / without boundary validation), HBM traffic + SQ instruction counters of the
#             main pass (separate passes), pmc_k_correspond.json (traffic AND VALU / SALU wave instructions per launch: bench.py's roofline.issue_frac),
#             the plain bench lines of the three worlds
#   ba        K sweep, kernel traces (1024 windows, 1 window, Estimator loop), MFMA / VALU counters and FETCH / WRITE traffic of k_ba_solve (1 and 1024
#             windows), k_marginalize phase cycles (LMONO_MG_PROF build in a scratch directory)
#   streams   EstimatorBatch: frames/s at 8 / 64 / 256 streams (bench.py --workload ba-seq --seq-streams N), every stream's digest against its single-stream run,
#             the lock-step frame's phase clocks
#   map       laserMapping single stream / 64 streams, kernel trace, FETCH / WRITE traffic of a frame
# usage: bash scripts/profile_round6.sh <tag> [headline|ba|streams|map]   (one gpurun call of <= 20 minutes per part)
set -u
TAG=${1:-final}; PART=${2:-headline}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
stats() { find $1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $2; rm -rf $1; }
if [ "$PART" = headline ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $OUT/final_bench_under_rocprof.json 2> $OUT/trace.err
stats $OUT/trace $OUT/final_kernel_stats_4541scans.csv
LMONO_BOUNDARY_TOL=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace0 -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $OUT/final_bench_under_rocprof_no_validation.json 2> $OUT/trace0.err
stats $OUT/trace0 $OUT/final_kernel_stats_4541scans_main_pass_only.csv
: > $OUT/final_pmc_4541scans.txt
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  LMONO_BOUNDARY_TOL=0 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $OUT/pmc_$tag.out 2> $OUT/pmc_$tag.err
  echo "## $grp" >> $OUT/final_pmc_4541scans.txt
  python3 scripts/pmc_summary.py $OUT/pmc_$tag >> $OUT/final_pmc_4541scans.txt 2>&1
  rm -rf $OUT/pmc_$tag $OUT/pmc_$tag.out
done
# pmc_k_correspond.json from THIS pass's counters (regenerated in the profile script's own run)
python3 - <<PY
import re, json
cur = None; val = {}
for ln in open("$OUT/final_pmc_4541scans.txt"):
    if ln.startswith("## "): cur = ln[3:].strip(); continue
    if "k_corr_flat" in ln:
        for k, v in re.findall(r"'(\w+)': (\d+)", ln): val[k] = float(v)
out = {"kernel": "k_correspond (k_corr_flat, the default search)", "chain_groups": 4,
       "source": "profiles/r6/final_pmc_4541scans.txt (scripts/profile_round6.sh headline: rocprofv3 --pmc in separate passes, LMONO_BOUNDARY_TOL=0 bench.py --steps 1 --warmup 0 "
                 "--cpu-sample 0 --no-extras: main-pass launches only, 64 chains per launch = 256 chains in 4 groups, lead 6; averages per launch)",
       "correction": "gfx950: FETCH_SIZE (KB) counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is"}
if "FETCH_SIZE" in val and "WRITE_SIZE" in val:
    out["fetch_size_bytes_per_launch"] = round(val["FETCH_SIZE"] * 1024); out["write_size_bytes_per_launch"] = round(val["WRITE_SIZE"] * 1024)
    out["hbm_bytes_per_launch"] = round((2 * val["FETCH_SIZE"] + val["WRITE_SIZE"]) * 1024)
if "TCC_HIT_sum" in val and "TCC_MISS_sum" in val: out["l2_hit_rate"] = round(val["TCC_HIT_sum"] / (val["TCC_HIT_sum"] + val["TCC_MISS_sum"]), 4)
for k, name in (("SQ_INSTS_VALU", "valu_wave_insts_per_launch"), ("SQ_INSTS_SALU", "salu_wave_insts_per_launch"), ("SQ_INSTS_VMEM_RD", "vmem_rd_wave_insts_per_launch"), ("SQ_INSTS_LDS", "lds_wave_insts_per_launch")):
    if k in val: out[name] = val[k]
json.dump(out, open("$OUT/pmc_k_correspond.json", "w"), indent=1)
print("k_corr_flat per launch:", {k: v for k, v in out.items() if k.endswith("per_launch") or k == "l2_hit_rate"})
PY
echo "[headline] traces and counters done"
timeout -k 10 400 python3 bench.py > $OUT/final_bench_4541scans.json 2> $OUT/bench.err
timeout -k 10 300 python3 bench.py --seq 1 --no-extras --cpu-sample 0 > $OUT/final_bench_seq1_held_out.json 2> $OUT/bench1.err
timeout -k 10 300 python3 bench.py --seq 2 --no-extras --cpu-sample 0 > $OUT/final_bench_seq2_stress.json 2> $OUT/bench2.err
head -6 $OUT/final_kernel_stats_4541scans_main_pass_only.csv | cut -c1-200
tail -1 $OUT/final_bench_4541scans.json | cut -c1-400
fi
if [ "$PART" = ba ]; then
bash scripts/ba_ksweep.sh prof_$TAG > /dev/null 2>&1
mv $OUT/ksweep.txt $OUT/ba_workgroups_per_window_K1_2_4_8.txt 2>/dev/null
rm -f $OUT/ba1_k*.err $OUT/seq_k*.err $OUT/ba1_k[124].json $OUT/seq_k[124].json
timeout -k 10 300 python3 bench.py --workload ba-seq > $OUT/final_bench_ba_seq_2761frames.json 2> $OUT/baseq.err
timeout -k 10 300 python3 bench.py --workload ba --windows 1024 > $OUT/ba_bench_1024windows.json 2> $OUT/ba1024.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ba -- python3 bench.py --workload ba --windows 1024 --steps 3 --warmup 1 > $OUT/ba_bench_1024windows_under_rocprof.json 2> $OUT/trace_ba.err
stats $OUT/trace_ba $OUT/ba_kernel_stats_1024windows.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ba1 -- python3 bench.py --workload ba --windows 1 --steps 20 --warmup 2 > $OUT/ba_bench_1window_under_rocprof.json 2> $OUT/trace_ba1.err
stats $OUT/trace_ba1 $OUT/ba_kernel_stats_1window.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_baseq -- python3 bench.py --workload ba-seq --frames-seq 600 --cpu-frames 0 > $OUT/ba_seq_bench_600frames_under_rocprof.json 2> $OUT/trace_baseq.err
stats $OUT/trace_baseq $OUT/ba_seq_kernel_stats_600frames.csv
: > $OUT/ba_pmc_k_ba_solve.txt
for win in 1 1024; do
  i=0
  for grp in "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY" \
             "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmcba -- python3 bench.py --workload ba --windows $win --steps 1 --warmup 0 > /dev/null 2> $OUT/pmcba.err || echo "pmc group $i ($win windows) failed" >> $OUT/ba_pmc_k_ba_solve.txt
    case "$grp" in FETCH_SIZE|WRITE_SIZE) echo "## $win window(s), $grp (KB per launch)" >> $OUT/ba_pmc_k_ba_solve.txt;; *) echo "## $win window(s), counter group $i" >> $OUT/ba_pmc_k_ba_solve.txt;; esac
    python3 scripts/pmc_summary.py $OUT/pmcba 2>&1 | grep "k_ba_solve" >> $OUT/ba_pmc_k_ba_solve.txt
    rm -rf $OUT/pmcba
  done
done
python3 - <<PY
import re, json
cur = None; val = {}
for ln in open("$OUT/ba_pmc_k_ba_solve.txt"):
    m = re.match(r"## (\d+) window\(s\), (FETCH_SIZE|WRITE_SIZE)", ln)
    if m: cur = (m.group(1), m.group(2)); continue
    if cur:
        k = re.search(r"'" + cur[1] + r"': (\d+)", ln)
        if k: val[cur] = float(k.group(1)); cur = None
out = {"source": "profiles/r6/ba_pmc_k_ba_solve.txt (scripts/profile_round6.sh ba: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --workload ba --windows 1 | 1024 --steps 1 --warmup 0; KB per launch)",
       "correction": "gfx950: FETCH_SIZE (KB) counts 128-B requests at 64 B -> doubled; WRITE_SIZE as is", "hbm_bytes_per_launch": {}}
for w in ("1", "1024"):
    if (w, "FETCH_SIZE") in val and (w, "WRITE_SIZE") in val:
        out["hbm_bytes_per_launch"][w] = round((2 * val[(w, "FETCH_SIZE")] + val[(w, "WRITE_SIZE")]) * 1024)
json.dump(out, open("$OUT/pmc_k_ba_solve.json", "w"), indent=1)
print("k_ba_solve traffic per launch:", out["hbm_bytes_per_launch"])
PY
bash scripts/prof_marg.sh prof_$TAG/mg > $OUT/marg_phase_cycles.txt 2>&1
rm -rf $OUT/mg
cat $OUT/ba_workgroups_per_window_K1_2_4_8.txt; head -4 $OUT/ba_seq_kernel_stats_600frames.csv | cut -c1-160; tail -7 $OUT/marg_phase_cycles.txt
echo "[ba] done"
fi
if [ "$PART" = streams ]; then
: > $OUT/ba_seq_streams.txt
# (streams, groups): one lock-step batch per size, then 256 streams as two batches of 128 driven by one thread (one batch's host passes under the other's solve)
for NG in "8 1" "64 1" "256 1" "256 2"; do
  set -- $NG; N=$1; G=$2; F=ba_seq_${N}streams; [ "$G" != 1 ] && F=${F}_${G}groups
  LMONO_HOST_TIMING=1 timeout -k 10 1000 python3 bench.py --workload ba-seq --seq-streams $N --seq-groups $G > $OUT/$F.json 2> $OUT/$F.err || { echo "N=$N G=$G failed" >> $OUT/ba_seq_streams.txt; continue; }
  python3 -c "
import json; d=json.load(open('$OUT/$F.json')); c=d['config']
print('streams', c['streams'], '| groups', c['groups'], '| frames/s', d['value'], '| ms per lock-step frame', c['ms_per_lockstep_frame'], '| inline marginalisation', c['inline_marginalisation']['frames_per_s'], '| every stream = its single-stream run:', c['every_stream_equals_its_single_stream_run'], '(%d files)' % c['files_verified'], '| single stream', c['single_stream_frames_per_s'], 'frames/s')" >> $OUT/ba_seq_streams.txt
  grep -h BATCHTIM $OUT/$F.err >> $OUT/ba_seq_streams.txt
done
cat $OUT/ba_seq_streams.txt
echo "[streams] done"
fi
if [ "$PART" = map ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_map -- python3 bench.py --workload map --scans 64 --streams 1 --no-extras > $OUT/map_bench_1stream_under_rocprof.json 2> $OUT/trace_map.err
stats $OUT/trace_map $OUT/map_kernel_stats_1stream.csv
timeout -k 10 300 python3 bench.py --workload map --scans 64 --streams 1 > $OUT/map_bench_1stream.json 2>> $OUT/trace_map.err
timeout -k 10 300 python3 bench.py --workload map --scans 64 --streams 64 > $OUT/map_bench_64streams.json 2>> $OUT/trace_map.err
: > $OUT/map_pmc_1stream.txt
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_map -- python3 bench.py --workload map --scans 64 --streams 1 --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $OUT/pmc_map.err
  echo "## $grp (KB, summed over the launches of the run: 64 frames)" >> $OUT/map_pmc_1stream.txt
  python3 scripts/pmc_summary.py $OUT/pmc_map --sum 2>&1 | grep -E "k_map|k_vox|k_grid|k_copy" >> $OUT/map_pmc_1stream.txt
  rm -rf $OUT/pmc_map
done
python3 - <<PY
import re, json
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
cur = None
for ln in open("$OUT/map_pmc_1stream.txt"):
    if ln.startswith("## "): cur = ln.split()[1]; continue
    m = re.search(r"'" + (cur or "x") + r"': (\d+)", ln)
    if m: tot[cur] += float(m.group(1))
frames = 64
d = {"source": "profiles/r6/map_pmc_1stream.txt (scripts/profile_round6.sh map: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --workload map --scans 64 --streams 1 --steps 1 --warmup 0 --no-extras: every k_map_* / k_vox_* / k_grid_* / k_copy_* launch of 64 frames)",
     "fetch_kb_total": tot["FETCH_SIZE"], "write_kb_total": tot["WRITE_SIZE"], "frames": frames,
     "correction": "gfx950: FETCH_SIZE (KB) counts 128-B requests at 64 B -> doubled; WRITE_SIZE as is",
     "hbm_bytes_per_frame": round((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / frames)}
open("$OUT/pmc_map_frame.json", "w").write(json.dumps(d, indent=1) + "\n")
print("map frame traffic:", d["hbm_bytes_per_frame"], "B")
PY
python3 -c "
import json
for f in ('map_bench_1stream', 'map_bench_64streams'):
    d = json.load(open('$OUT/' + f + '.json')); print(f, d['value'], d['unit'])"
echo "[map] done"
fi
find $OUT -name "*.err" -size 0 -delete
