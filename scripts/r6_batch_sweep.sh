#!/bin/bash
# EstimatorBatch: frames/s of the lock-step frame loop at 1 / 8 / 64 / 256 streams (bench.py --workload ba-seq --seq-streams N), every stream's digest checked
# against the single-stream run of its file.  usage (GPU box): bash scripts/r6_batch_sweep.sh [frames] [tag]
set -u
F=${1:-2761}; O=gpurun_out/${2:-batch_sweep}; mkdir -p $O
for N in 8 64 256; do
  LMONO_HOST_TIMING=1 timeout -k 10 900 python3 bench.py --workload ba-seq --frames-seq $F --seq-streams $N > $O/ba_seq_${N}streams.json 2> $O/ba_seq_${N}streams.err || { echo "N=$N failed"; tail -5 $O/ba_seq_${N}streams.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/ba_seq_${N}streams.json')); c=d['config']
print('streams', c['streams'], 'frames/s', d['value'], 'ms/lockstep', c['ms_per_lockstep_frame'], 'inline', c['inline_marginalisation']['frames_per_s'], 'equal', c['every_stream_equals_its_single_stream_run'], 'single', c['single_stream_frames_per_s'])"
done
