#!/bin/bash
# ThreadSanitizer over the lock-step frame loop (EstimatorBatch + its host pool + the marginalisation worker), CPU only: the host mirror over the oracle shim.
# usage (no GPU needed): bash scripts/tsan_estimator_batch.sh [frames] [streams] [groups]
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
make -s -C oracle estimator_seq_cpu
H=lmono_amd/host
g++ -O1 -g -fsanitize=thread -march=x86-64-v3 -ffp-contract=off -std=c++17 -pthread -I$H oracle/cpu_shim.cpp $H/lmono_host.cpp $H/estimator_seq.cpp $H/kitti_io.cpp oracle/cpu_shim_stubs.o \
    -o $T/eseq_tsan -Loracle -llmono_oracle -Wl,-rpath,$PWD/oracle -lm
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
K.write_stream('$T/s.bin', K.make_stream(${1:-40}, seed=2, stops=()))
PY
LMONO_HOST_THREADS=4 $T/eseq_tsan $T/s.bin - async streams=${2:-8} groups=${3:-2} digest 2>&1 | tail -4
rm -rf $T
