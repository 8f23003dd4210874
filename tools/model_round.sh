#!/bin/bash
# Model-level numbers of a round (run on the GPU box through gpurun, from the repo root): bash tools/model_round.sh <tag>
# -> gpurun_out/<tag>/bench_model.jsonl (one JSON line per configuration) and a steady-state kernel breakdown of the
#    bf16 training step (rocprofv3 --kernel-trace --stats; MIOpen's one-time convolution search shows up at 0.2 calls/step).
set -u
TAG=${1:-model}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
: > "$OUT/bench_model.jsonl"
run() { timeout -k 10 200 python3 "$ROOT/bench_model.py" "$@" >> "$OUT/bench_model.jsonl" 2>> "$OUT/bench_model.err" || { echo "failed: $*"; tail -5 "$OUT/bench_model.err"; exit 1; }; }
run --mode train --dtype bf16
run --mode train --dtype fp32
run --mode train --dtype bf16 --graph
run --mode train --dtype bf16 --size 512 --classes 9
run --mode semi --dtype fp32
run --mode semi --dtype bf16
run --mode infer --dtype bf16
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o m -- \
    python3 "$ROOT/bench_model.py" --mode train --dtype bf16 --steps 40 --warmup 10 > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
python3 "$ROOT/tools/kernel_breakdown.py" "$(find "$OUT/trace" -name '*kernel_stats.csv')" 50 60 > "$OUT/train224_bf16_kernel_breakdown.txt"
python3 -c "
import json,sys
for l in open('$OUT/bench_model.jsonl'):
    d=json.loads(l); print(d['metric'], d['dtype'][:12], 'graph' if d['hip_graph'] else '', d['value'], d['ms_per_step'], d['peak_mem_GB'])
"
