#!/bin/bash
# HBM traffic of one bf16 training step of MambaUNet (model-level roofline fraction): two rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; never combined with a trace domain) over a short run, summed per step by tools/model_hbm.py.
#   bash tools/model_hbm.sh <tag> [ms_per_step]     -> gpurun_out/<tag>/model_hbm.json
set -u
TAG=${1:-model_hbm}
MS=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- \
        python3 "$ROOT/bench_model.py" --mode train --dtype bf16 --steps 5 --warmup 4 > "$OUT/pmc_$c.log" 2>&1 || { echo "pmc $c failed"; tail -5 "$OUT/pmc_$c.log"; exit 1; }
    echo "pmc $c ok"
done
python3 "$ROOT/tools/model_hbm.py" "$OUT" "$MS" > "$OUT/model_hbm.json" && cat "$OUT/model_hbm.json"
