#!/bin/bash
# Two SQ counter passes of bench.py (3 steps) -> gpurun_out/<tag>/pmc_summary.txt.  Usage: bash tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    name=$(echo "$pass" | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o pmc -- \
        python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-model > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.txt" && cat "$OUT/pmc_summary.txt"
