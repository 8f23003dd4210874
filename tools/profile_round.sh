#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun, from the repo root):
#   bash tools/profile_round.sh <tag>
# 1. bench.py (the numbers of record)   2. rocprofv3 --kernel-trace --stats of the same command
# 3. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) -- never combined with a trace domain.
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 "$ROOT/bench.py" --steps 50 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o stats -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-model > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
echo "trace ok"
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    name=$(echo "$pass" | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o pmc -- \
        python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-model > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }   # stop: no GPU step after a failed one
    echo "pmc $name ok"
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" "$OUT/bench_pmc.json" > "$OUT/pmc_summary.txt" && cat "$OUT/pmc_summary.txt"
# keep for profiles/rNN/: bench.json, pmc_summary.txt, bench_pmc.json (bench.py reads the newest profiles/r*/bench_pmc.json),
# trace/**/stats_kernel_stats.csv
