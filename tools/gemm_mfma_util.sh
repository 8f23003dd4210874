#!/bin/bash
# MFMA utilisation of the package's own GEMM kernels (run on the GPU box through gpurun, from the repo root):
#   bash tools/gemm_mfma_util.sh <tag> [fp32]   ->  gpurun_out/<tag>/gemm_mfma_util.txt   (copy to profiles/rNN/)
# Two rocprofv3 passes over tools/gemm_mfma_util.py, the program directly behind `--`: durations (--kernel-trace), counters (--pmc).
set -u
TAG=${1:-mfma}
DT=${2:-bf16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/tools/gemm_mfma_util.py" $DT > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o p -- python3 "$ROOT/tools/gemm_mfma_util.py" $DT > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 "$ROOT/tools/gemm_mfma_util.py" --parse "$OUT/trace" "$OUT/pmc" $DT > "$OUT/gemm_mfma_util.txt" && cat "$OUT/gemm_mfma_util.txt"
