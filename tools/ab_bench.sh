#!/bin/bash
# A/B of library builds on ONE box (boxes differ by ~10 %): bash tools/ab_bench.sh name:path.so [name:path.so ...]
set -u
for v in "$@"; do
    n=${v%%:*}; l=${v##*:}
    SS2D_HIP_LIB=$l timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model > gpurun_out/bn_$n.json 2> gpurun_out/bn_$n.err || { echo "$n failed"; tail -5 gpurun_out/bn_$n.err; exit 1; }
    python -c "import json,sys; d=json.load(open(sys.argv[1])); print('%-8s' % sys.argv[2], d['value'], d['kernels_us'])" gpurun_out/bn_$n.json $n
done
