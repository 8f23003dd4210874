#!/bin/bash
# A/B of the MambaUNet training step on ONE box: bash tools/ab_model.sh "name:ENV=1 ENV2=x" "name2:" ...  (each run: env
# assignments in front of bench_model.py --mode train --dtype bf16; 60 steps after 15; two rounds, interleaved)
set -u
ARGS=${AB_ARGS:---mode train --dtype bf16 --steps 60 --warmup 15}
for round in 1 2; do
for v in "$@"; do
    n=${v%%:*}; e=${v#*:}
    env $e timeout -k 10 200 python bench_model.py $ARGS > gpurun_out/abm_$n.json 2> gpurun_out/abm_$n.err || { echo "$n failed"; tail -5 gpurun_out/abm_$n.err; exit 1; }
    python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-10s %8.1f slices/s %7.3f ms' % (sys.argv[2], d['value'], d['ms_per_step']))" gpurun_out/abm_$n.json $n
done
done
