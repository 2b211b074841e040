#!/bin/bash
# A/B of environment settings on the headline workload: tools/ab_env.sh "A=1" "A=2 B=3" ...  (each argument one setting, twice around)
mkdir -p gpurun_out/r03
run() { env $1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['loss'])"; }
for round in 1 2; do for v in "$@"; do run "$v" || exit 1; done; done
