#!/bin/bash
# A/B on the headline workload: shared first level written out / deferred, twice around.
mkdir -p gpurun_out/r03
run() { env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['loss'])"; }
run HGVAE_DEFER_PAIR=0 || exit 1
run HGVAE_DEFER_PAIR=1 || exit 1
run HGVAE_DEFER_PAIR=0 || exit 1
run HGVAE_DEFER_PAIR=1 || exit 1
