#!/bin/bash
# A/B of library builds on the headline workload: tools/ab_lib.sh NAME... (variants built by tools/build_variant.sh; "base" = default)
mkdir -p gpurun_out/r03
run() {
  local lib=""; [ "$1" != base ] && lib="$PWD/hgvae_amd/csrc/variants/libhgvae_hip_$1.so"
  HGVAE_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['loss'])"
}
for round in 1 2; do for v in "$@"; do run $v || exit 1; done; done
