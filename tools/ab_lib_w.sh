#!/bin/bash
# A/B of library builds on another workload: tools/ab_lib_w.sh WORKLOAD [--graph] -- NAME...  ("base" = default build)
w=$1; shift; extra=""
while [ "$1" != "--" ]; do extra="$extra $1"; shift; done; shift
run() {
  local lib=""; [ "$1" != base ] && lib="$PWD/hgvae_amd/csrc/variants/libhgvae_hip_$1.so"
  HGVAE_LIB=$lib timeout -k 10 200 python bench.py --workload $w $extra --steps 40 --warmup 10 --headline-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w$extra $1', d['ms_per_step'], d['loss'])"
}
for round in 1 2; do for v in "$@"; do run $v || exit 1; done; done
