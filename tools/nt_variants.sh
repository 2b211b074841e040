#!/bin/bash
# NT projection at the 256-graph row count, library variants side by side: tools/nt_variants.sh SHAPE NAME...   ("base" = default build)
s=$1; shift
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$PWD/hgvae_amd/csrc/variants/libhgvae_hip_$v.so"
  echo -n "$v: "; HGVAE_LIB=$lib M=${M:-1262000} NT_ONLY=1 ONLY=$s timeout -k 10 120 python tools/gemm_probe.py 2>/dev/null | cut -d'|' -f1 || exit 1
done
