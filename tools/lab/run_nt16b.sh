#!/bin/bash
cd "$(dirname "$0")/../.."
for b in base noload noload_nolds; do
  timeout -k 5 60 tools/bin/nt16_$b 160984 138 138 138 $b || exit 1
done
for m in 4096 16384 40000 80000 160984 321968 643936; do
  timeout -k 5 60 tools/bin/nt16_base $m 138 138 138 base || exit 1
done
for m in 16384 80000 321968; do
  timeout -k 5 60 tools/bin/nt16_noload_nolds $m 138 138 138 noload_nolds || exit 1
done
