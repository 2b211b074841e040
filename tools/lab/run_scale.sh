#!/bin/bash
# NT time against the row count for a single-part weight (N = 80: five n tiles) and the two-part one (N = 138)
cd "$(dirname "$0")/../.."
for m in 160984 643936 1285158; do
  for n in 80 64 138; do
    timeout -k 5 120 tools/bin/nt16_base $m 138 138 $n base || exit 1
  done
done
for m in 160984 1285158; do TN=1 timeout -k 5 120 tools/bin/nt16_base $m 138 138 138 base || exit 1; done
