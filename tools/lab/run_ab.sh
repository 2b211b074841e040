#!/bin/bash
# A/B of two lab builds (tools/bin/nt16_old, tools/bin/nt16_new) on the projection shapes of the benchmark, interleaved
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for shape in "160984 138 138 138" "160984 138 0 138" "160984 105 105 138" "160984 138 0 112" "160984 28 0 138" "40000 138 138 138"; do
  for b in old new; do timeout -k 5 60 tools/bin/nt16_$b $shape $b || exit 1; done
done
done
