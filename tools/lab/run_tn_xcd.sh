#!/bin/bash
# weight-gradient kernel: the two k groups of a row slice on different XCDs (base) or on the same one (tnx)
cd "$(dirname "$0")/../.."
for m in 160984 643936 1285158; do
  for v in base tnx; do
    TN=1 timeout -k 5 120 tools/bin/nt16_$v $m 138 138 138 $v || exit 1
  done
done
