#!/bin/bash
# direct aggregation: 16 lanes x 3 chunks per row (base) against 8 lanes x 5 chunks (cpl5) at F = 138, and other widths
cd "$(dirname "$0")/../.."
for f in 138 128 105 56 256; do
  for v in base cpl5; do
    echo -n "$v F=$f: "; DIRECT_ONLY=1 timeout -k 5 120 tools/bin/agg_lab_$v 160984 $f 0 | tail -1 || exit 1
  done
done
