#!/bin/bash
# runs every tools/bin/nt16_* variant on the K-concatenated hidden-width shape (and a few others for the base build)
cd "$(dirname "$0")/../.."
for b in tools/bin/nt16_*; do
  timeout -k 5 60 $b 160984 138 138 138 $(basename $b) || exit 1
done
timeout -k 5 60 tools/bin/nt16_base 160984 138 0 138 base || exit 1
timeout -k 5 60 tools/bin/nt16_base 160984 276 0 138 base || exit 1
TN=1 timeout -k 5 60 tools/bin/nt16_base 160984 138 138 138 base || exit 1
TN=1 timeout -k 5 60 tools/bin/nt16_base 160984 138 0 138 base || exit 1
