#!/bin/bash
# two-part NT with and without the paced walk of the two parts (and the allowed lead in load groups), against the row count
cd "$(dirname "$0")/../.."
for m in 160984 241476 321968 643936 1285158; do
  for v in never always; do
    timeout -k 5 120 tools/bin/nt16_$v $m 138 138 138 $v || exit 1
  done
done
