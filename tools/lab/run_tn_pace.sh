#!/bin/bash
# weight-gradient kernel 138 x (138+138): the two k groups of a row slice unpaced on different XCDs ("never") or paced on one
# XCD ("paced"), on dense operands and with three quarters of G zero (a gradient behind relu and dropout)
cd "$(dirname "$0")/../.."
for sp in 0 0.75; do
  for m in 160984 643936 1285158; do
    SPARSE=$sp TN=1 HGVAE_TN_PACE_MIN_MB=100000000 timeout -k 5 120 tools/bin/nt16_base $m 138 138 138 never_sp$sp || exit 1
    SPARSE=$sp TN=1 HGVAE_TN_PACE_MIN_MB=0 timeout -k 5 120 tools/bin/nt16_base $m 138 138 138 paced_sp$sp || exit 1
  done
done
