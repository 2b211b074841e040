#!/bin/bash
# build_variant.sh <tag> [sed expression ...]: compiles tools/lab/nt16_lab.hip against a sed-made variant of gemm_bf16.hip
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
mkdir -p tools/bin /tmp/nt16lab
src=/tmp/nt16lab/gemm_bf16_$tag.hip
cp hgvae_amd/csrc/gemm_bf16.hip $src
for e in "$@"; do sed -i -E "$e" $src; done
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Ihgvae_amd/csrc -Iinclude -DKSRC="\"$src\"" tools/lab/nt16_lab.hip -o tools/bin/nt16_$tag
echo built tools/bin/nt16_$tag
