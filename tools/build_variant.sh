#!/bin/bash
# A second build of the library with extra compiler flags on one source: tools/build_variant.sh NAME file.hip "-DFLAG=1 ..."
# -> hgvae_amd/csrc/variants/libhgvae_hip_NAME.so (select with HGVAE_LIB=...).  The default objects must be built already.
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../hgvae_amd/csrc"
mkdir -p variants
obj=variants/${src%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I../../include $flags -c $src -o $obj
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libhgvae_hip_$name.so $obj $others
rm -f $obj
echo variants/libhgvae_hip_$name.so
