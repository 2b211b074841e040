#!/bin/bash
# Build here: hipcc --offload-arch=gfx950 -O3 -o tools/lab/nt_mem_lab tools/lab/nt_mem_lab.hip ; run on the GPU box:
timeout -k 10 120 ./tools/lab/nt_mem_lab ${1:-1262000}
