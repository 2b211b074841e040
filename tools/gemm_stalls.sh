#!/bin/bash
# Stall / utilisation counters of the projection kernels on the probe shapes (tools/gemm_probe.py, M = 160 984), one
# rocprofv3 --pmc pass per counter group; per-kernel averages land in gpurun_out/gstalls/<group>.txt.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/gstalls
rm -rf $out; mkdir -p $out
i=0
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout -s KILL 120 rocprofv3 --pmc $group -d $out/g$i -o p --output-format csv -- python3 tools/gemm_probe.py > $out/g$i.log 2>&1 || { tail -5 $out/g$i.log; exit 1; }
  { echo "# rocprofv3 --pmc $group -- python3 tools/gemm_probe.py (default projection arithmetic; per-kernel averages; GRBM_GUI_ACTIVE summed over the 8 XCDs)"
    python3 tools/pmc_kernels.py $(find $out/g$i -name '*counter_collection.csv' | head -1) k_gemm; } > $out/g$i.txt || exit 1
  rm -rf $out/g$i
  echo "group $i done"
done
