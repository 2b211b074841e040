#!/bin/bash
# Counters of the NT projection kernel at the 256-graph row count (tools/gemm_probe.py, NT_ONLY, one shape): one
# rocprofv3 --pmc pass per group, per-kernel averages in gpurun_out/ntpmc/<tag>_g<i>.txt.   tools/nt_pmc.sh TAG SHAPE [M]
set -o pipefail
export TMPDIR=/tmp
tag=$1; export ONLY=${2:-0} M=${3:-1262000} NT_ONLY=1
out=gpurun_out/ntpmc
mkdir -p $out
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  timeout -s KILL 120 rocprofv3 --pmc $group -d $out/${tag}_g$i -o p --output-format csv -- python3 tools/gemm_probe.py > $out/${tag}_g$i.log 2>&1 || { tail -5 $out/${tag}_g$i.log; exit 1; }
  { echo "# rocprofv3 --pmc $group -- python3 tools/gemm_probe.py (M=$M shape $ONLY lib=${HGVAE_LIB:-default}; per-kernel averages; GRBM_GUI_ACTIVE summed over the 8 XCDs)"
    python3 tools/pmc_kernels.py $(find $out/${tag}_g$i -name '*counter_collection.csv' | head -1) k_gemm; } > $out/${tag}_g$i.txt || exit 1
  rm -rf $out/${tag}_g$i
  echo "group $i done"
done
