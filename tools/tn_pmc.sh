#!/bin/bash
# Counters of the weight-gradient kernel at the 256-graph row count (tools/gemm_probe.py, TN_ONLY): tools/tn_pmc.sh TAG SHAPE
set -o pipefail
export TMPDIR=/tmp
tag=$1; export ONLY=${2:-0} M=${3:-1262000} TN_ONLY=1
out=gpurun_out/tnpmc
mkdir -p $out
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  timeout -s KILL 120 rocprofv3 --pmc $group -d $out/${tag}_g$i -o p --output-format csv -- python3 tools/gemm_probe.py > $out/${tag}_g$i.log 2>&1 || { tail -5 $out/${tag}_g$i.log; exit 1; }
  { echo "# rocprofv3 --pmc $group -- python3 tools/gemm_probe.py (M=$M shape $ONLY TN_ONLY; per-kernel averages; GRBM_GUI_ACTIVE summed over the 8 XCDs)"
    python3 tools/pmc_kernels.py $(find $out/${tag}_g$i -name '*counter_collection.csv' | head -1) k_gemm_tn16; } > $out/${tag}_g$i.txt || exit 1
  rm -rf $out/${tag}_g$i
done
cat $out/${tag}_g1.txt $out/${tag}_g2.txt | grep -v "^#"
