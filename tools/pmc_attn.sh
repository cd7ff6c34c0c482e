#!/bin/bash
# PMC passes over the head-dim-64 attention kernels at one shape (run on the GPU box): where the waves' cycles go.
#   bash tools/pmc_attn.sh OUTDIR B H Lq Lk
set -e
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p$i -- python3 tools/attn_one.py "$@" > /dev/null 2> $OUT/p$i.err || { echo "pass $i failed"; tail -3 $OUT/p$i.err; continue; }
  python3 tools/pmc_summary.py $(ls $OUT/p$i/*.db | head -1) $OUT/pmc_$i.csv || true
  rm -rf $OUT/p$i
  echo "pass $i done"
done
cat $OUT/pmc_*.csv
