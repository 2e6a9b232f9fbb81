#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in dpp: lds:gpurun_exp/collds.so; do
  tag=${v%%:*}; lib=${v#*:}
  bash tools/exp/pmc_col.sh ${tag}_a "$lib" "1024,1024,512 14 1 1" SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  bash tools/exp/pmc_col.sh ${tag}_b "$lib" "1024,1024,512 14 1 1" SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
done
