#!/bin/bash
# Collect PMC counters for the conv kernels, one rocprofv3 pass per counter group (no tracing combined with --pmc).
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <outdir> <python script + args...>
set -u
OUT=$1; shift
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "${DYF_PMC_REGEX:-conv_igemm|conv_up_halo|up2x|stem|readout}" --output-format csv -d "$R/$OUT/p$i" -o p$i -- python "$R/$@" > "$R/$OUT/p$i.log" 2>&1
  echo "pass $i rc=$? : $grp"
done
