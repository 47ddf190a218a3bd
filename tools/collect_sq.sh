#!/bin/bash
# Developer tool (GPU box): SQ counters of the hot kernels over a short bench run -- MFMA pipe busy cycles and the wave
# stall breakdown (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles).  Separate passes, --pmc only (no trace domains).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="${@:---steps 2 --warmup 1 --no-cpu-baseline --no-extras}"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '+' | cut -c1-60)
  rocprofv3 --pmc $set --output-format csv -d $O/$tag -o pmc -- python3 $R/bench.py $ARGS > $O/$tag.log 2>&1; echo "pmc [$set] rc=$?"
done
