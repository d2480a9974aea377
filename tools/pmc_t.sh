#!/bin/bash
# development aid (GPU box): SQ counters of tools/t_bench variants
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4/pmc
cd /tmp
for b in "$@"; do
  n=$(basename $b)
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/r4/pmc/$n -o out --output-format csv -- $R/$b 262144 > $R/gpurun_out/r4/pmc/$n.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $R/gpurun_out/r4/pmc/${n}_b -o out --output-format csv -- $R/$b 262144 >> $R/gpurun_out/r4/pmc/$n.log 2>&1
done
find $R/gpurun_out/r4/pmc -name "*counter_collection.csv" | head
