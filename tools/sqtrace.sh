#!/bin/bash
# SQ counters per kernel of the default bench step (GPU box, repo root): usage tools/sqtrace.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/r5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out -o sq_$tag -- python bench.py --no-cpu-baseline --no-also --steps 4 --warmup 2 --no-gemm-events "$@" > $out/sq_$tag.log 2>&1
python tools/sq_summary.py $out/sq_${tag}_counter_collection.csv $out/sq_${tag}_kernel_trace.csv > $out/sq_counters_$tag.txt
cat $out/sq_counters_$tag.txt
