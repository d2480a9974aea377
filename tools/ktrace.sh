#!/bin/bash
# Kernel table of the default bench step under rocprofv3 (from the repo root on the GPU box): usage tools/ktrace.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/r5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks_$tag -- python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --no-gemm-events "$@" > $out/ks_$tag.log 2>&1
python tools/kstats.py $out/ks_${tag}_kernel_stats.csv 25 45 > $out/kernel_table_$tag.txt
head -${KT_LINES:-14} $out/kernel_table_$tag.txt
