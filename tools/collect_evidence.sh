#!/bin/bash
# Evidence run on the GPU box (from the repo root): bench line, rocprofv3 kernel stats and the two PMC traffic passes
# for one bench configuration.  usage: tools/collect_evidence.sh <tag> [bench.py args...]
set -e
tag=$1; shift
out=gpurun_out/ev_$tag
mkdir -p $out
python bench.py "$@" > $out/bench.json 2> $out/bench.err
tail -c 400 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-gemm-events "$@" > $out/ks.log 2>&1
python tools/kstats.py $out/ks_kernel_stats.csv 25 30 > $out/kernel_table.txt
head -8 $out/kernel_table.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 --no-gemm-events "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 --no-gemm-events "$@" > $out/write.log 2>&1
echo traffic passes done
