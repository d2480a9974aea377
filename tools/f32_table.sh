#!/bin/bash
# development aid: kernel table of the fp32 step under rocprofv3 (run on the GPU box from the repo root)
set -e
out=${1:-gpurun_out/f32p}; shift || true
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o f32 -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-gemm-events "$@" > $out/bench.log 2>&1
grep '^{' $out/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms", d["ms_per_step"])'
python tools/kstats.py $out/f32_kernel_stats.csv 25 12
