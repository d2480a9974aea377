#!/bin/bash
# Evidence of one round on the GPU box (from the repo root): usage tools/collect_round.sh <tag, e.g. r05>
# Writes everything under gpurun_out/ev_<tag>/ ; the summaries are then copied into profiles/ by hand.
# Order: the PMC passes first (their summaries go into THIS copy's profiles/, carrying the library's build id), then the bench legs — so that
# the bench lines of the same call quote `roofline.traffic` / `mfma_busy_pmc` of the very build they measure.
set -e
tag=$1
out=gpurun_out/ev_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-also --steps 4 --warmup 2 --no-gemm-events"
echo "== PMC passes"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- $B > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- $B > $out/write.log 2>&1
python tools/traffic_summary.py $out/fetch_counter_collection.csv $out/write_counter_collection.csv 6 512 128 hbm_traffic.json > /dev/null && cp profiles/hbm_traffic.json $out/hbm_traffic.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetchbf -- $B --dtype bf16 --samples 256 > $out/fetchbf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o writebf -- $B --dtype bf16 --samples 256 > $out/writebf.log 2>&1
python tools/traffic_summary.py $out/fetchbf_counter_collection.csv $out/writebf_counter_collection.csv 6 512 256 hbm_traffic_bf16.json > /dev/null && cp profiles/hbm_traffic_bf16.json $out/hbm_traffic_bf16.json
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out -o sq -- $B > $out/sq.log 2>&1
python tools/sq_summary.py $out/sq_counter_collection.csv $out/sq_kernel_trace.csv sq_counters.json > $out/sq_counters.txt && cp profiles/sq_counters.json $out/sq_counters.json
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $out -o valu -- $B > $out/valu.log 2>&1 || true
python tools/valu_summary.py $out/valu_counter_collection.csv > $out/valu_counters.txt 2>/dev/null || true
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks -- python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --no-gemm-events > $out/ks.log 2>&1
python tools/kstats.py $out/ks_kernel_stats.csv 25 45 > $out/kernel_table_f32.txt
head -14 $out/kernel_table_f32.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ksbf -- python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --no-gemm-events --dtype bf16 --samples 256 > $out/ksbf.log 2>&1
python tools/kstats.py $out/ksbf_kernel_stats.csv 25 30 > $out/kernel_table_bf16.txt
echo "== bench legs"
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 600 python bench.py "$@" > $out/$name.json 2> $out/$name.err; tail -c 200 $out/$name.json; echo; }
run bench_f32 --steps 50 --warmup 10
run bench_no_x2h --steps 50 --warmup 10 --no-x2h --no-cpu-baseline --no-also
run bench_render --mode render --steps 50 --warmup 10
run bench_mesh512 --mode mesh --resolution 512
run bench_bf16 --dtype bf16 --samples 256
run bench_device_rays --device-rays --no-cpu-baseline
run bench_device_rays_cfg5 --device-rays --stack 200x1024x1024 --no-cpu-baseline
run bench_noalbedo --no-albedo --no-cpu-baseline --no-also
run bench_deterministic --deterministic --no-cpu-baseline --no-also
run bench_dp2_rehearsal --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline
run bench_dp2_strong_rehearsal --gpus 2 --scaling strong --global-rays 1024 --steps 10 --warmup 3 --no-cpu-baseline
ls $out | head -60
echo done
