#!/bin/bash
# development aid (GPU box, repo root): raw per-kernel means of an arbitrary --pmc counter set for one library build.
# usage: tools/pmc_raw.sh <lib tag under tools/_libs | TREE> <out tag> COUNTER...   (prints kernel x counter means of the rnb:: MFMA kernels)
lib=$1; tag=$2; shift 2
out=gpurun_out/r5
mkdir -p $out
if [ "$lib" != TREE ]; then cp rnb-neus-fork_amd/librnbneus_hip.so /tmp/lib_orig_pmc.so && cp tools/_libs/lib_$lib.so rnb-neus-fork_amd/librnbneus_hip.so || exit 1; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -o pmc_$tag -- python bench.py --no-cpu-baseline --no-also --steps 4 --warmup 2 --no-gemm-events $BENCH_ARGS > $out/pmc_$tag.log 2>&1
rc=$?
if [ "$lib" != TREE ]; then cp /tmp/lib_orig_pmc.so rnb-neus-fork_amd/librnbneus_hip.so; fi
[ $rc -eq 0 ] || { tail -5 $out/pmc_$tag.log; exit $rc; }
python - $out/pmc_${tag}_counter_collection.csv <<'PY'
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[re.sub(r"\(.*", "", r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "rnb::" in k and any(t in k for t in ("fused_", "color_", "gemm_dw_x3")):
        print(f"{k[:58]:60s} " + "  ".join(f"{c} {sum(x) / len(x):.4g}" for c, x in sorted(v.items())))
PY
