#!/bin/bash
# development aid (GPU box, repo root): A/B of library builds under tools/_libs/lib_<tag>.so — same box, alternating runs.
# usage: tools/ab.sh "<bench args>" tagA tagB ...   (two rounds each; prints ms/step and the two largest kernel classes)
args=$1; shift
cp rnb-neus-fork_amd/librnbneus_hip.so /tmp/lib_orig.so
for round in 1 2; do
  for v in "$@"; do
    cp tools/_libs/lib_$v.so rnb-neus-fork_amd/librnbneus_hip.so || exit 1
    echo -n "== $v: "
    timeout -k 10 180 python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 8 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline'] or {}; print(d['ms_per_step'], d.get('ms_per_step_median'), ' '.join(k['kernel_class'].split('(')[0][:8] + ('s' if 'save' in k['kernel_class'] else '') + '=%.3f' % k['ms_per_step'] for k in r.get('by_kernel_class',[])))" || exit 1
  done
done
cp /tmp/lib_orig.so rnb-neus-fork_amd/librnbneus_hip.so
