#!/bin/bash
# development aid (GPU box): same-box A/B of bench ARGUMENT sets with the in-tree library: usage tools/args_ab.sh "<args A>" "<args B>" ...
for round in 1 2; do
  for a in "$@"; do
    echo -n "== [$a]: "
    timeout -k 10 180 python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 8 $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline'] or {}; print(d['ms_per_step'], d.get('ms_per_step_median'), ' '.join(k['kernel_class'].split('(')[0][:8] + ('s' if 'save' in k['kernel_class'] else '') + '=%.3f' % k['ms_per_step'] for k in r.get('by_kernel_class',[])))" || exit 1
  done
done
