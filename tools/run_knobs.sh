for cfg in "0 0" "1 0" "0 128" "1 128" "1 256"; do
  set -- $cfg
  echo "== RNB_HOOK_LATE=$1 RNB_STAGGER=$2"
  RNB_HOOK_LATE=$1 RNB_STAGGER=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
