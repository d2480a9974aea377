for cfg in "0" "64" "128" "256" "512"; do
  echo "== RNB_STAGGER=$cfg"
  RNB_STAGGER=$cfg timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
