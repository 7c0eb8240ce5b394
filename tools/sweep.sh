# A/B of traversal kernel variants (PTX_TRAV_VARIANT) on the bench workload
for v in "$@"; do
  echo -n "variant=$v "; PTX_TRAV_VARIANT=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-loop-leg 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s  traverse', d['roofline']['launch_ms'], 'ms  segment', d['roofline']['segment_kernel']['launch_ms'], 'ms')" || exit 1
done
