for v in 1 2 3 4 6 8; do
  echo -n "prefetch_group=$v "; PTX_TRAV_VARIANT=$v timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*'
done
