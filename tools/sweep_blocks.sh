for b in 5 6 8 16 32; do
  echo -n "blocks/CU=$b "; PTX_TRAV_BLOCKS=$b timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-loop-leg 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s  traverse', d['roofline']['launch_ms'], 'ms')" || exit 1
done
