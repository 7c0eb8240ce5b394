for b in 8 16 32 64; do
  echo -n "batch=$b "; PTX_WF_BATCH=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-loop-leg --spp-per-step $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s  traverse', d['roofline']['launch_ms'], 'ms  step', d['ms_per_step'])" || exit 1
done
