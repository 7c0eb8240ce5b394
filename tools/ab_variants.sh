# A/B of libraries on the bench step under a traversal variant: PTX_TRAV_VARIANT=15 bash tools/ab_variants.sh name1 name2 ...
for n in "$@"; do
  cp ab/$n/libptx_hip.so path-tracer_amd/libptx_hip.so || exit 1
  echo -n "$n: "; timeout -k 10 250 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg --no-loop-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s  step', d['ms_per_step'], 'ms')" || exit 1
done
