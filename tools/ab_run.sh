# A/B on the GPU box: for each name, ab/<name>/libptx_hip.so replaces the library of the (scratch) tree and the bench runs
# usage: bash tools/ab_run.sh name1 name2 ...
for n in "$@"; do
  cp ab/$n/libptx_hip.so path-tracer_amd/libptx_hip.so || exit 1
  echo -n "$n: "; timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg --no-loop-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['other_kernel'] or {}
ks={r['kernel'][:12]:r, (o.get('kernel') or 'none')[:12]:o}
print(d['value'], 'Mrays/s  step', r['step_kernel_ms'], 'ms;', ' '.join('%s %.3f ms x %d' % (k, v.get('launch_ms',0), v.get('launches',0)) for k,v in ks.items()))" || exit 1
done
