# BASELINE.json configs C2..C5 through bench.py (C4 is the headline); prints value, step time and the traversal's share
run() { echo -n "$1 ${2}x$3: "; timeout -k 10 280 python bench.py --scene $1 --width $2 --height $3 --textures $4 --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg --no-loop-leg 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r.get('other_kernel') or {}
tv = r if r['kernel'].startswith('wf_traverse') else o
print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', 'samples/s', d['config']['samples_per_s'], 'trav share', tv.get('share_of_step', 0), 'segment ms', r['segment_kernel']['launch_ms'])" || exit 1; }
run geoms 1024 768 0 && run teapot 1920 1080 0 && run lamborghini 1920 1080 1 && run sky_map 3840 2160 1
