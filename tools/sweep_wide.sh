# A/B of the wide traversal kernel's configurations (PTX_WIDE_CFG) and stack caps (PTX_WIDE_SCAP) on the bench workload
# usage (GPU box): bash tools/sweep_wide.sh "0 1 2" "12 16"
for c in $1; do for s in ${2:-12}; do
  echo -n "cfg=$c scap=$s "; PTX_WIDE_CFG=$c PTX_WIDE_SCAP=$s timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'Mrays/s  step', r['step_kernel_ms'], 'ms  segment', r['segment_kernel']['launch_ms'], 'x', r['segment_kernel']['launches'])" || exit 1
done; done
