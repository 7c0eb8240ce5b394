# Per-launch times of the last bench step for A/B libraries: bash tools/ab_launches.sh name1 name2 ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in "$@"; do
  cp ab/$n/libptx_hip.so path-tracer_amd/libptx_hip.so || exit 1
  rm -rf gpurun_out/abl_$n
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg --no-loop-leg > /dev/null 2>&1 || exit 1
  echo "$n: $(python3 tools/launch_times.py gpurun_out/abl_$n)"
done
