# A/B of the one-sample-per-call loop (tools/reference_loop_rate.py) and the bench step: bash tools/ab_loop.sh name1 name2 ...
for n in "$@"; do
  cp ab/$n/libptx_hip.so path-tracer_amd/libptx_hip.so || exit 1
  echo "== $n"
  timeout -k 10 150 python tools/reference_loop_rate.py --modes ${MODES:-loop,render1} --iters 200 2>/dev/null | grep "ms per call" || exit 1
done
