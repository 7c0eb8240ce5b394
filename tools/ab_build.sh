# Builds an A/B copy of libptx_hip.so with extra compiler flags: bash tools/ab_build.sh <name> "<flags>"  ->  ab/<name>/libptx_hip.so
# (on the GPU box: cp ab/<name>/libptx_hip.so path-tracer_amd/ before a run; the box's tree is scratch)
set -e
cd "$(dirname "$0")/../path-tracer_amd"
mkdir -p ../ab/$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value $2 -shared csrc/ptx_hip.hip -o ../ab/$1/libptx_hip.so 2>&1 | grep -E "error" || true
ls -la ../ab/$1/libptx_hip.so
