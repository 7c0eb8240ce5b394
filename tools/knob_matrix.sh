# The GPU suite under other DEFAULTS of the scheduling knobs (environment, read at ptx_ctx_create): every test must pass
# whatever schedule the library starts with. Usage (GPU box, from the repo root): bash tools/knob_matrix.sh
for cfg in "PTX_SEG_COMPACT=0" "PTX_SEG_MODE=0" "PTX_TRAV_VARIANT=15" "PTX_WF_BATCH=7" "PTX_WIDE_BUILD=2 PTX_GRP_MODE=0" "PTX_LOOK_AHEAD=1" "PTX_LOOK_AHEAD=128 PTX_TRAV_BLOCKS=1 PTX_SEG_BLOCKS=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not production_rays and not two_million" 2>&1 | tail -1 || exit 1
done
