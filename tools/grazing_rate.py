"""How often the default traversal deviates from the oracle on ADVERSARIAL rays (tools/grazing_stress.py's generator: rays
within a hair of triangle edges at incidences of 1e-6 ... 1e-1 rad), at a larger sample than the 2 000 000 of the fence test
in tests/test_gpu_wide.py: ten independent sets of 2 000 000 on the headline mesh, default traversal (0) and strict kernel (15) by default; GRAZE_VARIANTS=0,13,15,9,7 compares more, GRAZE_SAVE=file.npz keeps the rays variant 0 differs on.
Usage (GPU box): python tools/grazing_rate.py [sets] [rays per set]"""
import sys, os, importlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
pt = importlib.import_module("path-tracer_amd")
import gpuops, orc
from grazing_stress import grazing_rays

sets = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", 64, 48)
scene = pt.host().pth_scene_device()
d = desc.contents
obj = max(range(d.num_objects), key=lambda i: d.objects[i].num_nodes)
VARIANTS = tuple(int(v) for v in os.environ.get('GRAZE_VARIANTS', '0,15').split(','))
tot = {v: 0 for v in VARIANTS}
rays = 0
hard = []
for s in range(sets):
    rng = np.random.default_rng(1000 + s)
    o, dr, tgt = grazing_rays(desc, obj, n, rng)
    ho = orc.trace(desc, o, dr, obj)
    rays += len(o)
    line = f"set {s}: {len(o)} rays, oracle hits {(ho['object'] == obj).sum()}"
    for variant in VARIANTS:
        pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
        hg = gpuops.trace_queued(ctx, scene, o, dr, obj)
        bad = (ho["object"] != hg["object"]) | (ho["prim"] != hg["prim"]) | (ho["t_bits"] != hg["t_bits"])
        tot[variant] += int(bad.sum())
        line += f"; variant {variant}: {int(bad.sum())} differ"
        if variant == 0 and bad.any():
            idx = np.nonzero(bad)[0]
            hard.append({"o": o[idx], "d": dr[idx], **{k: ho[k][idx] for k in ("hit", "object", "prim", "t_bits", "bary")}})
    print(line, flush=True)
pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))
print(f"TOTAL {rays} adversarial rays: differing rays per traversal variant {tot}")
if hard and os.environ.get("GRAZE_SAVE"):
    out = {k: np.concatenate([h[k] for h in hard]) for k in hard[0]}
    np.savez(os.environ["GRAZE_SAVE"], **out)
    print("saved", len(out["o"]), "hard rays to", os.environ["GRAZE_SAVE"])

