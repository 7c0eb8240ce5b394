"""Phase shares of the segment kernel by the shader clock (counting variant, PtxStats.seg_clk), per bounce limit.
Usage (GPU box): python tools/segment_clock.py [spp]"""
import sys, os, importlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", 1920, 1080, textures=1)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
acc, st = gpuops.render(ctx, scene, cam, 0, spp, 0, stats=True)
d = st.as_dict()
c = [d["seg_clk%d" % k] for k in range(4)]
tot = float(sum(c)) or 1.0
print("segment kernel clock shares: generate/shade/compact %.1f%%  barriers %.1f%%  cheap objects + state write %.1f%%  queue appends %.1f%%" %
      tuple(100 * v / tot for v in c))
