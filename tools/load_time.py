"""Load time of a scene through the host layer (parse, OBJ / procedural mesh, texture decode, KD build) and of the packed
upload (ptx_scene_upload: leaf groups, both wide-tree candidates, copies).  Usage: python tools/load_time.py [scene] [w h]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pt = importlib.import_module("path-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "lamborghini"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
t = time.perf_counter(); pt.load_scene(name, width=w, height=h, upload=False); t_host = time.perf_counter() - t
print(f"{name}: host load {t_host:.2f} s", flush=True)
if pt.gpu_available():
    pt.host().pth_set_device(0); pt.host().pth_context()
    t = time.perf_counter(); pt.load_scene(name, width=w, height=h, upload=True); t_all = time.perf_counter() - t
    print(f"{name}: host load + upload {t_all:.2f} s (upload {t_all - t_host:.2f} s, {pt.lib().ptx_scene_device_bytes(pt.host().pth_scene_device()) / 1e6:.0f} MB on the device)")
