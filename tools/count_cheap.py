import importlib, os, sys
sys.path.insert(0, os.getcwd())
pt = importlib.import_module('path-tracer_amd')
for name in ["geoms", "teapot", "lamborghini", "sky_map", "lamborghini_door", "blender_comp", "blender_comp_open", "example"]:
    try:
        desc, cam = pt.load_scene(name, 64, 48); desc = desc.contents
    except Exception as e:
        print(name, 'ERR', str(e)[:80]); continue
    tot = sum(desc.objects[i].num_triangles for i in range(desc.num_objects) if desc.objects[i].type != 0 and desc.objects[i].num_nodes == 1)
    print(name, 'objects', desc.num_objects, 'single-leaf triangles', tot)
