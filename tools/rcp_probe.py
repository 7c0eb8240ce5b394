"""Where does pt_rcp_newton equal 1.0f/a?  usage: python tools/rcp_probe.py"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("path-tracer_amd")
L, H = pt.lib(), pt.host()
ctx = H.pth_context()
for lo, hi in ((5e-8, 2.0**126), (1e-38, 3.4e38), (2.0**-126, 2.0**126), (2.0**126, 3.4e38), (1e-45, 2.0**-126), (5e-8, 1.0), (1.0, 2.0**126)):
    c, m, f = C.c_uint64(), C.c_uint64(), C.c_uint32()
    pt.check(L.ptx_selftest_rcp(ctx, lo, hi, C.byref(c), C.byref(m), C.byref(f)))
    print(f"|a| in [{lo:g}, {hi:g}]: checked {c.value} mismatches {m.value} first {f.value:#010x}", flush=True)
