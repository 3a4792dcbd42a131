"""Debug build only (make CXXEXTRA=-DVPIC_HIP_DEBUG_COUNTERS): crossers / drain passes / window misses per step."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = importlib.import_module("old-vpic_amd")
n, ppc = 64, 32
dt = np.float32(0.95 / np.sqrt(3.0))
e = V.Engine(V.make_grid(n, n, n, float(n), float(n), float(n), dt))
e.set_vacuum()
q = -float((0.2 / float(dt)) ** 2 / (2 * ppc))
for k, s in enumerate((1.0, -1.0)):
    sp = e.new_species(-1.0, n ** 3 * ppc, 1024)
    e.load_maxwellian(sp, ppc, 1 + k, q, (s * 0.2, 0.0, 0.0), 0.02)
e.load_interpolator()
l = V.lib()
out = (C.c_int * 8)()
l.vpic_hip_debug_counters(out, 1)
npart = 2 * n ** 3 * ppc
for step in range(16):
    e.step(step, 0)
    l.vpic_hip_debug_counters(out, 1)
    c = list(out)
    print(f"step {step:2d} crossers {c[0]/npart:.4f}/particle  passes {c[1]}  ({c[0]/max(c[1],1):.1f} per pass)  loop iters/pass {c[3]/max(c[1],1):.2f}  window misses {c[2]} ({c[2]/npart:.5f}/particle)  run tails {c[4]/npart:.4f}/particle  miss beyond {c[5]} before {c[6]} other {c[7]}")
