import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = importlib.import_module("old-vpic_amd")
g = np.load("tests/golden/kernels.npz")
nx, ny, nz = [int(v) for v in g["k1_dims"]]
e = V.Engine(V.make_grid(nx, ny, nz, 6.0, 5.0, 4.0, np.float32(0.3)))
e.set_interpolator(g["k2_fi"])
p_in = g["k2_p_in"]
sp = e.new_species(-1.0, len(p_in) + 16, 4096)
e.set_particles(sp, p_in)
e.clear_accumulators()
print("nm", e.advance_p(sp))
out, ref = e.get_particles(sp), g["k2_p_out"]
for n in out.dtype.names:
    bad = np.nonzero(out[n] != ref[n])[0]
    print(n, len(bad), bad[:10], out[n][bad[:4]], ref[n][bad[:4]])
