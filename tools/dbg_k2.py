"""debug helper: the K2/K3 advance_p goldens on the current build, errors printed instead of asserted"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = importlib.import_module("old-vpic_amd")
g = np.load("tests/golden/kernels.npz")
nx, ny, nz = [int(v) for v in g["k1_dims"]]
for case in sys.argv[1:] or ["k2", "k3a", "k3b"]:
    kw = {}
    if case == "k3b":
        kw = dict(fbc=[int(x) for x in g["k3b_fbc"]], pbc=[int(x) for x in g["k3b_pbc"]])
    e = V.Engine(V.make_grid(nx, ny, nz, 6.0, 5.0, 4.0, np.float32(0.3), **kw))
    if os.environ.get("PUSH") == "fast":
        e.set_push_mode("fast")
    e.set_interpolator(g["k2_fi"])
    p_in = g["k2_p_in" if case == "k2" else "k3_p_in"]
    sp = e.new_species(-1.0, len(p_in) + 16, 4096)
    e.set_particles(sp, p_in)
    e.clear_accumulators()
    print(case, "np", len(p_in), flush=True)
    nm = e.advance_p(sp)
    e.sync()
    print(" advance_p done nm", nm, flush=True)
    got, ref = e.get_particles(sp), g[case + "_p_out"]
    bad = sum(int(np.sum(got[c].view(np.uint32) != ref[c].view(np.uint32))) for c in ("dx", "dy", "dz", "i", "ux", "uy", "uz", "q"))
    a, r = e.get_accumulator(), g[case + "_a_out"]
    err = max(np.abs(a[c].astype(np.float64) - r[c]).max() for c in ("jx", "jy", "jz")) / max(np.abs(r[c]).max() for c in ("jx", "jy", "jz"))
    print(f" words differing {bad}, accumulator rel err {err:.3e}", flush=True)
    if os.environ.get("PUSH") == "fast":
        for c in ("ux", "uy", "uz", "dx", "dy", "dz"):
            d = np.abs(got[c].astype(np.float64) - ref[c])
            print("  ", c, "max abs diff", d.max(), "same cell", (got["i"] == ref["i"]).mean())
