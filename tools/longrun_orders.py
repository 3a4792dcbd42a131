import importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
V = importlib.import_module("old-vpic_amd")
def run(mode, vth, steps=400):
    if mode == "reference": os.environ["VPIC_HIP_WINDOW"] = "wide"
    else: os.environ.pop("VPIC_HIP_WINDOW", None)
    n = 48
    dt = np.float32(0.95 / np.sqrt(3.0))
    e = V.Engine(V.make_grid(n, n, n, float(n), float(n), float(n), dt))
    e.set_vacuum()
    if mode != "reference": e.set_sort_order("engine")
    sps = []
    for k, drift in enumerate((0.2, -0.2)):
        sp = e.new_species(-1.0, n ** 3 * 40, 4096)
        e.load_maxwellian(sp, 32, 1 + k, -1.0 / 64, (drift, 0.0, 0.0), vth)
        sps.append(sp)
    e.load_interpolator()
    en = []
    for step in range(steps):
        e.step(step, -20 if mode == "adaptive" else 10)
        if step % 20 == 19: en.append(list(e.energy_f()) + [e.energy_p(sp) for sp in sps])
    orders = [e.species_order(sp) for sp in sps]
    e.close()
    return np.array(en), orders
for vth in (0.02, 0.3):
    a, oa = run("reference", vth)
    b, ob = run("adaptive", vth)
    c, oc = run("fixed", vth)
    ke = lambda x: x[:, 6:].sum(1)
    fe = lambda x: x[:, :6].sum(1)
    print("vth", vth, "orders", oa, ob, oc)
    print("  kinetic rel diff adaptive/fixed vs reference: %.2e %.2e" % (np.abs(ke(b)/ke(a)-1).max(), np.abs(ke(c)/ke(a)-1).max()))
    print("  field   rel diff adaptive/fixed vs reference: %.2e %.2e" % (np.abs(fe(b)/fe(a)-1).max(), np.abs(fe(c)/fe(a)-1).max()))
    print("  total energy drift (reference, adaptive): %.3e %.3e" % ((ke(a)+fe(a))[-1]/(ke(a)+fe(a))[0]-1, (ke(b)+fe(b))[-1]/(ke(b)+fe(b))[0]-1))
