"""400 steps of a 48^3 two-stream deck, sort every 10 steps: the engine's own choice between the sort inside the push and sort +
push (measured per species: engine.hip, sort_and_push) against sort + push throughout (VPIC_HIP_SORT_IN_PUSH=0, read when the
engine is created).  Energies every 20 steps, particle counts, how many pushes sorted."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
V = importlib.import_module("old-vpic_amd")
def run(fuse, vth, steps=400):
    if fuse: os.environ.pop("VPIC_HIP_SORT_IN_PUSH", None)
    else: os.environ["VPIC_HIP_SORT_IN_PUSH"] = "0"
    n = 48
    dt = np.float32(0.95 / np.sqrt(3.0))
    e = V.Engine(V.make_grid(n, n, n, float(n), float(n), float(n), dt))
    e.set_vacuum()
    e.set_sort_order("engine")
    e.profile_enable(True)
    sps = []
    for k, drift in enumerate((0.2, -0.2)):
        sp = e.new_species(-1.0, n ** 3 * 40, 4096)
        e.load_maxwellian(sp, 32, 1 + k, -1.0 / 64, (drift, 0.0, 0.0), vth)
        sps.append(sp)
    e.load_interpolator()
    en = []
    for step in range(steps):
        e.step(step, 10)
        if step % 20 == 19: en.append(list(e.energy_f()) + [e.energy_p(sp) for sp in sps])
    nps = [e.np(sp) for sp in sps]
    sorting = e.profile_read_sorting()[1]
    e.close()
    return np.array(en), nps, sorting
for vth in (0.02, 0.15):
    a, na, sa = run(False, vth)
    b, nb, sb = run(True, vth)
    ke = lambda x: x[:, 6:].sum(1)
    fe = lambda x: x[:, :6].sum(1)
    print("vth", vth, "particles", na, nb, "sorting pushes", sa, sb)
    print("  kinetic rel diff: %.2e   field rel diff: %.2e" % (np.abs(ke(b) / ke(a) - 1).max(), np.abs(fe(b) / fe(a) - 1).max()))
    print("  total energy drift (sort + push, sort inside the push): %.3e %.3e" % ((ke(a) + fe(a))[-1] / (ke(a) + fe(a))[0] - 1, (ke(b) + fe(b))[-1] / (ke(b) + fe(b))[0] - 1))
    assert na == nb and sa == 0 and sb > 0
