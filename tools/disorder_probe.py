"""Disorder fraction (vpic_hip_measure_disorder) and advance_p time per step since the last sort, for the
two-stream and the hot 4-species deck: calibration of the adaptive-sorting threshold."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
V = importlib.import_module("old-vpic_amd")
L = importlib.import_module("old-vpic_amd.layout")
class A: pass
for deckname in ("two-stream", "sheet"):
    a = A(); a.ppc = 32; a.grid = None; a.sort_interval = 0; a.deck = deckname
    d = bench.deck(a, 1)
    kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES]) if deckname == "sheet" else {}
    e = V.Engine(V.make_grid(d["gx"], d["gy"], d["gz"], float(d["gx"]), float(d["gy"]), float(d["gz"]), d["dt"], **kw))
    e.set_vacuum()
    n_sp = d["gx"] * d["gy"] * d["gz"] * d["ppc"]
    sps = []
    if deckname == "sheet":
        for k, (q_m, sgn, u, vth) in enumerate(d["species4"]):
            sp = e.new_species(q_m, n_sp, max(n_sp // 16, 1024)); e.load_maxwellian(sp, d["ppc"], 1 + k, sgn * abs(d["q"]), u, vth); sps.append(sp)
    else:
        for k, u in enumerate(d["species"]):
            sp = e.new_species(-1.0, n_sp, max(n_sp // 16, 1024)); e.load_maxwellian(sp, d["ppc"], 1 + k, d["q"], u, d["vth"]); sps.append(sp)
    e.load_interpolator()
    for step in range(3): e.step(step, 1)           # settle, sorted every step
    e.profile_enable(True)
    for step in range(3, 16):
        e.step(step, 0)                               # no sorting from here on
        e.sync()
        ms, launches, pushed = e.profile_read()
        e.profile_enable(True)
        print(deckname, "steps since sort %2d" % (step - 2), "push ms/launch %.3f" % (ms / max(launches, 1)),
              "disorder:", " ".join("%.4f" % e.measure_disorder(sp) for sp in sps))
    e.close()
