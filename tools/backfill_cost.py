"""What removals cost advance_p between sorts: particles that leave through a face are removed by back-filling from the end
of the array (boundary_p.c:264), which drops the array's last particles into other tiles' ranges.  128^3 x 32 ppc two-stream
with absorbing x faces against the periodic box (also as a 32-cell slab, what each of 8 GPUs holds of 256 cells), one sort, then 12 steps; advance_p time per launch, step by step.
    python tools/backfill_cost.py        (GPU box)"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
V = importlib.import_module("old-vpic_amd")
L = importlib.import_module("old-vpic_amd.layout")


def run(absorbing, nx=128):
    n, ppc = 128, 32
    dt = np.float32(0.95 / np.sqrt(3.0))
    kw = dict(pbc=[L.ABSORB_PARTICLES, 0, 0, L.ABSORB_PARTICLES, 0, 0]) if absorbing else {}
    e = V.Engine(V.make_grid(nx, n, n, float(nx), float(n), float(n), dt, **kw))
    e.set_vacuum()
    e.set_sort_order("engine")
    sps = []
    for k, drift in enumerate((0.2, -0.2)):
        sp = e.new_species(-1.0, nx * n * n * ppc + 4096, nx * n * n * ppc // 8)
        e.load_maxwellian(sp, ppc, 1 + k, -1.0 / ppc, (drift, 0.0, 0.0), 0.02)
        sps.append(sp)
    e.load_interpolator()
    out = []
    for step in range(13):
        e.profile_enable(True)
        e.step(step, 1000)            # sorts at step 0 only
        e.sync()
        ms, launches, parts = e.profile_read()
        out.append(ms / max(launches, 1))
    print("nx=%d" % nx, ("absorbing x" if absorbing else "periodic   "), " ".join("%.2f" % t for t in out), " np", [e.np(sp) for sp in sps])
    e.close()


for nx in (128, 32):
    run(False, nx)
    run(True, nx)
