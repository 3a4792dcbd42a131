"""What particles appended behind a tile-sorted species cost advance_p (arrivals from a neighbour domain pile up there
until the next sort): 128^3 x 32 ppc two-stream, then `frac` of the species appended in the two x boundary planes.
    python tools/tail_cost.py [frac ...]        (GPU box)"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
V = importlib.import_module("old-vpic_amd")
L = importlib.import_module("old-vpic_amd.layout")


def main():
    n, ppc = 128, 32
    fracs = [float(a) for a in sys.argv[1:]] or [0.0, 0.01, 0.03]
    dt = np.float32(0.95 / np.sqrt(3.0))
    for frac in fracs:
        e = V.Engine(V.make_grid(n, n, n, float(n), float(n), float(n), dt))
        e.set_vacuum()
        e.set_sort_order("engine")
        npart = n ** 3 * ppc
        extra = int(frac * npart)
        sp = e.new_species(-1.0, npart + extra + 4096, 1 << 20)
        e.load_maxwellian(sp, ppc, 1, -1.0 / ppc, (0.2, 0.0, 0.0), 0.02)
        e.load_interpolator()
        e.sort_p(sp)
        if extra:
            rng = np.random.default_rng(1)
            p = np.zeros(extra, L.particle_t)
            for c in ("dx", "dy", "dz"):
                p[c] = rng.uniform(-1, 1, extra).astype(np.float32)
            x = np.where(rng.random(extra) < 0.5, 1, n)
            p["i"] = L.voxel(x, rng.integers(1, n + 1, extra), rng.integers(1, n + 1, extra), n, n, n)
            p["ux"] = 0.2
            p["q"] = -1.0 / ppc
            e.append_particles(sp, p)
        e.profile_enable(True)
        for _ in range(3):
            e.clear_accumulators()
            e.advance_p(sp)
        e.sync()
        ms, launches, parts = e.profile_read()
        print("appended %.1f %% (%d particles): advance_p %.3f ms per launch" % (100 * frac, extra, ms / launches))
        e.close()


main()
