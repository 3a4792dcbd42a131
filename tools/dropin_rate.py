"""PCIe-inclusive rate of the drop-in twin of advance_p (include/vpic_hip_dropin.h: the reference's own signature,
host arrays in, host arrays out): what a caller that keeps its particles in host memory gets per call.
    python tools/dropin_rate.py [cells_per_side] [ppc]        (GPU box)"""
import ctypes as C
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
D = importlib.import_module("old-vpic_amd.dropin")
L = importlib.import_module("old-vpic_amd.layout")


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lib = D.ref()
    dt = np.float32(0.95 / np.sqrt(3.0))
    g = D.reference_grid(n, n, n, float(n), float(n), float(n), dt)
    nv = (n + 2) ** 3
    np_ = n * n * n * ppc
    rng = np.random.default_rng(1)
    p = np.zeros(np_, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, np_).astype(np.float32)
    cell = np.repeat(np.arange(n * n * n, dtype=np.int64), ppc)
    cz, r = np.divmod(cell, n * n)
    cy, cx = np.divmod(r, n)
    p["i"] = ((cx + 1) + (n + 2) * ((cy + 1) + (n + 2) * (cz + 1))).astype(np.int32)
    p["ux"] = (0.2 + 0.02 * rng.standard_normal(np_)).astype(np.float32)
    p["uy"] = (0.02 * rng.standard_normal(np_)).astype(np.float32)
    p["uz"] = (0.02 * rng.standard_normal(np_)).astype(np.float32)
    p["q"] = -1.0 / ppc
    fi = np.zeros(nv, L.interpolator_t)
    a = np.zeros(nv, L.accumulator_t)
    pm = np.zeros(max(np_ // 8, 4096), L.particle_mover_t)
    times = []
    for rep in range(4):
        lib.vpic_hip_ref_clear_accumulators(P(a), C.byref(g))
        t0 = time.perf_counter()
        nm = lib.vpic_hip_ref_advance_p(P(p), np_, -1.0, P(pm), len(pm), P(a), P(fi), C.byref(g))
        times.append(time.perf_counter() - t0)
    t = min(times[1:])
    moved = 2 * p.nbytes + 2 * a.nbytes + fi.nbytes
    print("advance_p twin, host arrays: %d^3 x %d ppc = %.1f M particles, %.1f ms per call = %.1f M pushes/s "
          "(%.2f GB over PCIe per call, %.1f GB/s), movers %d" % (n, ppc, np_ / 1e6, t * 1e3, np_ / t / 1e6, moved / 1e9, moved / t / 1e9, nm))


main()
