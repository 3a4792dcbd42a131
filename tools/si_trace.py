#!/usr/bin/env python3
"""A fixed sort interval watched step by step: python tools/si_trace.py <interval> <steps> [every step from] -- ms per step and the species' stats."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = importlib.import_module("old-vpic_amd")
si, steps = int(sys.argv[1]), int(sys.argv[2])
every_from = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
N, ppc = 256, 64
dt = np.float32(0.95 / np.sqrt(3.0))
e = V.Engine(V.make_grid(N, N, N, float(N), float(N), float(N), dt))
e.set_vacuum(); e.set_sort_order("engine")
q = -float((0.2 / float(dt)) ** 2 / (2 * ppc))
for k, u in enumerate(((0.2, 0, 0), (-0.2, 0, 0))):
    sp = e.new_species(-1.0, N ** 3 * ppc, N ** 3 * ppc // 16)
    e.load_maxwellian(sp, ppc, 1 + k, q, u, 0.02)
e.load_interpolator()
for step in range(steps):
    t0 = time.perf_counter(); e.step(step, si); e.sync(); ms = (time.perf_counter() - t0) * 1e3
    if step % 5 == 4 or ms > 60 or step >= every_from:
        print("step %3d %7.2f ms  %s" % (step, ms, e.species_stats(0)), flush=True)
