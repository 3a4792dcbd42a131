#!/usr/bin/env python3
"""What parts of advance_p cost, measured on ONE launch per part: a deck is stepped normally (every part on), then a single
advance_p of every species runs with some parts switched off and is timed with the engine's own events -- the state the
launch starts from is the real one, which a whole run with parts off cannot offer (a run without the crossers' stores
re-crosses the same particles every step).  Needs a -DVPIC_HIP_ABLATION build: VPIC_HIP_LIB=tools/ab/libablation.so.
    python tools/ablate_once.py [--deck two-stream|trecon] [--grid nx ny nz] [--ppc n] [--steps-before n] bits [bits ...]
bits (push.hip): 1 no in-cell deposit, 2 no crossing path, 4 no interpolator gather, 8 no flush, 16 no regrouping,
32 no mover deposit, 64 no drain, 128 no in-cell stores, 256 no stores of the crossers' final positions, 512 one of the four."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--deck", default="two-stream")
    ap.add_argument("--grid", type=int, nargs=3, default=None)
    ap.add_argument("--ppc", type=int, default=64)
    ap.add_argument("--steps-before", type=int, default=6)
    ap.add_argument("bits", type=int, nargs="+")
    a = ap.parse_args()
    V = importlib.import_module("old-vpic_amd")
    L = V.layout
    lib = V.lib()
    assert hasattr(lib, "vpic_hip_debug_set_ablate"), "not an ablation build (VPIC_HIP_LIB=tools/ab/libablation.so)"
    trecon = a.deck == "trecon"
    nx, ny, nz = a.grid or ((32, 256, 128) if trecon else (256, 256, 256))
    dt = np.float32(0.95 / np.sqrt(3.0))
    q = -float((0.2 / float(dt)) ** 2 / (2 * a.ppc))
    kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES]) if trecon else {}
    for bits in a.bits:
        e = V.Engine(V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), dt, **kw))
        e.set_vacuum()
        e.set_sort_order("engine")
        n = nx * ny * nz * a.ppc
        sps = []
        if trecon:
            for k, (qm, qq) in enumerate(((-1.0, q), (1.0, -q), (1.0, 0.0), (-1.0, 0.0))):
                sp = e.new_species(qm, n, n // 16)
                e.load_maxwellian(sp, a.ppc, 1 + k, qq, (0.0, 0.0, 0.0), 0.6)
                sps.append(sp)
        else:
            for k, u in enumerate(((0.2, 0.0, 0.0), (-0.2, 0.0, 0.0))):
                sp = e.new_species(-1.0, n, n // 16)
                e.load_maxwellian(sp, a.ppc, 1 + k, q, u, 0.02)
                sps.append(sp)
        e.load_interpolator()
        for step in range(a.steps_before):
            e.step(step, 10 if not trecon else 3)
        lib.vpic_hip_debug_set_ablate(e._h, bits)
        e.profile_enable(True)
        e.clear_accumulators()
        for sp in sps:
            e.advance_p(sp)
        per = ["%.3f" % (e.profile_read_species(sp)[0]) for sp in sps]
        ms, launches, parts = e.profile_read()
        print("ABLATE=%-4d one launch per species after %d ordinary steps: %s ms  (mean %.3f ms)" % (bits, a.steps_before, " ".join(per), ms / max(launches, 1)), flush=True)
        e.close()


if __name__ == "__main__":
    main()
