"""Deterministic accumulation (include/vpic_hip.h: vpic_hip_set_accumulation; push_device.h, Window<4>).  Deposits are
rounded to 64-bit fixed point and summed as integers, so nothing that is summed depends on the order of the particle
array, of the wavefronts or of the atomics: two runs agree BIT FOR BIT -- what the reference gets from private
accumulators reduced in a fixed order (sf_interface/reduce_accumulators.cxx:37-55).  Per-particle results are those of
the default mode (bit-exact against the goldens); the sums agree with the reference's float sums within the usual
summation-order tolerance.  GPU box only."""
import importlib
import os

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
ACC_TOL = 2e-6


@pytest.fixture(scope="module")
def V():
    v = importlib.import_module("old-vpic_amd")
    assert v.lib().vpic_hip_device_count() > 0, "no HIP device"
    return v


@pytest.mark.parametrize("case", ["k2", "k3a", "k3b"])
def test_goldens_in_deterministic_mode(V, golden, case):
    """K2 / K3 of the reference's goldens with fixed-point sums: particles and movers bit for bit, accumulators within the
    tolerance of the float mode (they are the exactly rounded sums of the same terms)."""
    nx, ny, nz = [int(v) for v in golden["k1_dims"]]
    kw = {}
    if case == "k3b":
        kw = dict(fbc=[int(x) for x in golden["k3b_fbc"]], pbc=[int(x) for x in golden["k3b_pbc"]])
    e = V.Engine(V.make_grid(nx, ny, nz, 6.0, 5.0, 4.0, np.float32(0.3), **kw))
    e.set_accumulation("deterministic")
    e.set_interpolator(golden["k2_fi"])
    p_in = golden["k2_p_in" if case == "k2" else "k3_p_in"]
    sp = e.new_species(-1.0, len(p_in) + 16, 4096)
    e.set_particles(sp, p_in)
    e.clear_accumulators()
    nm = e.advance_p(sp)
    assert bits_equal(e.get_particles(sp), golden[case + "_p_out"])
    a, ref = e.get_accumulator(), golden[case + "_a_out"]
    A = np.stack([a["jx"], a["jy"], a["jz"]]).astype(np.float64)
    R = np.stack([ref["jx"], ref["jy"], ref["jz"]]).astype(np.float64)
    assert np.abs(A - R).max() <= ACC_TOL * np.abs(R).max()
    if case == "k3b":
        assert nm == len(golden["k3b_pm"]) and bits_equal(e.get_movers(sp), golden["k3b_pm"])


def _deck(V, L, order_seed, nsteps, sort_order, mode, walls=False):
    """A 16 x 12 x 8 two-stream box, 2 species x 24 ppc, full steps; the particle arrays are handed to the engine in a random
    order drawn from order_seed.  Returns the fields and the particles (by tag) after nsteps."""
    nx, ny, nz, ppc = 16, 12, 8, 24
    dt = np.float32(0.95 / np.sqrt(3.0))
    kw = {}
    if walls:
        kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    e = V.Engine(V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), dt, **kw))
    e.set_vacuum()
    e.set_accumulation(mode)
    e.set_sort_order(sort_order)
    rng = np.random.default_rng(5)
    n = nx * ny * nz * ppc
    sps = []
    for k, drift in enumerate((0.2, -0.2)):
        p = np.zeros(n, L.particle_t)
        for c in ("dx", "dy", "dz"):
            p[c] = rng.uniform(-1, 1, n).astype(np.float32)
        cell = np.repeat(np.arange(nx * ny * nz), ppc)
        p["i"] = L.voxel(cell % nx + 1, (cell // nx) % ny + 1, cell // (nx * ny) + 1, nx, ny, nz)
        p["ux"] = (drift + 0.1 * rng.standard_normal(n)).astype(np.float32)
        p["uy"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        p["uz"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        p["q"] = (-0.004 * rng.uniform(0.5, 1.5, n)).astype(np.float32)
        p["tag"] = np.arange(n) + 1 + k * n
        p = p[np.random.default_rng(order_seed + k).permutation(n)]       # the SAME particles, another array order
        sp = e.new_species(-1.0, n + 64, n // 2)
        e.set_particles(sp, p)
        sps.append(sp)
    e.load_interpolator()
    for step in range(nsteps):
        e.step(step, 4)
    f = e.get_fields()
    ps = []
    for sp in sps:
        p = e.get_particles(sp)
        ps.append(p[np.argsort(p["tag"], kind="stable")])
    e.close()
    return f, ps


@pytest.mark.parametrize("sort_order,walls", [("engine", False), ("reference", False), ("engine", True)])
def test_runs_agree_bit_for_bit_whatever_the_array_order(V, L, sort_order, walls):
    """Two runs of 12 full steps whose particle arrays start in different random orders (so every wavefront, every run of
    equal cells, every atomic sees other neighbours): fields and particles identical to the last bit.  In tile order (LDS
    window in 64-bit words) and in the reference's order (every deposit a global 64-bit atomic); with reflecting walls."""
    fa, pa = _deck(V, L, 100, 12, sort_order, "deterministic", walls)
    fb, pb = _deck(V, L, 200, 12, sort_order, "deterministic", walls)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz", "jfx", "jfy", "jfz"):
        assert np.array_equal(fa[c].view(np.uint32), fb[c].view(np.uint32)), c
    for a, b in zip(pa, pb):
        assert bits_equal(a, b)
    assert np.abs(fa["ex"]).max() > 0


def test_float_mode_is_close_to_the_deterministic_one(V, L):
    """The default float sums against the fixed-point ones on the same deck: fields within 2e-5 of the field scale after
    12 steps (the float mode's own run-to-run spread is of that order)."""
    fa, _ = _deck(V, L, 100, 12, "engine", "deterministic")
    fb, _ = _deck(V, L, 100, 12, "engine", "float")
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        scale = max(np.abs(fa[k]).max() for k in (("ex", "ey", "ez") if c[0] == "e" else ("cbx", "cby", "cbz")))
        assert np.abs(fa[c].astype(np.float64) - fb[c]).max() <= 2e-5 * scale, c
