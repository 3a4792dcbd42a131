"""BASELINE.json configs[1] at full size (128^3, both 67 M-particle beams of the two-stream deck at 32 ppc), ONE SLAB of
configs[2] at its full per-GPU size (256^3 over 8 x-slabs = 32 x 256 x 256 cells, 2 species x 64 ppc = 2 x 134 M
particles; every face wraps onto the slab itself, so the single domain needs no neighbour) and ONE SLAB of configs[3] at
its full per-GPU size (decks/trecon-part scaled to 256 x 256 x 128 over 8 x-slabs = 32 x 256 x 128 cells; 4 species x
64 ppc: the pair plasma of turbulence.cxx:95-98 -- mi/me = 1, vthe = vthi = 0.6 c -- and its two charge-0 tracer
copies, tracer.cxx:64-70; conducting walls that reflect particles in z, turbulence.cxx:265-269) through
size-independent properties, plus a bit-exact spot check of a random particle sample against the CPU oracle.
GPU box only."""
import importlib

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
CASES = {"configs1_128cubed_32ppc": (128, 128, 128, 32, 2), "configs2_slab_32x256x256_64ppc": (32, 256, 256, 64, 2),
         "configs3_slab_32x256x128_4species_64ppc": (32, 256, 128, 64, 4)}


@pytest.fixture(scope="module", params=list(CASES))
def run(request, orc, L):
    NX, NY, NZ, PPC, NSP = CASES[request.param]
    V = importlib.import_module("old-vpic_amd")
    dt = np.float32(0.95 / np.sqrt(3.0))
    trecon = NSP == 4
    kw, okw = {}, {}
    if trecon:                                              # turbulence.cxx:265-269
        kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
        okw = dict(kw)
    e = V.Engine(V.make_grid(NX, NY, NZ, float(NX), float(NY), float(NZ), dt, **kw))
    e.set_vacuum()
    if trecon:
        e.set_sort_order("engine")                          # what the deck host and the multi-GPU driver run: tile order
    q = -float((0.2 / float(dt)) ** 2 / (2 * PPC))
    q_m = -1.0
    if trecon:
        # tracer copies of both populations (charge 0), the positrons-for-ions of the pair plasma, the electrons (checked) last
        for k, (qm, qq) in enumerate(((-1.0, 0.0), (1.0, 0.0), (1.0, -q), (-1.0, q))):
            sp = e.new_species(qm, NX * NY * NZ * PPC, NX * NY * NZ * PPC // 16)
            e.load_maxwellian(sp, PPC, 1 + k, qq, (0.0, 0.0, 0.0), 0.6)
    for k in range(0 if trecon else NSP):                   # the second beam of the two-stream deck (drift -0.2 c) first,
        u = (0.2, 0.0, 0.0) if k == NSP - 1 else (-0.2, 0.0, 0.0)   # the checked species last
        sp = e.new_species(-1.0, NX * NY * NZ * PPC, 4096)
        e.load_maxwellian(sp, PPC, 1 + k, q, u, 0.02)
    # a smooth non-trivial field so that the push is not a free flight
    nv = e.nv
    rng = np.random.default_rng(3)
    f = np.zeros(nv, L.field_t)
    idx = np.arange(nv)
    x, y, z = idx % (NX + 2), (idx // (NX + 2)) % (NY + 2), idx // ((NX + 2) * (NY + 2))
    for c, (a, b) in {"ex": (0.02, 3), "ey": (0.015, 5), "ez": (0.01, 7), "cbx": (0.03, 2), "cby": (0.02, 4), "cbz": (0.025, 6)}.items():
        f[c] = (a * np.sin(2 * np.pi * b * (x + 2 * y + 3 * z) / 128)).astype(np.float32)
    e.set_fields(f)
    e.load_interpolator()
    for step in range(3):                       # a few steps so that the array is no longer exactly sorted
        e.step(step, 0 if not trecon else -20)  # (the hot deck under the engine's own sort policy, in tile order)
    before = e.get_particles(sp)
    fi = e.get_interpolator()
    e.clear_accumulators()
    nm = e.advance_p(sp)
    after = e.get_particles(sp)
    acc = e.get_accumulator()
    yield dict(V=V, e=e, sp=sp, before=before, after=after, acc=acc, fi=fi, nm=nm, dt=dt, dims=(NX, NY, NZ), okw=okw, q_m=q_m)
    e.close()


def test_sample_is_bit_exact_against_the_oracle(run, orc, L):
    rng = np.random.default_rng(11)
    pick = np.unique(rng.integers(0, len(run["before"]), 200000))
    p = run["before"][pick].copy()
    NX, NY, NZ = run["dims"]
    g = orc.make_grid(NX, NY, NZ, float(NX), float(NY), float(NZ), run["dt"], **run["okw"])
    a = np.zeros(g.nv, L.accumulator_t)
    pm = np.zeros(64, L.particle_mover_t)
    assert orc.advance_p(p, len(p), run["q_m"], pm, a, run["fi"], g) == 0
    assert bits_equal(p, run["after"][pick])
    assert run["nm"] == 0


def test_deposited_current_equals_particle_displacement(run):
    """Summing a cell's four quarter-face entries of one component gives 4 q (half displacement);
    over all cells and streaks: sum(jx[0..3]) = 2 q * (total x displacement in cell units of 2)."""
    b, a, acc = run["before"], run["after"], run["acc"]
    NX, NY, NZ = run["dims"]
    sy, sz = NX + 2, (NX + 2) * (NY + 2)
    def coords(v):
        v = v.astype(np.int64)
        return v % sy, (v // sy) % (NY + 2), v // sz
    cb, ca = coords(b["i"]), coords(a["i"])
    for axis, (comp, d) in enumerate((("jx", "dx"), ("jy", "dy"), ("jz", "dz"))):
        hop, N = ca[axis] - cb[axis], run["dims"][axis]
        hop = np.where(hop > N // 2, hop - N, np.where(hop < -N // 2, hop + N, hop))   # periodic wrap
        disp = (a[d].astype(np.float64) - b[d].astype(np.float64)) + 2.0 * hop
        expect = 2.0 * np.sum(b["q"].astype(np.float64) * disp)
        got = acc[comp].astype(np.float64).sum()
        assert abs(got - expect) <= 2e-5 * abs(expect) + 1e-6, (comp, got, expect)


def test_sort_properties(run):
    e, sp = run["e"], run["sp"]
    p0 = run["after"]
    e.set_sort_order("reference")                           # partition[] belongs to the reference's order (sort_p.c:32)
    e.sort_p(sp)
    p1 = e.get_particles(sp)
    part = e.get_partition(sp)
    assert np.all(np.diff(p1["i"]) >= 0)                                      # sorted
    counts = np.bincount(p0["i"], minlength=e.nv)
    assert np.array_equal(np.diff(part), counts) and part[-1] == len(p0)      # partition = prefix of counts
    # same multiset: order-independent checksums over the raw bits of every field
    for n in ("dx", "dy", "dz", "i", "ux", "uy", "uz", "q"):
        v0, v1 = p0[n].view(np.uint32).astype(np.uint64), p1[n].view(np.uint32).astype(np.uint64)
        assert v0.sum() == v1.sum() and (v0 * v0 % 1000003).sum() == (v1 * v1 % 1000003).sum(), n
    e.sort_p(sp)                                                              # idempotent up to order within a voxel
    p2 = e.get_particles(sp)
    assert np.array_equal(p2["i"], p1["i"])


def test_species_beyond_two_to_the_thirty_particles(orc, L):
    """BASELINE.json configs[4] per GPU (256^3 x 512 ppc over 8 GPUs = 2^30 particles each, before any head room):
    128^3 cells x 520 ppc = 1.09e9 particles of a cold uniform drift in ONE species, pushed in two launch segments.
    Checked: the particles around the segment boundary and at the end of the array bit for bit against the oracle,
    the deposited current against its closed form (every particle makes the same move), the particle count."""
    import importlib
    V = importlib.import_module("old-vpic_amd")
    N, PPC = 128, 520
    dt = np.float32(0.95 / np.sqrt(3.0))
    e = V.Engine(V.make_grid(N, N, N, float(N), float(N), float(N), dt))
    e.set_vacuum()
    n = N ** 3 * PPC
    assert n > 2 ** 30
    sp = e.new_species(-1.0, n, 4096)
    q = -1e-3
    u = (0.1, 0.05, 0.02)
    e.load_maxwellian(sp, PPC, 7, q, u, 0.0)
    e.load_interpolator()                                   # zero fields: a free flight
    iters = int(0.9 * (62 - 8) * PPC / 256)
    iters = min(max(iters, 1), 64)
    seg = 2 ** 30 // (256 * iters) * (256 * iters)
    spots = [(0, 4096), (seg - 4096, 8192), (n - 4096, 4096)]
    before = [e.get_particles_range(sp, a, c) for a, c in spots]
    e.clear_accumulators()
    assert e.advance_p(sp) == 0 and e.np(sp) == n
    g = orc.make_grid(N, N, N, float(N), float(N), float(N), dt)
    fi = np.zeros(g.nv, L.interpolator_t)
    for (a, c), p0 in zip(spots, before):
        p = p0.copy()
        acc = np.zeros(g.nv, L.accumulator_t)
        pm = np.zeros(64, L.particle_mover_t)
        assert orc.advance_p(p, len(p), -1.0, pm, acc, fi, g) == 0
        assert bits_equal(p, e.get_particles_range(sp, a, c)), a
    # closed form: gamma = sqrt(1 + u.u); every particle is displaced by 2 u c dt / (gamma dx) cell units (cells span 2)
    gam = np.sqrt(1.0 + sum(x * x for x in u))
    acc = e.get_accumulator()
    for comp, ux in zip(("jx", "jy", "jz"), u):
        expect = 2.0 * q * n * 2.0 * (ux * float(dt) / gam)
        got = acc[comp].astype(np.float64).sum()
        assert abs(got - expect) <= 2e-4 * abs(expect), (comp, got, expect)
    e.close()
