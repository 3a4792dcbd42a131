"""Parity of the HIP engine (through the C ABI, libvpic_hip.so) with the reference-generated
golden vectors and with the CPU oracle on seeded inputs.  GPU box only.

Tolerances.  Everything that is computed per particle or per voxel without a reduction is
BIT-EXACT (fp contraction is off, sqrt/divide correctly rounded): particle states, interpolator
coefficients, advance_b/advance_e/unload results, mover lists.  Quantities that sum many particles
(accumulators, and the fields/energies downstream of them) are summed in a different order than the
CPU loop, so they carry fp32 round-off: ACC_TOL below, stated relative to the largest accumulator
entry.  The reference's own builds differ from each other by 1.6e-5 (scalar vs SSE) to 2e-4
(1 vs 4 ranks) in energy after 50 steps."""
import importlib

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
ACC_TOL = 2e-6      # |a_hip - a_ref| <= ACC_TOL * max|a_ref|  (single accumulation pass)


@pytest.fixture(scope="module")
def V():
    v = importlib.import_module("old-vpic_amd")
    assert v.lib().vpic_hip_device_count() > 0, "no HIP device"
    return v


def k1_grid(V, golden, **kw):
    nx, ny, nz = [int(v) for v in golden["k1_dims"]]
    return V.make_grid(nx, ny, nz, 6.0, 5.0, 4.0, np.float32(0.3), **kw)


def acc_close(a, ref):
    a = np.stack([a["jx"], a["jy"], a["jz"]]).astype(np.float64)
    r = np.stack([ref["jx"], ref["jy"], ref["jz"]]).astype(np.float64)
    scale = np.abs(r).max()
    err = np.abs(a - r).max()
    assert err <= ACC_TOL * scale, f"accumulator error {err:.3e} vs scale {scale:.3e}"


def test_k1_load_interpolator(V, golden):
    e = V.Engine(k1_grid(V, golden))
    e.set_fields(golden["k1_f"])
    e.load_interpolator()
    assert bits_equal(e.get_interpolator(), golden["k1_fi"])
    # the AoS mirror round-trips bit for bit
    assert bits_equal(e.get_fields(), golden["k1_f"])


@pytest.mark.parametrize("case", ["k2", "k3a", "k3b"])
def test_advance_p(V, golden, case):
    kw = {}
    if case == "k3b":
        kw = dict(fbc=[int(x) for x in golden["k3b_fbc"]], pbc=[int(x) for x in golden["k3b_pbc"]])
    e = V.Engine(k1_grid(V, golden, **kw))
    e.set_interpolator(golden["k2_fi"])
    p_in = golden["k2_p_in" if case == "k2" else "k3_p_in"]
    sp = e.new_species(-1.0, len(p_in) + 16, 4096)
    e.set_particles(sp, p_in)
    e.clear_accumulators()
    nm = e.advance_p(sp)
    assert bits_equal(e.get_particles(sp), golden[case + "_p_out"])
    acc_close(e.get_accumulator(), golden[case + "_a_out"])
    if case == "k3b":
        assert nm == len(golden["k3b_pm"])
        assert bits_equal(e.get_movers(sp), golden["k3b_pm"])
    else:
        assert nm == 0


@pytest.mark.parametrize("case", ["k2", "k3a", "k3b"])
def test_advance_p_chargeless_species(V, golden, case):
    """A species whose particles all have q == 0 (tracer copies) runs the advance_p instance without any
    deposition: particle states and movers are bit-identical to the charged run (the golden outputs, q
    aside), the accumulator stays untouched."""
    kw = {}
    if case == "k3b":
        kw = dict(fbc=[int(x) for x in golden["k3b_fbc"]], pbc=[int(x) for x in golden["k3b_pbc"]])
    e = V.Engine(k1_grid(V, golden, **kw))
    e.set_interpolator(golden["k2_fi"])
    p_in = golden["k2_p_in" if case == "k2" else "k3_p_in"].copy()
    p_in["q"] = 0
    sp = e.new_species(-1.0, len(p_in) + 16, 4096)
    e.set_particles(sp, p_in)
    e.clear_accumulators()
    nm = e.advance_p(sp)
    want = golden[case + "_p_out"].copy()
    want["q"] = 0
    assert bits_equal(e.get_particles(sp), want)
    a = e.get_accumulator()
    assert not np.any(a["jx"]) and not np.any(a["jy"]) and not np.any(a["jz"])
    if case == "k3b":
        assert nm == len(golden["k3b_pm"]) and bits_equal(e.get_movers(sp), golden["k3b_pm"])


def test_advance_p_sorted_cells_many_per_cell(V, orc, L):
    """Cell-sorted input with ~40 particles per cell: the wavefront-grouped LDS deposit path."""
    rng = np.random.default_rng(7)
    nx, ny, nz = 12, 6, 5
    g = V.make_grid(nx, ny, nz, 12.0, 6.0, 5.0, np.float32(0.4))
    og = orc.make_grid(nx, ny, nz, 12.0, 6.0, 5.0, np.float32(0.4))
    n = nx * ny * nz * 40
    p = np.zeros(n, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    x, y, z = rng.integers(1, nx + 1, n), rng.integers(1, ny + 1, n), rng.integers(1, nz + 1, n)
    p["i"] = np.sort(L.voxel(x, y, z, nx, ny, nz))
    for c, d in (("ux", 0.2), ("uy", 0.0), ("uz", 0.0)):
        p[c] = (rng.standard_normal(n) * 0.05 + d).astype(np.float32)
    p["q"] = -0.01
    f = np.zeros(og.nv, L.field_t)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        f[c] = (rng.standard_normal(og.nv) * 0.05).astype(np.float32)
    fi = np.zeros(og.nv, L.interpolator_t)
    orc.load_interpolator(fi, f, og)
    ref_p, ref_a = p.copy(), np.zeros(og.nv, L.accumulator_t)
    pm = np.zeros(16, L.particle_mover_t)
    assert orc.advance_p(ref_p, n, -1.0, pm, ref_a, fi, og) == 0
    e = V.Engine(g)
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, n, 1024)
    e.set_particles(sp, p)
    e.clear_accumulators()
    assert e.advance_p(sp) == 0
    assert bits_equal(e.get_particles(sp), ref_p)
    acc_close(e.get_accumulator(), ref_a)


def test_k4_unload_and_sync_jf(V, golden):
    e = V.Engine(k1_grid(V, golden))
    e.set_fields(golden["k4_f_in"])
    e.set_accumulator(golden["k4_a"])
    e.clear_jf()
    e.unload_accumulator()
    assert bits_equal(e.get_fields(), golden["k4_f_unloaded"])
    e.synchronize_jf()
    assert bits_equal(e.get_fields(), golden["k4_f_synced"])
    # the two calls fused into one pass (what the step driver runs): the same bits
    e.set_fields(golden["k4_f_in"])
    e.clear_jf_unload_accumulator()
    assert bits_equal(e.get_fields(), golden["k4_f_unloaded"])


@pytest.mark.parametrize("tiles", ["0", "2"])
def test_k5_advance_b_e_energy_f(V, golden, tiles, monkeypatch):
    monkeypatch.setenv("VPIC_HIP_FIELD_TILES", tiles)           # 2: advance_b / advance_e through LDS tiles whatever the grid's size
    e = V.Engine(k1_grid(V, golden))
    e.set_vacuum()
    e.set_fields(golden["k5_f_in"])
    e.advance_b(0.5)
    assert bits_equal(e.get_fields(), golden["k5_f_b"])
    e.advance_e()
    f = e.get_fields()
    # ghost values equal as numbers (the sign of a zero ghost may differ: 1*x + 0*y vs copy)
    for n in f.dtype.names:
        assert np.array_equal(f[n], golden["k5_f_e"][n]), n
    np.testing.assert_allclose(e.energy_f(), golden["k6_energy_f"], rtol=1e-12)


@pytest.mark.parametrize("tiles", ["0", "2"])
def test_k5d_damped_pec_z(V, golden, L, tiles, monkeypatch):
    monkeypatch.setenv("VPIC_HIP_FIELD_TILES", tiles)
    fbc = [0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS]
    pbc = [0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES]
    e = V.Engine(k1_grid(V, golden, damp=0.01, fbc=fbc, pbc=pbc))
    e.set_vacuum()
    e.set_fields(golden["k5_f_in"])
    e.advance_b(0.5)
    e.advance_e()
    e.advance_b(0.5)
    f = e.get_fields()
    for n in f.dtype.names:
        assert np.array_equal(f[n], golden["k5d_f_out"][n]), n
    e.set_fields(golden["k4_f_unloaded"])
    e.synchronize_jf()
    f = e.get_fields()
    for n in f.dtype.names:
        assert np.array_equal(f[n], golden["k5d_f_jf_synced"][n]), n


def test_k6_energy_p(V, golden):
    e = V.Engine(k1_grid(V, golden))
    e.set_interpolator(golden["k2_fi"])
    sp = e.new_species(-1.0, len(golden["k2_p_in"]), 64)
    e.set_particles(sp, golden["k2_p_in"])
    assert e.energy_p(sp) == pytest.approx(float(golden["k6_energy_p"]), rel=1e-12)


def canon(p):
    return p[np.lexsort((p["tag"], p["i"]))]


def test_k7_sort_p(V, golden):
    e = V.Engine(k1_grid(V, golden))
    p = golden["k7_p_in"]
    sp = e.new_species(-1.0, len(p), 64)
    e.set_particles(sp, p)
    e.sort_p(sp)
    out = e.get_particles(sp)
    assert np.all(np.diff(out["i"]) >= 0)
    assert np.array_equal(e.get_partition(sp), golden["k7_partition"])
    assert bits_equal(canon(out), canon(golden["k7_p_oop"]))   # same particles in every voxel (tags included)
    e.sort_p(sp)                                               # idempotent
    assert bits_equal(canon(e.get_particles(sp)), canon(out))


def test_empty_and_ragged(V, golden, L):
    """np = 0, np = 1 and np not a multiple of the workgroup chunk."""
    e = V.Engine(k1_grid(V, golden))
    e.set_interpolator(golden["k2_fi"])
    sp = e.new_species(-1.0, 5000, 64)
    e.clear_accumulators()
    assert e.advance_p(sp) == 0 and e.np(sp) == 0
    e.sort_p(sp)
    assert e.energy_p(sp) == 0.0
    for n in (1, 63, 65, 2049):
        e.set_particles(sp, golden["k2_p_in"][:n])
        e.clear_accumulators()
        assert e.advance_p(sp) == 0
        assert bits_equal(e.get_particles(sp), golden["k2_p_out"][:n])


def test_append_particles(V, golden, L):
    """vpic_hip_species_append_particles: particles added at the end of a species while the run is under way
    (inject_particle from a deck hook) -- the same state as uploading the whole list at once, tags included;
    a charge-0 species stops being one when a charged particle arrives."""
    e = V.Engine(k1_grid(V, golden))
    e.set_interpolator(golden["k2_fi"])
    p = golden["k2_p_in"].copy()
    p["tag"] = np.arange(len(p)) + 7
    sp = e.new_species(-1.0, len(p) + 16, 64)
    e.set_particles(sp, p[:1000])
    e.append_particles(sp, p[1000:1001])
    e.append_particles(sp, p[1001:])
    assert e.np(sp) == len(p) and bits_equal(e.get_particles(sp), p)
    e.clear_accumulators()
    assert e.advance_p(sp) == 0
    want = golden["k2_p_out"].copy(); want["tag"] = p["tag"]
    assert bits_equal(e.get_particles(sp), want)
    acc_close(e.get_accumulator(), golden["k2_a_out"])
    z = p[:500].copy(); z["q"] = 0
    sp2 = e.new_species(-1.0, 2000, 64)
    e.set_particles(sp2, z)                                  # charge-0 species: the no-deposit instance ...
    e.append_particles(sp2, p[500:1000])                     # ... until charged particles join
    e.clear_accumulators(); e.advance_p(sp2)
    a = e.get_accumulator()
    assert np.abs(a["jx"]).max() > 0
    with pytest.raises(V.VpicHipError):
        e.append_particles(sp2, p[:1501])                    # beyond max_np
    bad = p[:4].copy(); bad["i"][2] = 0                      # a ghost voxel: refused before it can send a kernel out of bounds
    with pytest.raises(V.VpicHipError):
        e.append_particles(sp2, bad)
    bad["i"][2] = 10 ** 8
    with pytest.raises(V.VpicHipError):
        e.set_particles(sp2, bad)


def test_accumulate_rhob(V, orc, golden, L):
    """vpic_hip_accumulate_rhob (boundary_p.c:9-71; inject_particle with update_rhob): particles on faces, edges and
    corners of a conducting box included (the doubled weights there), against the oracle; float-atomic order only."""
    kw = dict(fbc=[L.PEC_FIELDS] * 6, pbc=[L.REFLECT_PARTICLES] * 6)
    e = V.Engine(k1_grid(V, golden, **kw))
    og = k1_grid(orc, golden, **kw)
    e.set_vacuum()
    p = golden["k3_p_in"][:800].copy()
    f = golden["k11_f_in"].copy()
    e.set_fields(f)
    e.accumulate_rhob(p, -1.0)
    ref = f.copy()
    q = p.copy(); q["q"] = -q["q"]
    orc.accumulate_rhob(ref, q, og)
    got = e.get_fields()
    assert np.abs(got["rhob"].astype(np.float64) - ref["rhob"]).max() <= ACC_TOL * np.abs(ref["rhob"]).max()
    for n in got.dtype.names:
        if n != "rhob":
            assert np.array_equal(got[n], f[n]), n


def test_maxwellian_reflux_boundary(V, L):
    """vpic_hip_set_maxwellian_reflux (src/boundary/maxwellian_reflux.c:116-175): a cold beam runs into the +x wall
    of an empty box; every particle must come back, none may be lost, with the momentum distribution of the flux
    of a Maxwellian at the wall: normal component -sqrt(2) ut_para sqrt(-log U) (mean sqrt(pi/2) ut_para, second
    moment 2 ut_para^2), tangential components N(0, ut_perp).  Statistical: 40 k particles, 4-sigma bounds."""
    nx, ny, nz = 32, 4, 4
    code = -3
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5),
                    fbc=[L.PEC_FIELDS, 0, 0, L.PEC_FIELDS, 0, 0], pbc=[code, 0, 0, code, 0, 0])
    e = V.Engine(g)
    e.set_vacuum()
    n = 40000
    rng = np.random.default_rng(3)
    p = np.zeros(n, L.particle_t)
    x = rng.integers(25, nx + 1, n)
    p["i"] = L.voxel(x, rng.integers(1, ny + 1, n), rng.integers(1, nz + 1, n), nx, ny, nz)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    p["ux"], p["q"], p["tag"] = 0.5, -1e-9, np.arange(n)          # charge too small to make fields that matter
    sp = e.new_species(-1.0, n + 16, n)
    e.set_particles(sp, p)
    ut_para, ut_perp = np.zeros(32, np.float32), np.zeros(32, np.float32)
    ut_para[sp], ut_perp[sp] = 0.1, 0.05
    e.set_maxwellian_reflux(code, ut_para, ut_perp, seed=12345)
    e.load_interpolator()
    for step in range(48):
        e.step(step, sort_interval=10)
    out = e.get_particles(sp)
    assert len(out) == n                                            # refluxed, not absorbed
    back = out["ux"] < 0
    assert back.all(), (~back).sum()
    u0 = -out["ux"].astype(np.float64)
    assert abs(u0.mean() - np.sqrt(np.pi / 2) * 0.1) <= 4 * 0.0655 / np.sqrt(n)
    assert abs((u0 ** 2).mean() - 2 * 0.1 ** 2) <= 4 * 0.02 / np.sqrt(n)
    for c in ("uy", "uz"):
        v = out[c].astype(np.float64)
        assert abs(v.mean()) <= 4 * 0.05 / np.sqrt(n) and abs(v.std() - 0.05) <= 4 * 0.05 / np.sqrt(2 * n)
    assert abs(np.corrcoef(out["uy"], out["uz"])[0, 1]) < 0.03 and abs(np.corrcoef(u0, out["uy"])[0, 1]) < 0.03
    xs = (out["i"] % (nx + 2)).astype(int)
    assert xs.min() >= 1 and xs.max() <= nx                         # inside the box, on their way back
    with pytest.raises(V.VpicHipError):
        e.set_maxwellian_reflux(-2, ut_para, ut_perp)               # -1 / -2 are reflect / absorb, not handlers


def test_surface_emitter(V, L):
    """vpic_hip_emit (src/emitter/child-langmuir.c:43-97, ccube.c, ivory.c): a plane of -z faces in a uniform E_z.
    Every face emits n particles of charge eps0 dx dy dt sqrt(coef |q_m E^3| / dz) / n with a half-Maxwellian
    normal momentum, Maxwellian tangential ones and a random age; their charge, negated, is left in rhob; a
    field of the other sign or below the threshold emits nothing.  Statistical: 4-sigma bounds."""
    nx, ny, nz = 16, 16, 8
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.4))
    e = V.Engine(g)
    e.set_vacuum()
    f = np.zeros(e.nv, L.field_t)
    f["ez"] = -0.5
    e.set_fields(f); e.load_interpolator()
    sp = e.new_species(-1.0, 200000, 50000)
    x, y = np.meshgrid(np.arange(2, nx), np.arange(2, ny), indexing="ij")      # not the outermost cells: accumulate_rhob doubles
    cells = L.voxel(x.ravel(), y.ravel(), np.full(x.size, 3), nx, ny, nz)      # the weights of nodes on the domain surface there
    comp = (cells << 5) | 4                                   # BOUNDARY(0,0,-1): the low-z face of each cell, emitting upwards
    m, ut_perp, ut_para = 40, 0.03, 0.08
    e.emit(sp, comp, m, ut_perp, ut_para, coef=32.0 / 81.0, thresh=0.0, seed=99)
    p = e.get_particles(sp)
    n = len(comp) * m
    assert len(p) == n
    qp = -1.0 * 1.0 * 1.0 * 0.4 * np.sqrt(32.0 / 81.0 * abs(-1.0 * (-0.5) ** 3) / 1.0) / m
    assert np.allclose(p["q"], qp, rtol=1e-6)
    assert (p["uz"] > 0).all() and set(np.unique(p["i"] // ((nx + 2) * (ny + 2)))) <= {3, 4}      # moving up, at most one cell on
    uz = p["uz"].astype(np.float64)
    assert abs(uz.mean() - ut_para * np.sqrt(2 / np.pi)) <= 4 * ut_para * 0.603 / np.sqrt(n)
    for c in ("ux", "uy"):
        v = p[c].astype(np.float64)
        assert abs(v.mean()) <= 4 * ut_perp / np.sqrt(n) and abs(v.std() - ut_perp) <= 4 * ut_perp / np.sqrt(2 * n)
    for c in ("dx", "dy"):
        v = p[c].astype(np.float64)
        assert abs(v.mean()) <= 4 * 0.577 / np.sqrt(n) and abs(v.std() - 0.577) <= 0.01
    rhob = e.get_fields()["rhob"].astype(np.float64).reshape(nz + 2, ny + 2, nx + 2)
    # bound charge: -q of every particle, trilinear onto the nodes of the emitting plane
    assert abs(rhob.sum() - (-qp * n)) <= 1e-5 * abs(qp * n)
    assert np.abs(rhob[3]).sum() > 0.999 * np.abs(rhob).sum()
    for kw in (dict(thresh=0.6), dict(coef=1.0, thresh=0.6)):                  # below the threshold: nothing
        e.emit(sp, comp, m, ut_perp, ut_para, **{"coef": 32.0 / 81.0, **kw})
        assert e.np(sp) == n
    e.emit(sp, (cells << 5) | 22, m, ut_perp, ut_para)                         # +z faces: the field pushes electrons back in
    assert e.np(sp) == n
    e.emit(sp, (cells << 5) | 13, m, ut_perp, ut_para)                         # volume components: not faces, do not emit
    assert e.np(sp) == n


def test_trajectory_20_steps(V, golden):
    """The chained step (src/vpic/advance.cxx:38-214) against the reference's own 20-step run."""
    nx, ny, nz = [int(v) for v in golden["t_dims"]]
    e = V.Engine(V.make_grid(nx, ny, nz, 8.0, 8.0, 8.0, golden["t_dt"]))
    e.set_vacuum()
    sps = []
    for k in (0, 1):
        p = golden[f"t_p{k}_in"]
        sp = e.new_species(-1.0, len(p), 4096)
        e.set_particles(sp, p)
        sps.append(sp)
    e.load_interpolator()
    ref = golden["t_energies"]
    en = np.zeros_like(ref)
    for step in range(ref.shape[0]):
        e.step(step, sort_interval=0)
        en[step, :6] = e.energy_f()
        for k, sp in enumerate(sps):
            en[step, 6 + k] = e.energy_p(sp)
    # per-species kinetic energy and the dominant field energies: fp32 accumulation noise only
    np.testing.assert_allclose(en[:, 6:], ref[:, 6:], rtol=1e-6)
    big = ref[-1, :6] > 1e-3 * ref[-1, :6].max()
    np.testing.assert_allclose(en[:, :6][:, big], ref[:, :6][:, big], rtol=2e-4)
    f, fr = e.get_fields(), golden["t_f_out"]
    for n in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        scale = np.abs(fr[n]).max()
        assert np.abs(f[n] - fr[n]).max() <= 2e-4 * max(scale, np.abs(fr["ex"]).max()), n
    for k, sp in enumerate(sps):
        p, pr = e.get_particles(sp), golden[f"t_p{k}_out"]
        assert np.array_equal(p["i"], pr["i"]) or (p["i"] != pr["i"]).mean() < 1e-3
        assert np.abs(p["ux"] - pr["ux"]).max() < 1e-4


def test_uniform_drift_512ppc(V, orc, L):
    """BASELINE.json configs[4] in miniature: cold uniform drift, 512 particles per cell -- every
    particle of a cell follows the same path (worst-case deposition conflicts, deterministic
    crossing fraction).  16^3 x 512 ppc = 2.1 M particles, 4 steps, against the CPU oracle."""
    n, ppc = 16, 512
    dt = np.float32(0.95 / np.sqrt(3.0))
    g = V.make_grid(n, n, n, float(n), float(n), float(n), dt)
    og = orc.make_grid(n, n, n, float(n), float(n), float(n), dt)
    rng = np.random.default_rng(5)
    npart = n ** 3 * ppc
    p = np.zeros(npart, L.particle_t)
    cell = np.repeat(np.arange(n ** 3), ppc)
    p["i"] = L.voxel(cell % n + 1, (cell // n) % n + 1, cell // (n * n) + 1, n, n, n)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, npart).astype(np.float32)
    p["ux"], p["uy"], p["uz"] = 0.1, 0.05, 0.02
    p["q"] = -1.0 / ppc
    e = V.Engine(g)
    e.set_vacuum()
    sp = e.new_species(-1.0, npart, 4096)
    e.set_particles(sp, p)
    e.load_interpolator()
    f = np.zeros(og.nv, L.field_t)
    fi = np.zeros(og.nv, L.interpolator_t)
    a = np.zeros(og.nv, L.accumulator_t)
    m = orc.vacuum_coefficients()
    species = [dict(p=p.copy(), np=npart, q_m=-1.0, pm=np.zeros(64, L.particle_mover_t))]
    orc.load_interpolator(fi, f, og)
    for step in range(4):
        e.clear_accumulators()
        assert e.advance_p(sp) == 0
        orc.clear_accumulators(a, og)
        assert orc.advance_p(species[0]["p"], npart, -1.0, species[0]["pm"], a, fi, og) == 0
        if step == 0:                                   # same inputs: particles bit-exact, sums to round-off
            assert bits_equal(e.get_particles(sp), species[0]["p"])
            acc_close(e.get_accumulator(), a)
        for obj, fld in ((e, None),):
            e.clear_jf(); e.unload_accumulator(); e.synchronize_jf(); e.advance_b(0.5); e.advance_e(); e.advance_b(0.5); e.load_interpolator()
        orc.clear_jf(f, og); orc.unload_accumulator(f, a, og); orc.synchronize_jf_local(f, og)
        orc.advance_b(f, og, 0.5); orc.advance_e(f, m, og); orc.advance_b(f, og, 0.5); orc.load_interpolator(fi, f, og)
    got, ref = e.get_fields(), f
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        scale = max(np.abs(ref["ex"]).max(), np.abs(ref[c]).max(), 1e-20)
        assert np.abs(got[c] - ref[c]).max() <= 5e-4 * scale, c
    pg, pr = e.get_particles(sp), species[0]["p"]
    assert (pg["i"] != pr["i"]).mean() < 1e-4 and np.abs(pg["ux"] - pr["ux"]).max() < 1e-5


def test_k8_center_uncenter(V, golden):
    e = V.Engine(k1_grid(V, golden))
    e.set_interpolator(golden["k8_fi"])
    p = golden["k8_p_in"]
    sp = e.new_species(-1.0, len(p), 64)
    e.set_particles(sp, p)
    e.uncenter_p(sp)
    assert bits_equal(e.get_particles(sp), golden["k8_p_uncentered"])
    e.center_p(sp)
    assert bits_equal(e.get_particles(sp), golden["k8_p_recentered"])


@pytest.mark.parametrize("tag", ["per", "pec", "abs"])
def test_k9_divergence_cleaning(V, golden, L, tag):
    """Every stage of the K9 chain (oracle/gen_golden.py) starts from the reference's own previous
    stage; equal as numbers, except rhof from accumulate_rho_p (float atomics: ACC_TOL)."""
    kw = {} if tag == "per" else dict(damp=0.01, fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS],
                                      pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    if tag == "abs":                                     # absorbing x (fields and particles), PEC / reflecting z
        kw = dict(fbc=[int(x) for x in golden["k3b_fbc"]], pbc=[int(x) for x in golden["k3b_pbc"]])
    e = V.Engine(k1_grid(V, golden, **kw))
    e.set_vacuum()
    G = lambda name: golden[f"k9{tag}_{name}"]

    def same(name, skip=()):
        f, ref = e.get_fields(), G(name)
        for n in f.dtype.names:
            if n not in skip:
                assert np.array_equal(f[n], ref[n]), (name, n)
        return f, ref

    p = golden["k9_p"]
    sp = e.new_species(-1.0, len(p), 64)
    e.set_particles(sp, p)
    e.set_fields(G("f_in")); e.clear_rhof(); e.accumulate_rho_p(sp)
    f, ref = same("f_rho_p", skip=("rhof",))
    assert np.abs(f["rhof"].astype(np.float64) - ref["rhof"]).max() <= ACC_TOL * np.abs(ref["rhof"]).max()
    e.set_fields(G("f_rho_p")); e.synchronize_rho(); same("f_rho_sync")
    e.set_fields(G("f_rho_sync")); e.compute_rhob(); same("f_rhob")
    f = G("f_rhob").copy(); f["rhob"] *= np.float32(0.9)
    e.set_fields(f); e.compute_div_e_err(); same("f_div_e")
    assert e.compute_rms_div_e_err() == pytest.approx(float(G("rms_div_e")), rel=1e-12)
    e.clean_div_e(); same("f_clean_e")
    e.compute_div_b_err(); same("f_div_b")
    assert e.compute_rms_div_b_err() == pytest.approx(float(G("rms_div_b")), rel=1e-12)
    e.clean_div_b(); same("f_clean_b")
    e.compute_curl_b(); same("f_curl_b")
    err = e.synchronize_tang_e_norm_b(); same("f_sync")
    assert err == pytest.approx(float(G("sync_err")), rel=1e-12)


@pytest.mark.parametrize("tag", ["per", "pec"])
def test_k13_several_materials(V, golden, L, tag):
    """Three materials (vacuum, anisotropic dielectric/magnetic, anisotropic conductor), ids per voxel and
    component: every field kernel that looks materials up, against the reference's outputs, equal as numbers."""
    kw = {} if tag == "per" else dict(damp=0.01, fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS],
                                      pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    e = V.Engine(k1_grid(V, golden, **kw))
    e.set_material_coefficients(golden[f"k13{tag}_mc"])
    G = lambda name: golden[f"k13{tag}_{name}"]

    def same(name):
        f, ref = e.get_fields(), G(name)
        for n in f.dtype.names:
            assert np.array_equal(f[n], ref[n]), (name, n)

    e.set_fields(G("f_in")); same("f_in")                      # the ids survive the round trip
    e.compute_curl_b(); same("f_curl_b")
    e.advance_b(0.5); e.advance_e(); same("f_e")
    np.testing.assert_allclose(e.energy_f(), G("en"), rtol=1e-12)
    e.compute_rhob(); same("f_rhob")
    f = G("f_rhob").copy(); f["rhob"] *= np.float32(0.9)
    e.set_fields(f); e.compute_div_e_err(); same("f_div_e")
    assert e.compute_rms_div_e_err() == pytest.approx(float(G("rms_div_e")), rel=1e-12)
    e.clean_div_e(); same("f_clean_e")


@pytest.mark.parametrize("tag", ["per", "pec"])
def test_k10_hydro(V, golden, L, tag):
    kw = {} if tag == "per" else dict(damp=0.01, fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS],
                                      pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    e = V.Engine(k1_grid(V, golden, **kw))
    e.set_interpolator(golden["k8_fi"])
    p = golden["k10_p"]
    sp = e.new_species(-1.0, len(p), 64)
    e.set_particles(sp, p)
    junk = np.zeros(e.nv, L.hydro_t); junk["ke"] = 3.0
    e.set_hydro(junk); e.clear_hydro(); e.accumulate_hydro_p(sp)
    h, ref = e.get_hydro(), golden[f"k10{tag}_h_acc"]
    for n in h.dtype.names[:-1]:                           # sums of float atomics: ACC_TOL of the largest entry
        assert np.abs(h[n].astype(np.float64) - ref[n]).max() <= ACC_TOL * np.abs(ref[n]).max(), n
    e.set_hydro(ref); e.synchronize_hydro()
    h, ref = e.get_hydro(), golden[f"k10{tag}_h_sync"]
    for n in h.dtype.names[:-1]:
        assert np.array_equal(h[n], ref[n]), n


def test_hydro_and_rho_from_sorted_cells(V, orc, L):
    """From a few particles per voxel on, accumulate_hydro_p / accumulate_rho_p sort the species and sum
    cell by cell (112 resp. 8 atomics per occupied cell instead of per particle).  Unsorted input, ~40
    particles per cell, reflecting z walls; against the CPU oracle, float-sum tolerance."""
    rng = np.random.default_rng(11)
    nx, ny, nz = 12, 6, 5
    kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    g = V.make_grid(nx, ny, nz, 12.0, 6.0, 5.0, np.float32(0.4), **kw)
    og = orc.make_grid(nx, ny, nz, 12.0, 6.0, 5.0, np.float32(0.4), **kw)
    n = nx * ny * nz * 40
    p = np.zeros(n, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    p["i"] = L.voxel(rng.integers(1, nx + 1, n), rng.integers(1, ny + 1, n), rng.integers(1, nz + 1, n), nx, ny, nz)
    for c in ("ux", "uy", "uz"):
        p[c] = (rng.standard_normal(n) * 0.3).astype(np.float32)
    p["q"] = rng.uniform(0.5, 1.5, n).astype(np.float32) * -0.01
    p["tag"] = np.arange(n)
    f = np.zeros(og.nv, L.field_t)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        f[c] = (rng.standard_normal(og.nv) * 0.2).astype(np.float32)
    fi = np.zeros(og.nv, L.interpolator_t)
    orc.load_interpolator(fi, f, og)
    ref_h = np.zeros(og.nv, L.hydro_t)
    orc.accumulate_hydro_p(ref_h, p, n, -1.0, fi, og)
    ref_f = f.copy(); ref_f["rhof"] = 0
    orc.accumulate_rho_p(ref_f, p, n, og)
    e = V.Engine(g)
    e.set_fields(f); e.set_interpolator(fi)
    sp = e.new_species(-1.0, n, 1024)
    e.set_particles(sp, p)
    e.clear_hydro(); e.accumulate_hydro_p(sp)
    h = e.get_hydro()
    for c in h.dtype.names[:-1]:
        assert np.abs(h[c].astype(np.float64) - ref_h[c]).max() <= ACC_TOL * np.abs(ref_h[c]).max(), c
    e.clear_rhof(); e.accumulate_rho_p(sp)
    rho = e.get_fields()["rhof"]
    assert np.abs(rho.astype(np.float64) - ref_f["rhof"]).max() <= ACC_TOL * np.abs(ref_f["rhof"]).max()
    got = e.get_particles(sp)                       # the species came back cell-sorted, nothing lost or altered
    assert np.all(np.diff(got["i"]) >= 0)
    assert bits_equal(got[np.argsort(got["tag"])], p)


def test_k12_absorbing_field_boundary(V, golden):
    """Higdon absorbing ghosts on x (local.c:84-108), PEC on z: two full field steps, equal as numbers."""
    e = V.Engine(k1_grid(V, golden, fbc=[int(x) for x in golden["k3b_fbc"]], pbc=[int(x) for x in golden["k3b_pbc"]]))
    e.set_vacuum()
    e.set_fields(golden["k5_f_in"])
    for _ in range(2):
        e.advance_b(0.5); e.advance_e(); e.advance_b(0.5)
    f = e.get_fields()
    for n in f.dtype.names:
        assert np.array_equal(f[n], golden["k12_f_out"][n]), n


@pytest.mark.parametrize("strides", [(1, 1, 1), (2, 4, 1), (4, 2, 8), (8, 8, 8)])
def test_dump_gather_layouts(V, L, strides):
    """vpic_hip_dump_gather (the payload of field_dump / hydro_dump, dump.cxx:1116-1552) against the
    numpy restatement pinned on the reference's files (oracle/dumpfmt.py): byte work, bit-exact.
    Two materials, so that the packed material-id words 16-19 and the aliasing words 20-23 are real."""
    from oracle import dumpfmt as D
    n = 8
    e = V.Engine(V.make_grid(n, n, n, 8.0, 8.0, 8.0, np.float32(0.3)))
    m = np.zeros(2, L.material_coefficient_t)
    for c in ("decayx", "decayy", "decayz", "drivex", "drivey", "drivez", "rmux", "rmuy", "rmuz", "nonconductive", "epsx", "epsy", "epsz"):
        m[c] = 1.0
    e.set_material_coefficients(m)
    rng = np.random.default_rng(20261004)
    f = np.zeros(e.nv, L.field_t)
    for c in f.dtype.names:
        f[c] = rng.integers(0, 2, e.nv) if f.dtype[c] == np.uint16 else rng.standard_normal(e.nv).astype(np.float32)
    e.set_fields(f)
    h = np.zeros(e.nv, L.hydro_t)
    for c in h.dtype.names[:-1]:
        h[c] = rng.standard_normal(e.nv).astype(np.float32)
    e.set_hydro(h)
    h = e.get_hydro()
    fw, hw = [0, 1, 2, 4, 5, 6, 15, 16, 17, 18, 23], [0, 1, 2, 7, 8, 13]
    for what, rec, words in ((0, f, fw), (0, f, list(range(24))), (1, h, hw), (1, h, [3])):
        assert np.array_equal(e.dump_gather(what, D.BAND, words, strides), D.gather(rec, n, n, n, D.BAND, words, strides))
    assert np.array_equal(e.dump_gather(0, D.INTERLEAVE, (), strides), D.gather(f, n, n, n, D.INTERLEAVE, (), strides))
    assert np.array_equal(e.dump_gather(1, D.INTERLEAVE, (), strides), D.gather(h, n, n, n, D.INTERLEAVE, (), strides))
    assert np.array_equal(e.dump_gather(1, D.INTERLEAVE_INNER, (), strides), D.gather(h, n, n, n, D.INTERLEAVE_INNER, (), strides))
    if strides == (1, 1, 1):
        # interleaved with unit strides is the array itself
        assert np.array_equal(e.dump_gather(0, D.INTERLEAVE, (), strides).reshape(-1), f.view(np.uint32))
        for bad in (dict(strides=(3, 1, 1)), dict(words=[24]), dict(words=[-1]), dict(what=1, words=[16])):
            kw = dict(what=0, layout=D.BAND, words=[0], strides=(1, 1, 1)); kw.update(bad)
            with pytest.raises(V.VpicHipError):
                e.dump_gather(**kw)


# ---- the FAST arithmetic instance of advance_p (include/vpic_hip.h: vpic_hip_set_push_mode) --------------------
# Stated tolerance: momenta within FAST_ULP units in the last place (of the particle's largest momentum component
# before or after the step) of the scalar pipeline's after one step (the
# reference's own V4 pipelines are not bit-identical to its scalar ones either: BASELINE.md section 2, 1.6e-5 in
# energy after 50 steps), positions within FAST_POS (cell units), the same cell for >= 99.99 % of the particles
# (a particle that ends a step within round-off of a face may be assigned to the neighbour cell), accumulators
# within FAST_ACC of the largest entry.
FAST_ULP, FAST_POS, FAST_ACC = 8, 2e-6, 5e-6


def ulp_diff(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("case", ["k2", "k3a", "k3b"])
def test_advance_p_fast_mode_within_stated_tolerance(V, golden, case):
    kw = {}
    if case == "k3b":
        kw = dict(fbc=[int(x) for x in golden["k3b_fbc"]], pbc=[int(x) for x in golden["k3b_pbc"]])
    e = V.Engine(k1_grid(V, golden, **kw))
    e.set_push_mode("fast")
    e.set_interpolator(golden["k2_fi"])
    p_in = golden["k2_p_in" if case == "k2" else "k3_p_in"]
    sp = e.new_species(-1.0, len(p_in) + 16, 4096)
    e.set_particles(sp, p_in)
    e.clear_accumulators()
    nm = e.advance_p(sp)
    got, ref = e.get_particles(sp), golden[case + "_p_out"]
    assert len(got) == len(ref)
    for c in ("ux", "uy", "uz"):
        # ulp of the largest momentum component of the particle: a component near zero carries the absolute
        # round-off of the cross products it was formed from
        scale = np.maximum.reduce([np.abs(ref[n]) for n in ("ux", "uy", "uz")] + [np.abs(p_in[n]) for n in ("ux", "uy", "uz")])
        assert np.all(np.abs(got[c].astype(np.float64) - ref[c]) <= FAST_ULP * np.spacing(scale)), c
    same = got["i"] == ref["i"]
    assert same.mean() >= 0.9999
    for c in ("dx", "dy", "dz"):
        assert np.abs(got[c][same].astype(np.float64) - ref[c][same]).max() <= FAST_POS, c
    a, r = e.get_accumulator(), golden[case + "_a_out"]
    for c in ("jx", "jy", "jz"):
        scale = max(np.abs(r[c]).max(), 1e-30)
        if same.all():
            assert np.abs(a[c].astype(np.float64) - r[c]).max() <= FAST_ACC * scale, c
    if case == "k3b":
        assert abs(nm - len(golden["k3b_pm"])) <= 1
    else:
        assert nm == 0


def test_fast_mode_deck_energies_track_the_exact_run(V, L):
    """50 steps of a 16^3 two-stream box in both arithmetic modes: kinetic and field energies of the FAST run stay
    within the spread the reference shows between its own scalar and SSE builds (1.6e-5
    relative on the smallest field-energy component after 50 steps; here 1e-4 of the total field energy and 1e-6
    of the kinetic energy)."""
    out = {}
    for mode in ("exact", "fast"):
        dt = np.float32(0.95 / np.sqrt(3.0))
        e = V.Engine(V.make_grid(16, 16, 16, 16.0, 16.0, 16.0, dt))
        e.set_vacuum()
        e.set_push_mode(mode)
        sps = []
        for k, u in enumerate(((0.2, 0, 0), (-0.2, 0, 0))):
            sp = e.new_species(-1.0, 16 ** 3 * 16, 4096)
            e.load_maxwellian(sp, 16, 1 + k, -float((0.2 / float(dt)) ** 2 / 32), u, 0.02)
            sps.append(sp)
        e.load_interpolator()
        for step in range(50):
            e.step(step, 10)
        out[mode] = (sum(e.energy_p(sp) for sp in sps), e.energy_f())
    (ke0, ef0), (ke1, ef1) = out["exact"], out["fast"]
    assert abs(ke1 - ke0) <= 1e-6 * abs(ke0)
    assert np.abs(ef1 - ef0).max() <= 1e-4 * ef0.sum()


def test_maxwellian_reflux_particle_by_particle(V, orc, L):
    """The reflux handler on the device with the draws of tests/golden/reflux.npz (vpic_hip_set_reflux_draws), against
    the oracle's restatement -- which that fixture pins on the reference's own handler bit for bit -- particle by
    particle: 240 particles parked on the six reflux faces of a box with unequal cells, one boundary_p.  New momenta
    within 4e-6 relative (the device's logf / sqrtf are not the host libm's), positions within 1e-5 of a cell, every
    particle in the oracle's cell."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reflux.npz"))
    nx, ny, nz = [int(v) for v in g["dims"]]
    lx, ly, lz, dt = [float(v) for v in g["box"]]
    code = -3
    grid = V.make_grid(nx, ny, nz, lx, ly, lz, np.float32(dt), fbc=[L.PEC_FIELDS] * 6, pbc=[code] * 6)
    og = orc.make_grid(nx, ny, nz, lx, ly, lz, np.float32(dt), fbc=[L.PEC_FIELDS] * 6, pbc=[code] * 6)
    p, pm, n = g["p"].copy(), g["pm"].copy(), len(g["p"])
    e = V.Engine(grid)
    e.set_vacuum()
    sp = e.new_species(-1.0, 2 * n, 2 * n)
    e.set_particles(sp, p)
    e.set_movers(sp, pm)
    ut_para, ut_perp = np.zeros(32, np.float32), np.zeros(32, np.float32)
    ut_para[sp], ut_perp[sp] = g["ut"]
    e.set_maxwellian_reflux(code, ut_para, ut_perp, seed=1)
    e.set_reflux_draws(g["draws"])
    e.clear_accumulators()
    e.boundary_p_pack()
    got = e.get_particles(sp)
    # oracle: every particle is a mover; its injector, then the injection (append + finish the move)
    inj = np.zeros(n, L.particle_injector_t)
    for k in range(n):
        inj[k] = orc.maxwellian_reflux(g["draws"][k], p, pm[k:k + 1], og, float(g["ut"][0]), float(g["ut"][1]), int(g["face"][k]))
    ref = np.zeros(2 * n, L.particle_t)
    rpm = np.zeros(2 * n, L.particle_mover_t)
    acc = np.zeros(og.nv, L.accumulator_t)
    new_np, _ = orc.boundary_p_inject(ref, 0, rpm, 0, inj, acc, og)
    ref = ref[:new_np]
    assert len(got) == n == new_np
    a, b = got[np.argsort(got["q"])], ref[np.argsort(ref["q"])]          # the charge identifies the particle
    assert np.array_equal(a["q"], b["q"])
    for c in ("ux", "uy", "uz"):
        scale = np.maximum.reduce([np.abs(b[m]) for m in ("ux", "uy", "uz")])
        assert np.all(np.abs(a[c].astype(np.float64) - b[c]) <= 4e-6 * scale), c
    assert np.array_equal(a["i"], b["i"])
    for c in ("dx", "dy", "dz"):
        assert np.abs(a[c].astype(np.float64) - b[c]).max() <= 1e-5, c
    e.set_reflux_draws(np.zeros((0, 3), np.float32))


def test_child_langmuir_emitter_particle_by_particle(V, orc, L):
    """vpic_hip_emit with the draws of tests/golden/reflux.npz (vpic_hip_set_emit_draws) against the oracle's
    restatement of child-langmuir.c -- pinned on the reference's own model bit for bit by that fixture -- particle by
    particle: 72 faces of all six orientations in a random interpolator.  Every emitted particle of the oracle has its
    twin on the device (same cell, position within 1e-5, momentum within 1e-6 relative, charge within 2e-7 relative:
    the device takes the square root of the charge law in float, the reference in double); bound charge within 1e-6."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reflux.npz"))
    nx, ny, nz = [int(v) for v in g["dims"]]
    lx, ly, lz, dt = [float(v) for v in g["box"]]
    grid = V.make_grid(nx, ny, nz, lx, ly, lz, np.float32(dt))
    og = orc.make_grid(nx, ny, nz, lx, ly, lz, np.float32(dt))
    n_emit, ut_perp, ut_para, q_m = g["emit_par"]
    n_emit = int(n_emit)
    comp, fi = g["emit_component"], g["emit_fi"]
    # oracle
    cap = 4 * len(comp) * n_emit
    p, pm = np.zeros(cap, L.particle_t), np.zeros(cap, L.particle_mover_t)
    f, a = np.zeros(og.nv, L.field_t), np.zeros(og.nv, L.accumulator_t)
    new_np, nm = orc.child_langmuir(p, 0, pm, 0, comp, n_emit, ut_perp, ut_para, q_m, fi, f, a, og, g["emit_draws"])
    ref = p[:new_np]
    # the device's table is indexed by slot (component, k); the reference draws only for faces that emit, in order
    slot_draws = np.zeros((len(comp) * n_emit, 6))
    k = 0
    for c, cid in enumerate(comp):
        t, v = int(cid) & 31, int(cid) >> 5
        axis, dirn = {12: (0, 1), 10: (1, 1), 4: (2, 1), 14: (0, -1), 16: (1, -1), 22: (2, -1)}.get(t, (None, 0))
        if axis is None or not q_m * (dirn * float(fi[("ex", "ey", "ez")[axis]][v])) > 0:
            continue
        slot_draws[c * n_emit:(c + 1) * n_emit] = g["emit_draws"][k:k + n_emit]
        k += n_emit
    assert k == new_np
    e = V.Engine(grid)
    e.set_vacuum()
    sp = e.new_species(float(q_m), cap, cap)
    e.set_interpolator(fi)
    e.set_emit_draws(slot_draws)
    # both draw tables on ONE engine, then the reflux table replaced and cleared: the emitter's table must survive
    # (round-2 advisor: vpic_hip_set_reflux_draws used to free it too and leave the pointer dangling)
    e.set_reflux_draws(np.full((16, 3), 0.5, np.float32))
    e.set_reflux_draws(np.full((8, 3), 0.25, np.float32))
    e.set_reflux_draws(np.zeros((0, 3), np.float32))
    e.clear_accumulators()
    e.emit(sp, comp, n_emit, float(ut_perp), float(ut_para))
    got = e.get_particles(sp)
    assert len(got) == new_np and e.nm(sp) == 0
    used = np.zeros(len(got), bool)
    for r in ref:
        cand = np.where((got["i"] == r["i"]) & ~used)[0]
        d = np.abs(np.stack([got[c][cand].astype(np.float64) - r[c] for c in ("dx", "dy", "dz")])).max(axis=0)
        j = cand[np.argmin(d)]
        assert d.min() <= 1e-5, (r, got[j])
        used[j] = True
        scale = max(abs(float(r["ux"])), abs(float(r["uy"])), abs(float(r["uz"])))
        for c in ("ux", "uy", "uz"):
            assert abs(float(got[c][j]) - float(r[c])) <= 1e-6 * scale, c
        assert abs(float(got["q"][j]) / float(r["q"]) - 1) <= 2e-7
    rb = e.get_fields()["rhob"].astype(np.float64)
    assert np.abs(rb - f["rhob"]).max() <= 1e-6 * np.abs(f["rhob"]).max()
    e.set_emit_draws(np.zeros((0, 6)))


@pytest.mark.parametrize("walls", [False, True])
@pytest.mark.parametrize("dims", [(70, 19, 37), (64, 8, 32), (5, 3, 2), (130, 9, 65), (63, 7, 31), (65, 9, 33)])
def test_field_advance_through_lds_tiles_equals_per_voxel(V, L, dims, walls, monkeypatch):
    """advance_b and advance_e tiled into LDS with a one-cell halo (fields.hip: advance_b_tiled_kernel, advance_e_tiled_kernel --
    a plane of E / cB read once per tile into a ring of two planes, swept along z) against the one-thread-per-voxel kernels
    (themselves bit-exact against the reference's K5 goldens, and so are the tiles: test_k5_* above): half a B advance, the E
    advance -- whole box, and split into the planes x = 2..nx and the two outer ones as the multi-domain step runs it -- and the
    second half B advance, with damping, periodic and conducting walls; the same bits in every component on boxes that are not
    multiples of the tile, one voxel short of and beyond a tile, smaller than a tile and longer than one z sweep."""
    nx, ny, nz = dims
    rng = np.random.default_rng(11)
    kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES]) if walls else {}
    g = V.make_grid(nx, ny, nz, float(nx), 1.5 * ny, 0.75 * nz, np.float32(0.3), damp=0.01, **kw)
    nv = (nx + 2) * (ny + 2) * (nz + 2)
    f0 = np.zeros(nv, L.field_t)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz", "tcax", "tcay", "tcaz", "jfx", "jfy", "jfz"):
        f0[c] = rng.standard_normal(nv).astype(np.float32)
    out = []
    for tiles in ("2", "0"):
        monkeypatch.setenv("VPIC_HIP_FIELD_TILES", tiles)
        for split in (False, True):
            e = V.Engine(g)
            e.set_vacuum()
            e.set_fields(f0)
            e.advance_b(0.5)
            if split:
                e.advance_e_part(1)
                e.advance_e_part(2)
            else:
                e.advance_e()
            e.advance_b(0.5)
            out.append(e.get_fields())
            e.close()
    assert bits_equal(out[0], out[2]) and bits_equal(out[1], out[3])      # tiles against per voxel: whole box, split box
    for n in out[0].dtype.names:
        assert np.array_equal(out[0][n], out[1][n]), n                      # (and the split changes nothing)


@pytest.mark.parametrize("dims", [(70, 19, 37), (64, 8, 32), (5, 3, 2), (130, 9, 65)])
def test_unload_through_lds_tiles_equals_unload_per_voxel(V, L, dims, monkeypatch):
    """clear_jf + unload_accumulator tiled into LDS with a one-cell halo (fields.hip, clear_unload_tiled_kernel: every
    accumulator record read once per tile, swept along z) against the one-thread-per-voxel pass of rounds 2-3 and against
    the two separate calls (themselves bit-exact against the reference's K4 golden above): the same bits on boxes that
    are not multiples of the tile, smaller than a tile, and longer than one z sweep."""
    nx, ny, nz = dims
    rng = np.random.default_rng(7)
    g = V.make_grid(nx, ny, nz, float(nx), 1.5 * ny, 0.75 * nz, np.float32(0.4))
    nv = (nx + 2) * (ny + 2) * (nz + 2)
    a = np.zeros(nv, L.accumulator_t)
    for c in ("jx", "jy", "jz"):
        a[c] = rng.standard_normal((nv, 4)).astype(np.float32)
    f0 = np.zeros(nv, L.field_t)
    for c in ("jfx", "jfy", "jfz", "ex", "cbz"):
        f0[c] = rng.standard_normal(nv).astype(np.float32)           # (old jf must be overwritten, the rest left alone)
    out = []
    for tiled in ("2", "0", None):                        # 2: the tiles whatever the grid's size (by default only where they fill the chip)
        if tiled is not None:
            monkeypatch.setenv("VPIC_HIP_UNLOAD_TILED", tiled)
        e = V.Engine(g)
        e.set_fields(f0)
        e.set_accumulator(a)
        if tiled is None:
            e.clear_jf()
            e.unload_accumulator()
        else:
            e.clear_jf_unload_accumulator()
        out.append(e.get_fields())
        e.close()
    assert bits_equal(out[0], out[2]) and bits_equal(out[1], out[2])
