"""The oracle (oracle/vpic_oracle.c) against vectors written by the reference itself
(oracle/gen_golden.py ran /root/reference's own compiled scalar sources).  CPU-only.
Everything is bit-exact: same operation order, same compiler flags."""
import os

import numpy as np
import pytest

from conftest import bits_equal


def k1_grid(orc, golden, **kw):
    nx, ny, nz = [int(v) for v in golden["k1_dims"]]
    return orc.make_grid(nx, ny, nz, 6.0, 5.0, 4.0, np.float32(0.3), **kw)


def acc(L, g, n_pipeline):
    stride = (g.nv + 1) & ~1
    return np.zeros((1 + n_pipeline) * stride, L.accumulator_t)


def test_k1_load_interpolator(orc, golden, L):
    g = k1_grid(orc, golden)
    fi = np.zeros(g.nv, L.interpolator_t)
    orc.load_interpolator(fi, golden["k1_f"].copy(), g)
    assert bits_equal(fi, golden["k1_fi"])


@pytest.mark.parametrize("case", ["k2", "k3a", "k3b"])
def test_advance_p(orc, golden, L, case):
    npipe = int(golden["n_pipeline"])
    kw = {}
    if case == "k3b":
        kw = dict(fbc=list(golden["k3b_fbc"]), pbc=list(golden["k3b_pbc"]))
    g = k1_grid(orc, golden, **kw)
    p = golden["k2_p_in" if case == "k2" else "k3_p_in"].copy()
    a = acc(L, g, npipe)
    pm = np.zeros(4096, L.particle_mover_t)
    nm = orc.advance_p(p, len(p), -1.0, pm, a, golden["k2_fi"].copy(), g, n_pipeline=npipe)
    orc.reduce_accumulators(a, g, npipe)
    assert bits_equal(p, golden[case + "_p_out"])
    assert bits_equal(a[:g.nv], golden[case + "_a_out"])
    if case == "k3b":
        assert nm == len(golden["k3b_pm"]) and nm > 0
        assert bits_equal(pm[:nm], golden["k3b_pm"])
        assert np.all(np.diff(pm["i"][:nm]) > 0)      # movers ascend in particle index
    else:
        assert nm == 0


def test_k4_unload_and_sync_jf(orc, golden, L):
    g = k1_grid(orc, golden)
    f = golden["k4_f_in"].copy()
    orc.clear_jf(f, g)
    orc.unload_accumulator(f, golden["k4_a"].copy(), g)
    assert bits_equal(f, golden["k4_f_unloaded"])
    orc.synchronize_jf_local(f, g)
    assert bits_equal(f, golden["k4_f_synced"])


def test_k5_advance_b_e(orc, golden, L):
    g = k1_grid(orc, golden)
    m = orc.vacuum_coefficients()
    f = golden["k5_f_in"].copy()
    orc.advance_b(f, g, 0.5)
    assert bits_equal(f, golden["k5_f_b"])
    orc.advance_e(f, m, g)
    assert bits_equal(f, golden["k5_f_e"])
    assert np.array_equal(orc.energy_f(f, m, g), golden["k6_energy_f"])


def test_k5d_damped_pec_z(orc, golden, L):
    fbc = [0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS]
    pbc = [0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES]
    g = k1_grid(orc, golden, damp=0.01, fbc=fbc, pbc=pbc)
    m = orc.vacuum_coefficients()
    f = golden["k5_f_in"].copy()
    orc.advance_b(f, g, 0.5)
    orc.advance_e(f, m, g)
    orc.advance_b(f, g, 0.5)
    assert bits_equal(f, golden["k5d_f_out"])
    f = golden["k4_f_unloaded"].copy()
    orc.synchronize_jf_local(f, g)
    assert bits_equal(f, golden["k5d_f_jf_synced"])


def test_k6_energy_p(orc, golden, L):
    g = k1_grid(orc, golden)
    e = orc.energy_p(golden["k2_p_in"].copy(), len(golden["k2_p_in"]), -1.0, golden["k2_fi"].copy(), g)
    # the reference adds its per-pipeline partial sums (energy_p.cxx:151-153); the oracle sums
    # sequentially: equal to a few ulp of a double
    assert e == pytest.approx(float(golden["k6_energy_p"]), rel=1e-13)


@pytest.mark.parametrize("oop", [1, 0])
def test_k7_sort_p(orc, golden, L, oop):
    g = k1_grid(orc, golden)
    p = golden["k7_p_in"].copy()
    part = np.zeros(g.nv + 1, np.int32)
    orc.sort_p(p, len(p), part, g, out_of_place=oop)
    assert bits_equal(p, golden["k7_p_oop" if oop else "k7_p_inplace"])
    assert np.array_equal(part, golden["k7_partition"])
    assert np.all(np.diff(p["i"]) >= 0)


def test_trajectory_20_steps(orc, golden, L):
    """Kernels chained as vpic_simulation::advance does (src/vpic/advance.cxx:38-214):
    8^3 periodic two-stream, 2 species, 20 steps -- energies every step and the final state."""
    nx, ny, nz = [int(v) for v in golden["t_dims"]]
    npipe = int(golden["n_pipeline"])
    g = orc.make_grid(nx, ny, nz, 8.0, 8.0, 8.0, golden["t_dt"])
    m = orc.vacuum_coefficients()
    f = np.zeros(g.nv, L.field_t)
    fi = np.zeros(g.nv, L.interpolator_t)
    a = acc(L, g, npipe)
    ps = [golden["t_p0_in"].copy(), golden["t_p1_in"].copy()]
    pm = np.zeros(4096, L.particle_mover_t)
    orc.load_interpolator(fi, f, g)
    en = np.zeros_like(golden["t_energies"])
    for step in range(en.shape[0]):
        orc.clear_accumulators(a, g, npipe)
        for p in ps:
            assert orc.advance_p(p, len(p), -1.0, pm, a, fi, g, n_pipeline=npipe) == 0
        orc.reduce_accumulators(a, g, npipe)
        orc.clear_jf(f, g)
        orc.unload_accumulator(f, a, g)
        orc.synchronize_jf_local(f, g)
        orc.advance_b(f, g, 0.5)
        orc.advance_e(f, m, g)
        orc.advance_b(f, g, 0.5)
        orc.load_interpolator(fi, f, g)
        en[step, :6] = orc.energy_f(f, m, g)
        for s, p in enumerate(ps):
            en[step, 6 + s] = orc.energy_p(p, len(p), -1.0, fi, g)
    assert bits_equal(f, golden["t_f_out"])
    assert bits_equal(ps[0], golden["t_p0_out"]) and bits_equal(ps[1], golden["t_p1_out"])
    np.testing.assert_allclose(en, golden["t_energies"], rtol=1e-13)


def test_k8_center_uncenter(orc, golden, L):
    g = k1_grid(orc, golden)
    p = golden["k8_p_in"].copy()
    orc.uncenter_p(p, len(p), -1.0, golden["k8_fi"].copy(), g)
    assert bits_equal(p, golden["k8_p_uncentered"])
    orc.center_p(p, len(p), -1.0, golden["k8_fi"].copy(), g)
    assert bits_equal(p, golden["k8_p_recentered"])
    # center undoes uncenter to round-off (it is its inverse in exact arithmetic)
    assert np.abs(p["ux"] - golden["k8_p_in"]["ux"]).max() < 1e-5


def k9_steps(api, golden, tag, g, m, sync_rho, sync_te, check):
    """The K9 chain of oracle/gen_golden.py, each stage compared before the next one starts."""
    p = golden["k9_p"]
    f = golden[f"k9{tag}_f_in"].copy()
    api.clear_rhof(f, g); api.accumulate_rho_p(f, p, len(p), g); check(f, "f_rho_p")
    sync_rho(f, g); check(f, "f_rho_sync")
    api.compute_rhob(f, m, g); check(f, "f_rhob")
    f["rhob"] *= np.float32(0.9)
    api.compute_div_e_err(f, m, g); check(f, "f_div_e")
    rms_e = api.compute_rms_div_e_err(f, g)
    api.clean_div_e(f, m, g); check(f, "f_clean_e")
    api.compute_div_b_err(f, g); check(f, "f_div_b")
    rms_b = api.compute_rms_div_b_err(f, g)
    api.clean_div_b(f, g); check(f, "f_clean_b")
    api.compute_curl_b(f, m, g); check(f, "f_curl_b")
    err = sync_te(f, g); check(f, "f_sync")
    return rms_e, rms_b, err


def k9_grid_kw(golden, L, tag):
    if tag == "per":
        return {}
    if tag == "pec":
        return dict(damp=0.01, fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    return dict(fbc=list(golden["k3b_fbc"]), pbc=list(golden["k3b_pbc"]))      # absorbing x, PEC z


@pytest.mark.parametrize("tag", ["per", "pec", "abs"])
def test_k9_divergence_cleaning(orc, golden, L, tag):
    g = k1_grid(orc, golden, **k9_grid_kw(golden, L, tag))
    m = orc.vacuum_coefficients()

    def check(f, name):
        assert bits_equal(f, golden[f"k9{tag}_{name}"]), name

    rms_e, rms_b, err = k9_steps(orc, golden, tag, g, m, orc.synchronize_rho_local, orc.synchronize_tang_e_norm_b_local, check)
    # double sums in another order than the reference's per-pipeline partial sums
    assert abs(rms_e - float(golden[f"k9{tag}_rms_div_e"])) <= 1e-12 * abs(rms_e)
    assert abs(rms_b - float(golden[f"k9{tag}_rms_div_b"])) <= 1e-12 * abs(rms_b)
    assert abs(err - float(golden[f"k9{tag}_sync_err"])) <= 1e-12 * abs(err)


@pytest.mark.parametrize("tag", ["per", "pec"])
def test_k10_hydro(orc, golden, L, tag):
    kw = {} if tag == "per" else dict(damp=0.01, fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS],
                                      pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    g = k1_grid(orc, golden, **kw)
    h = np.zeros(g.nv, L.hydro_t)
    h["ke"] = 3.0
    orc.clear_hydro(h, g)
    p = golden["k10_p"]
    orc.accumulate_hydro_p(h, p, len(p), -1.0, golden["k8_fi"].copy(), g)
    assert bits_equal(h, golden[f"k10{tag}_h_acc"])
    orc.synchronize_hydro_local(h, g)
    assert bits_equal(h, golden[f"k10{tag}_h_sync"])


def test_k11_boundary_p_absorbing(orc, golden, L):
    """boundary_p on one rank: movers on absorbing faces are charged to rhob and removed by
    back-filling from the end of the array, in reverse mover order (boundary_p.c:9-71,244-265)."""
    g = k1_grid(orc, golden, fbc=list(golden["k3b_fbc"]), pbc=list(golden["k3b_pbc"]))
    p, pm, f = golden["k3b_p_out"].copy(), golden["k3b_pm"].copy(), golden["k11_f_in"].copy()
    new_np, per_face = orc.boundary_p_pack(p, len(p), pm, len(pm), 0, f, g, len(pm))
    assert all(len(x) == 0 for x in per_face)
    assert new_np == len(golden["k11_p_out"])
    assert bits_equal(p[:new_np], golden["k11_p_out"])
    assert bits_equal(f, golden["k11_f_out"])


@pytest.mark.parametrize("tag", ["per", "abs"])
def test_k11_move_p(orc, golden, L, tag):
    kw = {} if tag == "per" else dict(fbc=list(golden["k3b_fbc"]), pbc=list(golden["k3b_pbc"]))
    g = k1_grid(orc, golden, **kw)
    p = golden["k3_p_in"][:64].copy()
    pm = golden[f"k11{tag}_pm_in"].copy()
    a = np.zeros(g.nv, L.accumulator_t)
    ret = np.array([orc.move_p(p, pm[k:k + 1], a, g) for k in range(64)], np.int32)
    assert np.array_equal(ret, golden[f"k11{tag}_ret"])
    assert bits_equal(p, golden[f"k11{tag}_p_out"]) and bits_equal(pm, golden[f"k11{tag}_pm_out"])
    assert bits_equal(a, golden[f"k11{tag}_a_out"])


def test_k12_absorbing_field_boundary(orc, golden, L):
    """Higdon absorbing ghosts on x (local.c:84-108), PEC on z: two full field steps."""
    g = k1_grid(orc, golden, fbc=list(golden["k3b_fbc"]), pbc=list(golden["k3b_pbc"]))
    m = orc.vacuum_coefficients()
    f = golden["k5_f_in"].copy()
    for _ in range(2):
        orc.advance_b(f, g, 0.5); orc.advance_e(f, m, g); orc.advance_b(f, g, 0.5)
    assert bits_equal(f, golden["k12_f_out"])


def test_dump_layout_restatement_matches_reference_files(golden_deck=None):
    """oracle/dumpfmt.py is pinned when oracle/deck16.py writes the fixtures (every strided / banded file
    of the reference equals gather() of its raw dump); here: the index rule on its own, and the array
    headers the reference wrote for those shapes."""
    import os
    from oracle import deck16, dumpfmt as D
    assert list(D.offsets(16, 2, False)) == [0, 1, 3, 5, 7, 9, 11, 13, 15, 17]
    assert list(D.offsets(16, 1, False)) == [0, 0] + list(range(1, 16)) + [17]        # stride 1 inside a strided dump: i-1
    assert list(D.offsets(4, 1, True)) == [0, 1, 2, 3, 4, 5]
    assert list(D.offsets(16, 4, False, inner=True)) == [0, 3, 7, 11]
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "deck16.npz"))
    f = gold["dump_fields16"]
    for name, kind, layout, words, strides in deck16.DUMP_CASES:
        head = gold["dump_" + name + "_head"]
        elem, ndim, dx, dy, dz = head[D.HEADER_V0:].view(np.int32)
        inner = layout == D.INTERLEAVE_INNER
        assert (elem, ndim) == (80 if kind == "f" else 64, 3)
        assert (dx, dy, dz) == tuple(16 // s + (0 if inner else 2) for s in strides), name
    g = D.gather(f, 16, 16, 16, D.INTERLEAVE, (), (1, 1, 1))
    assert np.array_equal(g.reshape(-1), f.view(np.uint32))
    assert np.array_equal(D.gather(f, 16, 16, 16, D.BAND, [4], (1, 1, 1))[0].reshape(-1).view(np.float32), f["cbx"])


@pytest.mark.parametrize("tag", ["per", "pec"])
def test_k13_several_materials(orc, golden, L, tag):
    """Vacuum + an anisotropic dielectric/magnetic material + an anisotropic conductor, ids drawn per voxel and
    component: the coefficient table (sfa.c:145-177) and the field operations that look materials up, bit for bit."""
    g = k1_grid(orc, golden, **k9_grid_kw(golden, L, tag))
    m = orc.material_coefficients(golden["k13_props"], g.dt, g.eps0)
    assert bits_equal(m, golden[f"k13{tag}_mc"])
    G = lambda name: golden[f"k13{tag}_{name}"]
    f = G("f_in").copy()
    orc.compute_curl_b(f, m, g); assert bits_equal(f, G("f_curl_b"))
    orc.advance_b(f, g, 0.5); orc.advance_e(f, m, g); assert bits_equal(f, G("f_e"))
    np.testing.assert_allclose(orc.energy_f(f, m, g), G("en"), rtol=1e-12)
    orc.compute_rhob(f, m, g); assert bits_equal(f, G("f_rhob"))
    f["rhob"] *= np.float32(0.9)
    orc.compute_div_e_err(f, m, g); assert bits_equal(f, G("f_div_e"))
    assert abs(orc.compute_rms_div_e_err(f, g) - float(G("rms_div_e"))) <= 1e-12 * float(G("rms_div_e"))
    orc.clean_div_e(f, m, g); assert bits_equal(f, G("f_clean_e"))


def test_maxwellian_reflux_restatement_against_the_references_handler(orc, L):
    """tests/golden/reflux.npz (oracle/gen_reflux.py): the reference's own maxwellian_reflux (boundary.h) called for
    240 particles parked on the six faces of a box with unequal cell sizes, together with the three numbers each call
    drew from the reference's generator.  The restatement fed the same numbers writes the same injector, bit for bit."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reflux.npz"))
    nx, ny, nz = [int(v) for v in g["dims"]]
    lx, ly, lz, dt = [float(v) for v in g["box"]]
    og = orc.make_grid(nx, ny, nz, lx, ly, lz, np.float32(dt))
    for k in range(len(g["p"])):
        out = orc.maxwellian_reflux(g["draws"][k], g["p"], g["pm"][k:k + 1], og, float(g["ut"][0]), float(g["ut"][1]), int(g["face"][k]))
        ref = g["inj"][k]
        for c in ref.dtype.names:
            assert np.asarray(out[c]).view(np.uint32) == np.asarray(ref[c]).view(np.uint32), (k, c)


def test_child_langmuir_restatement_against_the_references_model(orc, L):
    """tests/golden/reflux.npz, emit_*: the reference's own child_langmuir (emitter.h) on 72 faces of all six
    orientations (+ a cell body) in a random interpolator, with the six numbers each emitted particle drew.  The
    restatement fed the same numbers produces the same particles, bound charge and currents, bit for bit."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reflux.npz"))
    nx, ny, nz = [int(v) for v in g["dims"]]
    lx, ly, lz, dt = [float(v) for v in g["box"]]
    og = orc.make_grid(nx, ny, nz, lx, ly, lz, np.float32(dt))
    n_emit, ut_perp, ut_para, q_m = g["emit_par"]
    cap = 4 * len(g["emit_component"]) * int(n_emit)
    p, pm = np.zeros(cap, L.particle_t), np.zeros(cap, L.particle_mover_t)
    f, a = np.zeros(og.nv, L.field_t), np.zeros(og.nv, L.accumulator_t)
    new_np, nm = orc.child_langmuir(p, 0, pm, 0, g["emit_component"], int(n_emit), ut_perp, ut_para, q_m, g["emit_fi"], f, a, og, g["emit_draws"])
    assert new_np == len(g["emit_p"]) and nm == 0
    for c in ("dx", "dy", "dz", "i", "ux", "uy", "uz", "q"):          # (the model leaves the tags of a new particle as they were)
        assert bits_equal(p[c][:new_np], g["emit_p"][c]), c
    assert bits_equal(f["rhob"], g["emit_rhob"]) and bits_equal(a, g["emit_a"])


def test_deposition_conserves_charge_in_the_oracle(orc, L):
    """The property tests/test_gpu_headline.py checks the GPU's deposits by at the bench's size, pinned here on the CPU
    restatement (itself bit-identical to the reference: K2/K3/K4 above): rho from accumulate_rho_p (rho_p.c:23-86) before
    and after one advance_p (+ move_p) of a hot species, jf from unload_accumulator (unload_accumulator.cxx:36-52):
    (rho_1 - rho_0) / dt + div jf = 0 at every node clear of the ghost bookkeeping, to float-sum rounding."""
    nx, ny, nz, ppc = 12, 10, 8, 16
    dt = np.float32(0.95 / np.sqrt(3))
    g = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), dt)
    rng = np.random.default_rng(1)
    n = nx * ny * nz * ppc
    p = np.zeros(n, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    p["i"] = L.voxel(rng.integers(1, nx + 1, n), rng.integers(1, ny + 1, n), rng.integers(1, nz + 1, n), nx, ny, nz)
    for c in ("ux", "uy", "uz"):
        p[c] = (0.5 * rng.standard_normal(n)).astype(np.float32)
    p["q"] = (-0.01 * rng.uniform(0.5, 1.5, n)).astype(np.float32)
    f = np.zeros(g.nv, L.field_t)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        f[c] = (0.05 * rng.standard_normal(g.nv)).astype(np.float32)
    fi = np.zeros(g.nv, L.interpolator_t)
    orc.load_interpolator(fi, f, g)

    def rho(p):
        ff = np.zeros(g.nv, L.field_t)
        orc.accumulate_rho_p(ff, p, len(p), g)
        return ff["rhof"].copy()

    rho0 = rho(p)
    a, pm = np.zeros(g.nv, L.accumulator_t), np.zeros(64, L.particle_mover_t)
    assert orc.advance_p(p, n, -1.0, pm, a, fi, g) == 0
    ff = np.zeros(g.nv, L.field_t)
    orc.unload_accumulator(ff, a, g)
    from test_gpu_headline import continuity_residual
    res, scale = continuity_residual(rho0, rho(p), ff, (nx, ny, nz), dt)
    assert res <= 1e-5 * scale, (res, scale)
