#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (oracle/_ref/libvpic_ref.so, built
from /root/reference by oracle/Makefile) on seeded inputs.  Container-only: the GPU box has no
reference tree; it consumes the committed vectors.  TEST INFRASTRUCTURE.

    python oracle/gen_golden.py            # rewrites tests/golden/kernels.npz

Cases.  Every case stores inputs AND the reference's outputs:
  K1 load_interpolator            K2 advance_p, in-cell only        K3 advance_p with crossings:
  K4 clear_jf+unload+sync_jf      K5 advance_b / advance_e             periodic wrap, reflecting z,
  K6 energy_p / energy_f          K7 sort_p (both variants)            absorbing x (-> movers)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

L = importlib.import_module("old-vpic_amd.layout")
from oracle import pyref  # noqa: E402


def rand_field(rng, nv, scale=1.0, material=False):
    f = np.zeros(nv, L.field_t)
    for n in ("ex", "ey", "ez", "cbx", "cby", "cbz", "tcax", "tcay", "tcaz", "jfx", "jfy", "jfz", "rhob", "rhof"):
        f[n] = (rng.standard_normal(nv) * scale).astype(np.float32)
    return f


def rand_particles(rng, n, nx, ny, nz, u_scale, q=-0.37, sorted_cells=False):
    p = np.zeros(n, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    x = rng.integers(1, nx + 1, n)
    y = rng.integers(1, ny + 1, n)
    z = rng.integers(1, nz + 1, n)
    p["i"] = L.voxel(x, y, z, nx, ny, nz)
    if sorted_cells:
        p["i"] = np.sort(p["i"])
    for c in ("ux", "uy", "uz"):
        p[c] = (rng.standard_normal(n) * u_scale).astype(np.float32)
    p["q"] = np.float32(q) * rng.uniform(0.5, 1.5, n).astype(np.float32)
    p["tag"] = np.arange(n)
    return p


def acc_copies(nv, n_pipeline):
    stride = (nv + 1) & ~1
    return np.zeros((1 + n_pipeline) * stride, L.accumulator_t), stride


def main():
    out = {}
    rng = np.random.default_rng(20260101)
    npipe = pyref.n_pipeline()
    out["n_pipeline"] = np.int32(npipe)

    # ---- K1 load_interpolator: 6x5x4 -------------------------------------------------------
    nx, ny, nz = 6, 5, 4
    nv = L.nv(nx, ny, nz)
    dt = np.float32(0.3)
    g = pyref.new_periodic_grid(nx, ny, nz, 6.0, 5.0, 4.0, dt)
    out["k1_dims"] = np.array([nx, ny, nz], np.int32)
    f = rand_field(rng, nv)
    fi = np.zeros(nv, L.interpolator_t)
    pyref.load_interpolator(fi, f, g)
    out["k1_f"], out["k1_fi"] = f, fi

    # ---- K2 advance_p in-cell only, K6 energy_p ---------------------------------------------
    npart = 3001   # not a multiple of 16: exercises the host's straggler pass
    p = rand_particles(rng, npart, nx, ny, nz, u_scale=0.02)
    p["dx"] *= 0.9; p["dy"] *= 0.9; p["dz"] *= 0.9
    fi2 = fi.copy()
    for n in fi2.dtype.names:
        if n != "_pad":
            fi2[n] *= np.float32(0.05)
    out["k2_p_in"], out["k2_fi"] = p.copy(), fi2
    out["k2_q_m"] = np.float32(-1.0)
    out["k6_energy_p"] = np.float64(pyref.energy_p(p, npart, -1.0, fi2, g))
    a, stride = acc_copies(nv, npipe)
    pm = np.zeros(64, L.particle_mover_t)
    nm = pyref.advance_p(p, npart, -1.0, pm, a, fi2, g)
    assert nm == 0, nm
    pyref.reduce_accumulators(a, g)
    out["k2_p_out"], out["k2_a_out"] = p.copy(), a[:nv].copy()

    # ---- K3 advance_p with crossings --------------------------------------------------------
    # (a) periodic: every crossing wraps locally, nm == 0
    p = rand_particles(rng, npart, nx, ny, nz, u_scale=1.5)
    out["k3_p_in"] = p.copy()
    a, _ = acc_copies(nv, npipe)
    pm = np.zeros(4096, L.particle_mover_t)
    nm = pyref.advance_p(p, npart, -1.0, pm, a, fi2, g)
    assert nm == 0
    pyref.reduce_accumulators(a, g)
    out["k3a_p_out"], out["k3a_a_out"] = p.copy(), a[:nv].copy()
    # (b) reflecting z faces (PEC fields), absorbing x faces -> movers, periodic y
    gb = pyref.new_periodic_grid(nx, ny, nz, 6.0, 5.0, 4.0, dt)
    for face in (2, 5):
        pyref.set_face_bc(gb, face, L.PEC_FIELDS, L.REFLECT_PARTICLES)
    for face in (0, 3):
        pyref.set_face_bc(gb, face, L.ABSORB_FIELDS, L.ABSORB_PARTICLES)
    p = out["k3_p_in"].copy()
    a, _ = acc_copies(nv, npipe)
    pm = np.zeros(4096, L.particle_mover_t)
    nm = pyref.advance_p(p, npart, -1.0, pm, a, fi2, gb)
    assert nm > 0
    pyref.reduce_accumulators(a, gb)
    out["k3b_p_out"], out["k3b_a_out"], out["k3b_pm"] = p.copy(), a[:nv].copy(), pm[:nm].copy()
    out["k3b_fbc"] = np.array([L.ABSORB_FIELDS, 0, L.PEC_FIELDS, L.ABSORB_FIELDS, 0, L.PEC_FIELDS], np.int32)
    out["k3b_pbc"] = np.array([L.ABSORB_PARTICLES, 0, L.REFLECT_PARTICLES, L.ABSORB_PARTICLES, 0, L.REFLECT_PARTICLES], np.int32)

    # ---- K4 clear_jf + unload_accumulator + synchronize_jf (periodic) -----------------------
    f4 = rand_field(rng, nv)
    a4 = np.zeros(nv, L.accumulator_t)
    ix = np.arange(nv)
    xx, yy, zz = ix % (nx + 2), (ix // (nx + 2)) % (ny + 2), ix // ((nx + 2) * (ny + 2))
    interior = (xx >= 1) & (xx <= nx) & (yy >= 1) & (yy <= ny) & (zz >= 1) & (zz <= nz)
    for n in ("jx", "jy", "jz"):
        a4[n][interior] = rng.standard_normal((interior.sum(), 4)).astype(np.float32)
    out["k4_f_in"], out["k4_a"] = f4.copy(), a4
    pyref.clear_jf(f4, g)
    pyref.unload_accumulator(f4, a4, g)
    out["k4_f_unloaded"] = f4.copy()
    pyref.synchronize_jf(f4, g)
    out["k4_f_synced"] = f4.copy()

    # ---- K5 advance_b / advance_e ----------------------------------------------------------
    m = pyref.vacuum_coefficients(g)
    f5 = rand_field(rng, nv)
    out["k5_f_in"] = f5.copy()
    pyref.advance_b(f5, g, 0.5)
    out["k5_f_b"] = f5.copy()
    pyref.advance_e(f5, m, g)
    out["k5_f_e"] = f5.copy()
    out["k6_energy_f"] = pyref.energy_f(f5, m, g)
    # damped, PEC in z, periodic x,y
    gd = pyref.new_periodic_grid(nx, ny, nz, 6.0, 5.0, 4.0, dt, damp=0.01)
    for face in (2, 5):
        pyref.set_face_bc(gd, face, L.PEC_FIELDS, L.REFLECT_PARTICLES)
    md = pyref.vacuum_coefficients(gd)
    f5d = out["k5_f_in"].copy()
    pyref.advance_b(f5d, gd, 0.5)
    pyref.advance_e(f5d, md, gd)
    pyref.advance_b(f5d, gd, 0.5)
    out["k5d_f_out"] = f5d.copy()
    f5j = out["k4_f_unloaded"].copy()
    pyref.synchronize_jf(f5j, gd)
    out["k5d_f_jf_synced"] = f5j

    # ---- K8 center_p / uncenter_p ---------------------------------------
    p = out["k3_p_in"].copy()
    p["ux"] *= 0.2; p["uy"] *= 0.2; p["uz"] *= 0.2
    out["k8_p_in"] = p.copy()
    fi8 = fi.copy()                                    # the un-scaled K1 interpolator: O(1) fields
    out["k8_fi"] = fi8
    pyref.uncenter_p(p, len(p), -1.0, fi8, g)
    out["k8_p_uncentered"] = p.copy()
    pyref.center_p(p, len(p), -1.0, fi8, g)
    out["k8_p_recentered"] = p.copy()

    # ---- K7 sort_p -------------------------------------------------------------------------
    p = rand_particles(rng, 2000, nx, ny, nz, u_scale=0.1)
    out["k7_p_in"] = p.copy()
    ps, part = pyref.sort_p(p, len(p), g, nv, 1, L.particle_t)
    out["k7_p_oop"], out["k7_partition"] = ps, part
    ps, part2 = pyref.sort_p(p, len(p), g, nv, 0, L.particle_t)
    out["k7_p_inplace"] = ps
    assert np.array_equal(part, part2)

    # ---- multi-step single-domain trajectory (kernels chained like advance.cxx) -------------
    # 8x8x8 periodic, 2 species, 20 steps: energies per step + final state
    nx = ny = nz = 8
    nv = L.nv(nx, ny, nz)
    dtc = np.float32(0.95 / np.sqrt(3.0))
    g8 = pyref.new_periodic_grid(nx, ny, nz, 8.0, 8.0, 8.0, dtc)
    m8 = pyref.vacuum_coefficients(g8)
    f = np.zeros(nv, L.field_t)
    fi = np.zeros(nv, L.interpolator_t)
    species = []
    for s, drift in enumerate((0.2, -0.2)):
        n = 8 * 8 * 8 * 6
        p = rand_particles(rng, n, nx, ny, nz, u_scale=0.05, sorted_cells=True)
        p["ux"] += np.float32(drift)
        p["q"] = np.float32(-1.0 / 12)
        species.append(dict(p=p, np=n, q_m=-1.0, pm=np.zeros(n // 4, L.particle_mover_t)))
    out["t_dims"] = np.array([nx, ny, nz], np.int32)
    out["t_dt"] = dtc
    out["t_p0_in"], out["t_p1_in"] = species[0]["p"].copy(), species[1]["p"].copy()
    a, _ = acc_copies(nv, npipe)
    nsteps = 20
    en = np.zeros((nsteps, 8))
    pyref.load_interpolator(fi, f, g8)
    for step in range(nsteps):
        pyref.clear_accumulators(a, g8)
        for sp in species:
            nm = pyref.advance_p(sp["p"], sp["np"], sp["q_m"], sp["pm"], a, fi, g8)
            assert nm == 0
        pyref.reduce_accumulators(a, g8)
        pyref.clear_jf(f, g8)
        pyref.unload_accumulator(f, a, g8)
        pyref.synchronize_jf(f, g8)
        pyref.advance_b(f, g8, 0.5)
        pyref.advance_e(f, m8, g8)
        pyref.advance_b(f, g8, 0.5)
        pyref.load_interpolator(fi, f, g8)
        en[step, :6] = pyref.energy_f(f, m8, g8)
        for s, sp in enumerate(species):
            en[step, 6 + s] = pyref.energy_p(sp["p"], sp["np"], sp["q_m"], fi, g8)
    out["t_energies"] = en
    out["t_f_out"] = f
    out["t_p0_out"], out["t_p1_out"] = species[0]["p"], species[1]["p"]

    # ---- K9 divergence cleaning family and charge densities ----------------
    # own generator, so that adding cases here leaves the vectors above as they were
    rng9 = np.random.default_rng(20260909)
    nx, ny, nz = 6, 5, 4
    nv = L.nv(nx, ny, nz)
    p9 = rand_particles(rng9, 1500, nx, ny, nz, u_scale=0.1)
    out["k9_p"] = p9
    mb = pyref.vacuum_coefficients(gb)
    for tag, gg, mm in (("per", g, m), ("pec", gd, md), ("abs", gb, mb)):
        if tag == "abs":
            rng9 = np.random.default_rng(20260911)          # added later: leaves the K10/K11 draws as they were
        f9 = rand_field(rng9, nv)
        out[f"k9{tag}_f_in"] = f9.copy()
        pyref.clear_rhof(f9, gg)
        pyref.accumulate_rho_p(f9, p9, len(p9), gg)
        out[f"k9{tag}_f_rho_p"] = f9.copy()
        pyref.synchronize_rho(f9, gg)
        out[f"k9{tag}_f_rho_sync"] = f9.copy()
        pyref.compute_rhob(f9, mm, gg)
        out[f"k9{tag}_f_rhob"] = f9.copy()
        f9["rhob"] *= np.float32(0.9)                      # otherwise div_e_err is pure round-off
        pyref.compute_div_e_err(f9, mm, gg)
        out[f"k9{tag}_f_div_e"] = f9.copy()
        out[f"k9{tag}_rms_div_e"] = np.float64(pyref.compute_rms_div_e_err(f9, gg))
        pyref.clean_div_e(f9, mm, gg)
        out[f"k9{tag}_f_clean_e"] = f9.copy()
        pyref.compute_div_b_err(f9, gg)
        out[f"k9{tag}_f_div_b"] = f9.copy()
        out[f"k9{tag}_rms_div_b"] = np.float64(pyref.compute_rms_div_b_err(f9, gg))
        pyref.clean_div_b(f9, gg)
        out[f"k9{tag}_f_clean_b"] = f9.copy()
        pyref.compute_curl_b(f9, mm, gg)
        out[f"k9{tag}_f_curl_b"] = f9.copy()
        out[f"k9{tag}_sync_err"] = np.float64(pyref.synchronize_tang_e_norm_b(f9, gg))
        out[f"k9{tag}_f_sync"] = f9.copy()

    rng9 = np.random.default_rng(20260910)
    # ---- K12 absorbing (Higdon) field boundary on x, PEC on z: two field steps ----------------------
    f12 = out["k5_f_in"].copy()
    for _ in range(2):
        pyref.advance_b(f12, gb, 0.5)
        pyref.advance_e(f12, mb, gb)
        pyref.advance_b(f12, gb, 0.5)
    out["k12_f_out"] = f12

    # ---- K10 hydro moments -----------------------------------------------------
    p10 = rand_particles(rng9, 1200, nx, ny, nz, u_scale=0.6)
    out["k10_p"] = p10
    for tag, gg in (("per", g), ("pec", gd)):
        h = np.zeros(nv, L.hydro_t)
        h["jx"] = 7.0                                       # clear_hydro must wipe it
        pyref.clear_hydro(h, gg)
        pyref.accumulate_hydro_p(h, p10, len(p10), -1.0, out["k8_fi"], gg)
        out[f"k10{tag}_h_acc"] = h.copy()
        pyref.synchronize_hydro(h, gg)
        out[f"k10{tag}_h_sync"] = h.copy()

    # ---- K11 boundary_p on one rank with absorbing x walls, and move_p called directly -------------
    f11 = rand_field(rng9, nv)
    out["k11_f_in"] = f11.copy()
    a11 = np.zeros(2 * ((nv + 1) & ~1), L.accumulator_t)
    nm11 = len(out["k3b_pm"])
    out["k11_p_out"] = pyref.boundary_p(out["k3b_p_out"].copy(), len(out["k3b_p_out"]), out["k3b_pm"].copy(), nm11, f11, a11, gb, L)
    out["k11_f_out"] = f11.copy()
    assert not np.any(a11["jx"]) and len(out["k11_p_out"]) == len(out["k3b_p_out"]) - nm11
    # move_p: a handful of the K3 particles pushed by hand across faces (periodic grid: returns 0;
    # absorbing/reflecting grid: some return 1)
    for tag, gg in (("per", g), ("abs", gb)):
        p = out["k3_p_in"][:64].copy()
        a = np.zeros(2 * ((nv + 1) & ~1), L.accumulator_t)
        pm = np.zeros(64, L.particle_mover_t)
        pm["i"] = np.arange(64)
        pm["dispx"], pm["dispy"], pm["dispz"] = [rng9.uniform(-1.4, 1.4, 64).astype(np.float32) for _ in range(3)]
        out[f"k11{tag}_pm_in"] = pm.copy()
        ret = np.zeros(64, np.int32)
        for k in range(64):
            ret[k] = pyref.move_p(p, pm[k:k + 1], a, gg)
        out[f"k11{tag}_p_out"], out[f"k11{tag}_pm_out"], out[f"k11{tag}_ret"], out[f"k11{tag}_a_out"] = p, pm, ret, a[:nv].copy()
    assert out["k11abs_ret"].sum() > 0 and out["k11per_ret"].sum() == 0

    # ---- K13 several materials: an anisotropic dielectric / magnetic one and an anisotropic conductor next
    # to vacuum, ids drawn per voxel and component (advance_e.c:8-25, sfa.c:145-177, energy_f.c:50-82,
    # compute_div_e_err.c, compute_rhob.c, clean_div_e.c): the coefficient table and a chain of field ops
    rng13 = np.random.default_rng(20261013)
    props = np.array([[1, 1, 1, 1, 1, 1, 0, 0, 0],
                      [2.5, 1.5, 3.0, 1.2, 0.8, 1.0, 0, 0, 0],
                      [1.0, 1.3, 0.9, 1.0, 1.0, 1.1, 0.7, 0.3, 1.1]], np.float32)
    out["k13_props"] = props
    for tag, gg in (("per", g), ("pec", gd)):
        ptr, table = pyref.material_coefficients(gg, props)
        table["pad"] = 0                                    # malloc'ed, never written or read by the reference
        out[f"k13{tag}_mc"] = table
        f13 = rand_field(rng13, nv)
        for c in ("ematx", "ematy", "ematz", "nmat", "fmatx", "fmaty", "fmatz", "cmat"):
            f13[c] = rng13.integers(0, 3, nv)
        out[f"k13{tag}_f_in"] = f13.copy()
        pyref.compute_curl_b(f13, ptr, gg)
        out[f"k13{tag}_f_curl_b"] = f13.copy()
        pyref.advance_b(f13, gg, 0.5)
        pyref.advance_e(f13, ptr, gg)
        out[f"k13{tag}_f_e"] = f13.copy()
        out[f"k13{tag}_en"] = pyref.energy_f(f13, ptr, gg)
        pyref.compute_rhob(f13, ptr, gg)
        out[f"k13{tag}_f_rhob"] = f13.copy()
        f13["rhob"] *= np.float32(0.9)
        pyref.compute_div_e_err(f13, ptr, gg)
        out[f"k13{tag}_f_div_e"] = f13.copy()
        out[f"k13{tag}_rms_div_e"] = np.float64(pyref.compute_rms_div_e_err(f13, gg))
        pyref.clean_div_e(f13, ptr, gg)
        out[f"k13{tag}_f_clean_e"] = f13.copy()

    dst = os.path.join(ROOT, "tests", "golden", "kernels.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB;", len(out), "arrays; reference n_pipeline =", npipe)


if __name__ == "__main__":
    main()
