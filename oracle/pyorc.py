"""ctypes binding of oracle/libvpic_oracle.so (TEST INFRASTRUCTURE, see vpic_oracle.h)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
L = importlib.import_module("old-vpic_amd.layout")


class Grid(C.Structure):
    """orc_grid_t"""
    _fields_ = [("dt", C.c_float), ("cvac", C.c_float), ("eps0", C.c_float), ("damp", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float),
                ("rdx", C.c_float), ("rdy", C.c_float), ("rdz", C.c_float),
                ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("fbc", C.c_int * 6), ("pbc", C.c_int * 6), ("rank", C.c_int)]

    @property
    def nv(self):
        return (self.nx + 2) * (self.ny + 2) * (self.nz + 2)


def make_grid(nx, ny, nz, lx, ly, lz, dt, cvac=1.0, eps0=1.0, damp=0.0, fbc=None, pbc=None, rank=0):
    """Cell sizes the way partition_periodic_box computes them (src/grid/partition.c:60-66):
    double arithmetic, then stored as float."""
    g = Grid()
    g.dt, g.cvac, g.eps0, g.damp = dt, cvac, eps0, damp
    g.dx, g.dy, g.dz = lx / nx, ly / ny, lz / nz
    g.rdx, g.rdy, g.rdz = nx / lx, ny / ly, nz / lz
    g.nx, g.ny, g.nz = nx, ny, nz
    g.rank = rank
    for f in range(6):
        g.fbc[f] = rank if fbc is None else fbc[f]
        g.pbc[f] = rank if pbc is None else pbc[f]
    return g


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libvpic_oracle.so")
        src = os.path.join(HERE, "vpic_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
        _lib = C.CDLL(so)
        _lib.orc_energy_p.restype = C.c_double
        _lib.orc_synchronize_tang_e_norm_b_local.restype = C.c_double
        _lib.orc_synchronize_tang_e_norm_b_self.restype = C.c_double
        _lib.orc_unpack_msg.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_interpolator(fi, f, g):
    lib().orc_load_interpolator(_p(fi), _p(f), C.byref(g))


def clear_accumulators(a, g, n_pipeline=0):
    lib().orc_clear_accumulators(_p(a), C.byref(g), n_pipeline)


def reduce_accumulators(a, g, n_pipeline):
    lib().orc_reduce_accumulators(_p(a), C.byref(g), n_pipeline)


def unload_accumulator(f, a, g):
    lib().orc_unload_accumulator(_p(f), _p(a), C.byref(g))


def advance_p(p, np_, q_m, pm, a, fi, g, n_pipeline=0):
    return lib().orc_advance_p(_p(p), int(np_), C.c_float(q_m), _p(pm), len(pm), _p(a), _p(fi), C.byref(g), n_pipeline)


def center_p(p, np_, q_m, fi, g):
    lib().orc_center_p(_p(p), int(np_), C.c_float(q_m), _p(fi), C.byref(g))


def uncenter_p(p, np_, q_m, fi, g):
    lib().orc_uncenter_p(_p(p), int(np_), C.c_float(q_m), _p(fi), C.byref(g))


def sort_p(p, np_, partition, g, out_of_place=1):
    lib().orc_sort_p(_p(p), int(np_), _p(partition), C.byref(g), out_of_place)


def energy_p(p, np_, q_m, fi, g):
    return lib().orc_energy_p(_p(p), int(np_), C.c_float(q_m), _p(fi), C.byref(g))


def energy_f(f, m, g):
    en = np.zeros(6, np.float64)
    lib().orc_energy_f(_p(en), _p(f), _p(m), C.byref(g))
    return en


def vacuum_coefficients():
    m = np.zeros(1, L.material_coefficient_t)
    lib().orc_vacuum_coefficients(_p(m))
    return m


def material_coefficients(props, dt, eps0=1.0):
    """props: [n, 9] floats (eps xyz, mu xyz, sigma xyz) -> n coefficient records (sfa.c:145-177)."""
    props = np.ascontiguousarray(props, np.float32)
    m = np.zeros(len(props), L.material_coefficient_t)
    for k in range(len(props)):
        lib().orc_material_coefficients(C.c_void_p(m.ctypes.data + k * m.itemsize), props[k].ctypes.data_as(C.c_void_p),
                                        C.c_float(dt), C.c_float(eps0))
    return m


def clear_jf(f, g):
    lib().orc_clear_jf(_p(f), C.byref(g))


def advance_b(f, g, frac):
    lib().orc_advance_b(_p(f), C.byref(g), C.c_float(frac))


def advance_e(f, m, g):
    lib().orc_advance_e(_p(f), _p(m), C.byref(g))


def clear_rhof(f, g):
    lib().orc_clear_rhof(_p(f), C.byref(g))


def accumulate_rho_p(f, p, np_, g):
    lib().orc_accumulate_rho_p(_p(f), _p(p), int(np_), C.byref(g))


def synchronize_rho_local(f, g):
    lib().orc_synchronize_rho_local(_p(f), C.byref(g))


def accumulate_rhob(f, p, g):
    """boundary_p.c:9-71 for every particle of p."""
    for k in range(len(p)):
        lib().orc_accumulate_rhob(_p(f), C.c_void_p(p.ctypes.data + k * p.itemsize), C.byref(g))


def compute_rhob(f, m, g):
    lib().orc_compute_rhob(_p(f), _p(m), C.byref(g))


def compute_curl_b(f, m, g):
    lib().orc_compute_curl_b(_p(f), _p(m), C.byref(g))


def synchronize_tang_e_norm_b_local(f, g):
    return lib().orc_synchronize_tang_e_norm_b_local(_p(f), C.byref(g))


def compute_div_e_err(f, m, g):
    lib().orc_compute_div_e_err(_p(f), _p(m), C.byref(g))


def _rms(fn, f, g):
    s = np.zeros(2, np.float64)
    fn(_p(s), _p(f), C.byref(g))
    return float(g.eps0 * np.sqrt(s[0] / s[1]))          # compute_rms_div_e_err.c:156-159 on one domain


def compute_rms_div_e_err(f, g):
    return _rms(lib().orc_rms_div_e_err_local, f, g)


def clean_div_e(f, m, g):
    lib().orc_clean_div_e(_p(f), _p(m), C.byref(g))


def compute_div_b_err(f, g):
    lib().orc_compute_div_b_err(_p(f), C.byref(g))


def compute_rms_div_b_err(f, g):
    return _rms(lib().orc_rms_div_b_err_local, f, g)


def clean_div_b(f, g):
    lib().orc_clean_div_b(_p(f), C.byref(g))


def clear_hydro(h, g):
    lib().orc_clear_hydro(_p(h), C.byref(g))


def accumulate_hydro_p(h, p, np_, q_m, fi, g):
    lib().orc_accumulate_hydro_p(_p(h), _p(p), int(np_), C.c_float(q_m), _p(fi), C.byref(g))


def synchronize_hydro_local(h, g):
    lib().orc_synchronize_hydro_local(_p(h), C.byref(g))


def local_adjust_rho(f, g):
    lib().orc_local_adjust_rho(_p(f), C.byref(g))


def synchronize_rho_self(f, g, axis):
    lib().orc_synchronize_rho_self(_p(f), C.byref(g), axis)


def rho_count(g, d):
    return lib().orc_rho_count(C.byref(g), d)


def pack_rho(f, g, d):
    buf = np.zeros(rho_count(g, d), np.float32)
    lib().orc_pack_rho(_p(buf), _p(f), C.byref(g), d)
    return buf


def unpack_rho(f, buf, g, d):
    lib().orc_unpack_rho(_p(f), _p(np.ascontiguousarray(buf, np.float32)), C.byref(g), d)


def msg_count(g, kind, d):
    return lib().orc_msg_count(C.byref(g), kind, d)


def pack_msg(f, g, kind, d):
    buf = np.zeros(msg_count(g, kind, d), np.float32)
    lib().orc_pack_msg(_p(buf), _p(f), C.byref(g), kind, d)
    return buf


def unpack_msg(f, buf, g, kind, d):
    return lib().orc_unpack_msg(_p(f), _p(np.ascontiguousarray(buf, np.float32)), C.byref(g), kind, d)


def local_adjust_tang_e_norm_b(f, g):
    lib().orc_local_adjust_tang_e_norm_b(_p(f), C.byref(g))


def synchronize_tang_e_norm_b_self(f, g, axis):
    return lib().orc_synchronize_tang_e_norm_b_self(_p(f), C.byref(g), axis)


def rms_local(f, g, which):
    s = np.zeros(2, np.float64)
    (lib().orc_rms_div_e_err_local if which == "e" else lib().orc_rms_div_b_err_local)(_p(s), _p(f), C.byref(g))
    return float(s[0]), float(s[1])


def synchronize_jf_local(f, g):
    lib().orc_synchronize_jf_local(_p(f), C.byref(g))


def synchronize_jf_self(f, g, axis):
    """One axis of synchronize_jf for faces the domain shares with itself (remote.c:477-500)."""
    if g.fbc[axis] != g.rank or g.fbc[axis + 3] != g.rank:
        return
    lo, hi = pack_jf(f, g, axis), pack_jf(f, g, axis + 3)
    unpack_jf(f, lo, g, axis)
    unpack_jf(f, hi, g, axis + 3)


def tang_b_count(g, d):
    return lib().orc_tang_b_count(C.byref(g), d)


def pack_tang_b(f, g, d):
    buf = np.zeros(tang_b_count(g, d), np.float32)
    lib().orc_pack_tang_b(_p(buf), _p(f), C.byref(g), d)
    return buf


def unpack_tang_b(f, buf, g, d):
    lib().orc_unpack_tang_b(_p(f), _p(buf), C.byref(g), d)


def pack_jf(f, g, d):
    buf = np.zeros(tang_b_count(g, d), np.float32)
    lib().orc_pack_jf(_p(buf), _p(f), C.byref(g), d)
    return buf


def unpack_jf(f, buf, g, d):
    lib().orc_unpack_jf(_p(f), _p(buf), C.byref(g), d)


def local_adjust_jf(f, g):
    lib().orc_local_adjust_jf(_p(f), C.byref(g))


def move_p(p, pm1, a, g):
    """orc_move_p on particle pm1['i'] of p (in place); returns 1 when it stopped on a face."""
    return lib().orc_move_p(_p(p), _p(pm1), _p(a), C.byref(g))


def boundary_p_pack(p, np_, pm, nm, sp_id, f, g, cap):
    outs = [np.zeros(cap, L.particle_injector_t) for _ in range(6)]
    ptrs = (C.c_void_p * 6)(*[o.ctypes.data for o in outs])
    ns = (C.c_int * 6)()
    new_np = lib().orc_boundary_p_pack(_p(p), int(np_), _p(pm), int(nm), sp_id, _p(f), C.byref(g), ptrs, ns, cap)
    return new_np, [o[:ns[k]] for k, o in enumerate(outs)]


def maxwellian_reflux(draws, p, pm, g, ut_para, ut_perp, face, sp_id=0):
    """orc_maxwellian_reflux for mover record pm (one element) of particle array p; returns the injector record."""
    out = np.zeros(1, L.particle_injector_t)
    d = np.ascontiguousarray(draws, np.float32)
    part = p[int(pm["i"][0]):int(pm["i"][0]) + 1]
    lib().orc_maxwellian_reflux(_p(d), _p(part), _p(pm), C.byref(g), C.c_float(ut_para), C.c_float(ut_perp), int(face), int(sp_id), _p(out))
    return out[0]


def child_langmuir(p, np_, pm, nm, component, n_emit, ut_perp, ut_para, q_m, fi, f, a, g, draws):
    """orc_child_langmuir; returns (new np, new nm)."""
    nm_c = C.c_int(nm)
    comp = np.ascontiguousarray(component, np.int32)
    d = np.ascontiguousarray(draws, np.float64)
    new_np = lib().orc_child_langmuir(_p(p), int(np_), len(p), _p(pm), C.byref(nm_c), len(pm), _p(comp), len(comp), int(n_emit),
                                      C.c_float(ut_perp), C.c_float(ut_para), C.c_float(q_m), _p(fi), _p(f), _p(a), C.byref(g), _p(d))
    return new_np, nm_c.value


def boundary_p_inject(p, np_, pm, nm, inj, a, g):
    nm_c = C.c_int(nm)
    inj = np.ascontiguousarray(inj)
    new_np = lib().orc_boundary_p_inject(_p(p), int(np_), _p(pm), C.byref(nm_c), _p(inj), len(inj), _p(a), C.byref(g))
    return new_np, nm_c.value


def step(f, fi, a, m, species, g, sort=False, clean_e=False, clean_b=False, sync_shared=False):
    """One vpic_simulation::advance() of a single self-periodic / locally bounded domain
    (src/vpic/advance.cxx:38-214 without emitters, collisions, injection and div cleaning).
    species: list of dicts {p, np, q_m, pm, partition}.  Returns nothing; arrays updated in place."""
    clear_accumulators(a, g)
    for sp in species:
        if sort:
            sort_p(sp["p"], sp["np"], sp["partition"], g)
    for sp in species:
        sp["nm"] = advance_p(sp["p"], sp["np"], sp["q_m"], sp["pm"], a, fi, g)
    for k, sp in enumerate(species):
        if sp["nm"]:
            sp["np"], _ = boundary_p_pack(sp["p"], sp["np"], sp["pm"], sp["nm"], k, f, g, sp["nm"])
            sp["nm"] = 0
    clear_jf(f, g)
    unload_accumulator(f, a, g)
    synchronize_jf_local(f, g)
    advance_b(f, g, 0.5)
    advance_e(f, m, g)
    advance_b(f, g, 0.5)
    if clean_e:                                             # advance.cxx:151-173
        accumulate_rho(f, species, g)
        compute_div_e_err(f, m, g)
        if compute_rms_div_e_err(f, g) > 0:
            clean_div_e(f, m, g)
            compute_div_e_err(f, m, g)
            if compute_rms_div_e_err(f, g) > 0:
                clean_div_e(f, m, g)
    if clean_b:                                             # advance.cxx:177-195
        compute_div_b_err(f, g)
        if compute_rms_div_b_err(f, g) > 0:
            clean_div_b(f, g)
            compute_div_b_err(f, g)
            if compute_rms_div_b_err(f, g) > 0:
                clean_div_b(f, g)
    if sync_shared:                                         # advance.cxx:199-207
        synchronize_tang_e_norm_b_local(f, g)
    load_interpolator(fi, f, g)


def accumulate_rho(f, species, g):
    clear_rhof(f, g)
    for sp in species:
        accumulate_rho_p(f, sp["p"], sp["np"], g)
    synchronize_rho_local(f, g)


def initialize_fields(f, m, species, g):
    """initialize.cxx:32-76 on one domain."""
    synchronize_tang_e_norm_b_local(f, g)
    compute_div_b_err(f, g)
    clean_div_b(f, g)
    compute_curl_b(f, m, g)
    accumulate_rho(f, species, g)
    compute_rhob(f, m, g)
    compute_div_e_err(f, m, g)
    if compute_rms_div_e_err(f, g) > 0:
        clean_div_e(f, m, g)
    synchronize_tang_e_norm_b_local(f, g)
