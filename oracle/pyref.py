"""ctypes binding of oracle/_ref/libvpic_ref.so -- the reference's own scalar sources compiled
by oracle/Makefile (TEST INFRASTRUCTURE; container-only, /root/reference does not travel)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "libvpic_ref.so")
_lib = None


def available():
    return os.path.exists(SO) or os.path.isdir("/root/reference/src")


def lib(tpp=1):
    """Load and boot the reference (thread.boot/serial.boot/mp_init as src/main.cxx:72-79)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-s", "-C", HERE, "ref", "-j8"])
        _lib = C.CDLL(SO)
        _lib.ref_new_periodic_grid.restype = C.c_void_p
        _lib.ref_new_periodic_grid.argtypes = [C.c_float] * 4 + [C.c_double] * 3 + [C.c_int] * 3
        _lib.ref_new_vacuum_coefficients.restype = C.c_void_p
        _lib.ref_new_vacuum_coefficients.argtypes = [C.c_void_p]
        _lib.energy_p.restype = C.c_double
        for n in ("ref_synchronize_tang_e_norm_b", "ref_compute_rms_div_e_err", "ref_compute_rms_div_b_err"):
            getattr(_lib, n).restype = C.c_double
        _lib.ref_new_species.restype = C.c_void_p
        for n in ("ref_species_p", "ref_species_pm", "ref_species_partition"):
            getattr(_lib, n).restype = C.c_void_p
        _lib.ref_boot(tpp)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def V(h):
    return C.c_void_p(h)


def new_periodic_grid(nx, ny, nz, lx, ly, lz, dt, cvac=1.0, eps0=1.0, damp=0.0):
    return lib().ref_new_periodic_grid(dt, cvac, eps0, damp, lx, ly, lz, nx, ny, nz)


def set_face_bc(g, face, fbc, pbc):
    lib().ref_set_face_bc(V(g), face, fbc, pbc)


def grid_info(g):
    out = (C.c_float * 10)()
    n = (C.c_int * 3)()
    lib().ref_grid_info(V(g), out, n)
    return list(out), list(n)


def n_pipeline():
    return lib().ref_n_pipeline()


def vacuum_coefficients(g):
    return lib().ref_new_vacuum_coefficients(V(g))


def material_coefficients(g, props):
    """props: [n, 9] (eps xyz, mu xyz, sigma xyz).  Returns (pointer for the calls below, numpy copy)."""
    from importlib import import_module
    L = import_module("old-vpic_amd.layout")
    props = np.ascontiguousarray(props, np.float32)
    l = lib()
    l.ref_new_coefficients.restype = C.c_void_p
    l.ref_new_coefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    ptr = l.ref_new_coefficients(V(g), props.ctypes.data_as(C.c_void_p), len(props))
    table = np.frombuffer((C.c_char * (len(props) * L.material_coefficient_t.itemsize)).from_address(ptr), L.material_coefficient_t).copy()
    return ptr, table


def load_interpolator(fi, f, g):
    lib().load_interpolator(_p(fi), _p(f), V(g))


def clear_accumulators(a, g):
    lib().clear_accumulators(_p(a), V(g))


def reduce_accumulators(a, g):
    lib().reduce_accumulators(_p(a), V(g))


def unload_accumulator(f, a, g):
    lib().unload_accumulator(_p(f), _p(a), V(g))


def advance_p(p, np_, q_m, pm, a, fi, g):
    return lib().advance_p(_p(p), int(np_), C.c_float(q_m), _p(pm), len(pm), _p(a), _p(fi), V(g))


def center_p(p, np_, q_m, fi, g):
    lib().center_p(_p(p), int(np_), C.c_float(q_m), _p(fi), V(g))


def uncenter_p(p, np_, q_m, fi, g):
    lib().uncenter_p(_p(p), int(np_), C.c_float(q_m), _p(fi), V(g))


def energy_p(p, np_, q_m, fi, g):
    return lib().energy_p(_p(p), int(np_), C.c_float(q_m), _p(fi), V(g))


def energy_f(f, m, g):
    en = np.zeros(6, np.float64)
    lib().ref_energy_f(_p(en), _p(f), V(m), V(g))
    return en


def clear_jf(f, g):
    lib().ref_clear_jf(_p(f), V(g))


def synchronize_jf(f, g):
    lib().ref_synchronize_jf(_p(f), V(g))


def advance_b(f, g, frac):
    lib().ref_advance_b(_p(f), V(g), C.c_float(frac))


def advance_e(f, m, g):
    lib().ref_advance_e(_p(f), V(m), V(g))


def clear_rhof(f, g):
    lib().ref_clear_rhof(_p(f), V(g))


def accumulate_rho_p(f, p, np_, g):
    lib().accumulate_rho_p(_p(f), _p(p), int(np_), V(g))


def synchronize_rho(f, g):
    lib().ref_synchronize_rho(_p(f), V(g))


def compute_rhob(f, m, g):
    lib().ref_compute_rhob(_p(f), V(m), V(g))


def compute_curl_b(f, m, g):
    lib().ref_compute_curl_b(_p(f), V(m), V(g))


def synchronize_tang_e_norm_b(f, g):
    return lib().ref_synchronize_tang_e_norm_b(_p(f), V(g))


def compute_div_e_err(f, m, g):
    lib().ref_compute_div_e_err(_p(f), V(m), V(g))


def compute_rms_div_e_err(f, g):
    return lib().ref_compute_rms_div_e_err(_p(f), V(g))


def clean_div_e(f, m, g):
    lib().ref_clean_div_e(_p(f), V(m), V(g))


def compute_div_b_err(f, g):
    lib().ref_compute_div_b_err(_p(f), V(g))


def compute_rms_div_b_err(f, g):
    return lib().ref_compute_rms_div_b_err(_p(f), V(g))


def clean_div_b(f, g):
    lib().ref_clean_div_b(_p(f), V(g))


def clear_hydro(h, g):
    lib().clear_hydro(_p(h), V(g))


def accumulate_hydro_p(h, p, np_, q_m, fi, g):
    lib().accumulate_hydro_p(_p(h), _p(p), int(np_), C.c_float(q_m), _p(fi), V(g))


def synchronize_hydro(h, g):
    lib().synchronize_hydro(_p(h), V(g))


def boundary_p(p, np_, pm, nm, f, a, g, L_):
    """The reference's boundary_p on one reference-owned species; returns the surviving particles."""
    l = lib()
    sp = l.ref_new_species(C.c_float(-1.0), int(np_), max(int(nm), 16), 1)
    pt, mt = L_.particle_t, L_.particle_mover_t
    C.memmove((C.c_char * (pt.itemsize * np_)).from_address(l.ref_species_p(V(sp))), p.ctypes.data, pt.itemsize * np_)
    C.memmove((C.c_char * (mt.itemsize * nm)).from_address(l.ref_species_pm(V(sp))), pm.ctypes.data, mt.itemsize * nm)
    l.ref_species_set_counts(V(sp), int(np_), int(nm))
    l.boundary_p(V(sp), _p(f), _p(a), V(g), None)
    n = l.ref_species_np(V(sp))
    assert l.ref_species_nm(V(sp)) == 0
    return np.frombuffer((C.c_char * (pt.itemsize * n)).from_address(l.ref_species_p(V(sp))), dtype=pt).copy()


def move_p(p, pm1, a, g):
    """The reference's move_p on particle pm1['i'] of p (in place); returns its return value."""
    return lib().move_p(_p(p), _p(pm1), _p(a), V(g))


def sort_p(p, np_, g, nv, out_of_place, dtype):
    """Runs the reference's sort_p on a reference-owned species; returns (sorted p, partition)."""
    l = lib()
    sp = l.ref_new_species(C.c_float(-1.0), int(np_), 16, out_of_place)
    dst = (C.c_char * (dtype.itemsize * np_)).from_address(l.ref_species_p(V(sp)))
    C.memmove(dst, p.ctypes.data, dtype.itemsize * np_)
    l.ref_species_set_counts(V(sp), int(np_), 0)
    l.sort_p(V(sp), V(g))
    out = np.frombuffer((C.c_char * (dtype.itemsize * np_)).from_address(l.ref_species_p(V(sp))), dtype=dtype).copy()
    part = np.frombuffer((C.c_int * (nv + 1)).from_address(l.ref_species_partition(V(sp))), dtype=np.int32).copy()
    return out, part
