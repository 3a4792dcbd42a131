"""ctypes view of the drop-in entry points (include/vpic_hip_dropin.h) and of the reference's grid_t.

reference_grid() builds a grid_t the way the reference's own helpers do for a single-rank box:
size_grid (src/grid/ops.c:25-133: bc[], range[], neighbor[] with reflecting faces), join_grid for
periodic faces (ops.c:135-182), set_fbc / set_pbc (ops.c:184-231)."""
import ctypes as C

import numpy as np

from . import layout as L
from ._lib import lib


class RefGrid(C.Structure):
    """grid_t (src/grid/grid.h:112-167)."""
    _fields_ = [("mp", C.c_void_p),
                ("dt", C.c_float), ("cvac", C.c_float), ("eps0", C.c_float), ("damp", C.c_float),
                ("x0", C.c_float), ("y0", C.c_float), ("z0", C.c_float),
                ("x1", C.c_float), ("y1", C.c_float), ("z1", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float),
                ("rdx", C.c_float), ("rdy", C.c_float), ("rdz", C.c_float),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("bc", C.c_int32 * 27),
                ("range", C.c_void_p), ("neighbor", C.c_void_p),
                ("rangel", C.c_int64), ("rangeh", C.c_int64),
                ("nb", C.c_int32), ("boundary", C.c_void_p)]


assert C.sizeof(RefGrid) == 240


def _boundary(i, j, k):
    return (i + 1) + 3 * ((j + 1) + 3 * (k + 1))


FACE_DIR = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 0, 0), (0, 1, 0), (0, 0, 1)]


def reference_grid(nx, ny, nz, lx, ly, lz, dt, cvac=1.0, eps0=1.0, damp=0.0, fbc=None, pbc=None):
    """Single-rank box.  fbc/pbc[f]: None or 0 = periodic onto this rank, else a local code."""
    g = RefGrid()
    g.dt, g.cvac, g.eps0, g.damp = dt, cvac, eps0, damp
    g.x0 = g.y0 = g.z0 = 0.0
    g.x1, g.y1, g.z1 = lx, ly, lz
    g.dx, g.dy, g.dz = lx / nx, ly / ny, lz / nz
    g.rdx, g.rdy, g.rdz = nx / lx, ny / ly, nz / lz
    g.nx, g.ny, g.nz = nx, ny, nz
    nv = L.nv(nx, ny, nz)
    for k in range(27):
        g.bc[k] = L.PEC_FIELDS                                   # ops.c:41-45
    g.bc[13] = 0
    rng = np.array([0, nv], np.int64)
    nb = np.full((nv, 6), L.REFLECT_PARTICLES, np.int64)         # ghosts and faces: reflect (ops.c:84-97)
    sy, sz = nx + 2, (nx + 2) * (ny + 2)
    x, y, z = np.meshgrid(np.arange(1, nx + 1), np.arange(1, ny + 1), np.arange(1, nz + 1), indexing="ij")
    v = (x + sy * y + sz * z).ravel()
    x, y, z = x.ravel(), y.ravel(), z.ravel()
    n = (nx, ny, nz)
    stride = (1, sy, sz)
    coord = (x, y, z)
    for f in range(6):
        a, hi = f % 3, f >= 3
        inside = coord[a] < n[a] if hi else coord[a] > 1
        nb[v[inside], f] = v[inside] + (stride[a] if hi else -stride[a])          # ops.c:77-82
        edge = ~inside
        fb = 0 if fbc is None else fbc[f]
        pb = 0 if pbc is None else pbc[f]
        g.bc[_boundary(*FACE_DIR[f])] = fb
        if pb >= 0:                                                                # join_grid: wrap (ops.c:157-171)
            nb[v[edge], f] = v[edge] + (-(n[a] - 1) if hi else (n[a] - 1)) * stride[a]
        else:
            nb[v[edge], f] = pb                                                    # set_pbc (ops.c:218-229)
    g._keep = (rng, nb)                                                            # keep the arrays alive
    g.range = rng.ctypes.data
    g.neighbor = nb.ctypes.data
    g.rangel, g.rangeh = 0, nv - 1
    g.nb, g.boundary = 0, None
    return g


def ref():
    """The library with argument types of the drop-in entry points set."""
    l = lib()
    l.vpic_hip_ref_advance_p.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    l.vpic_hip_ref_advance_p.restype = C.c_int
    l.vpic_hip_ref_energy_p.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    l.vpic_hip_ref_energy_p.restype = C.c_double
    l.vpic_hip_ref_advance_b.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    for n in ("load_interpolator", "unload_accumulator", "advance_e", "energy_f"):
        getattr(l, "vpic_hip_ref_" + n).argtypes = [C.c_void_p] * 3 if n != "energy_f" else [C.c_void_p] * 4
    for n in ("clear_accumulators", "reduce_accumulators", "clear_jf", "synchronize_jf", "sort_p", "clear_rhof", "synchronize_rho",
              "synchronize_tang_e_norm_b", "compute_rms_div_e_err", "compute_div_b_err", "compute_rms_div_b_err", "clean_div_b"):
        getattr(l, "vpic_hip_ref_" + n).argtypes = [C.c_void_p] * 2
    for n in ("compute_rhob", "compute_curl_b", "compute_div_e_err", "clean_div_e"):
        getattr(l, "vpic_hip_ref_" + n).argtypes = [C.c_void_p] * 3
    l.vpic_hip_ref_move_p.argtypes = [C.c_void_p] * 4
    l.vpic_hip_ref_move_p.restype = C.c_int
    l.vpic_hip_ref_boundary_p.argtypes = [C.c_void_p] * 5
    for n in ("clear_hydro", "synchronize_hydro", "local_adjust_hydro"):
        getattr(l, "vpic_hip_ref_" + n).argtypes = [C.c_void_p] * 2
    l.vpic_hip_ref_accumulate_hydro_p.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    l.vpic_hip_ref_accumulate_rho_p.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    l.vpic_hip_ref_accumulate_rhob.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for n in ("synchronize_tang_e_norm_b", "compute_rms_div_e_err", "compute_rms_div_b_err"):
        getattr(l, "vpic_hip_ref_" + n).restype = C.c_double
    return l


DROPIN_EXPORTS = """vpic_hip_ref_set_transport vpic_hip_ref_set_accumulator_copies vpic_hip_ref_set_material_count vpic_hip_ref_load_interpolator
vpic_hip_ref_clear_accumulators vpic_hip_ref_reduce_accumulators vpic_hip_ref_unload_accumulator
vpic_hip_ref_advance_p vpic_hip_ref_energy_p vpic_hip_ref_center_p vpic_hip_ref_uncenter_p vpic_hip_ref_sort_p vpic_hip_ref_advance_b vpic_hip_ref_advance_e
vpic_hip_ref_clear_jf vpic_hip_ref_synchronize_jf vpic_hip_ref_energy_f
vpic_hip_ref_clear_rhof vpic_hip_ref_accumulate_rhob vpic_hip_ref_accumulate_rho_p vpic_hip_ref_synchronize_rho vpic_hip_ref_compute_rhob
vpic_hip_ref_compute_curl_b vpic_hip_ref_synchronize_tang_e_norm_b vpic_hip_ref_compute_div_e_err
vpic_hip_ref_compute_rms_div_e_err vpic_hip_ref_clean_div_e vpic_hip_ref_compute_div_b_err
vpic_hip_ref_compute_rms_div_b_err vpic_hip_ref_clean_div_b
vpic_hip_ref_move_p vpic_hip_ref_boundary_p vpic_hip_ref_clear_hydro vpic_hip_ref_accumulate_hydro_p
vpic_hip_ref_synchronize_hydro vpic_hip_ref_local_adjust_hydro
vpic_hip_ref_new_field vpic_hip_ref_delete_field vpic_hip_ref_new_material_coefficients
vpic_hip_ref_delete_material_coefficients vpic_hip_ref_new_hydro vpic_hip_ref_delete_hydro
vpic_hip_ref_new_interpolator vpic_hip_ref_delete_interpolator vpic_hip_ref_new_accumulators
vpic_hip_ref_delete_accumulators""".split()
# field_advance_methods_t, slot by slot (src/field_advance/field_advance.h:185-302): the data symbol
# vpic_hip_ref_field_advance_methods holds these entry points in this order
FIELD_ADVANCE_SLOTS = """new_field delete_field new_material_coefficients delete_material_coefficients advance_b advance_e
energy_f clear_jf synchronize_jf clear_rhof synchronize_rho compute_rhob compute_curl_b synchronize_tang_e_norm_b
compute_div_e_err compute_rms_div_e_err clean_div_e compute_div_b_err compute_rms_div_b_err clean_div_b""".split()
