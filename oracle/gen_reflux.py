"""TEST INFRASTRUCTURE: golden vectors of the custom particle boundary handler maxwellian_reflux
(src/boundary/maxwellian_reflux.c:48-176) from the COMPILED REFERENCE (oracle/_ref/libvpic_ref.so), with the random
numbers the handler drew.  The handler is called directly (it is a public function, boundary.h) for particles parked
on each of the six faces; a second generator seeded alike is read through the reference's public mtrand API in the
handler's draw order (mt_frand, mt_frandn, mt_frandn) to learn the three numbers each call consumed.
-> tests/golden/reflux.npz.  Needs /root/reference (python oracle/gen_reflux.py)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref as R  # noqa: E402

L = importlib.import_module("old-vpic_amd.layout")
NX, NY, NZ = 6, 5, 4
LX, LY, LZ, DT = 6.0, 7.5, 3.0, 0.3            # unequal cell sizes: dx = 1, dy = 1.5, dz = 0.75


class Params(C.Structure):                    # boundary.h: maxwellian_reflux_t
    _fields_ = [("ut_perp", C.c_float * 32), ("ut_para", C.c_float * 32)]


def main():
    l = R.lib()
    g = R.new_periodic_grid(NX, NY, NZ, LX, LY, LZ, np.float32(DT))
    rng = np.random.default_rng(21)
    n_per_face = 40
    n = 6 * n_per_face
    p = np.zeros(n, L.particle_t)
    pm = np.zeros(n, L.particle_mover_t)
    face = np.repeat(np.arange(6), n_per_face).astype(np.int32)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    for c in ("ux", "uy", "uz"):
        p[c] = (0.8 * rng.standard_normal(n)).astype(np.float32)
    p["q"] = (-1 - np.arange(n) / 1024.0).astype(np.float32)          # a different charge each: identifies the particle
    x, y, z = rng.integers(1, NX + 1, n), rng.integers(1, NY + 1, n), rng.integers(1, NZ + 1, n)
    for k in range(n):                           # parked on its face, moving outwards, something of the step left
        f = int(face[k]); axis, hi = f % 3, f >= 3
        d, u = ("dx", "dy", "dz")[axis], ("ux", "uy", "uz")[axis]
        p[d][k] = 1.0 if hi else -1.0
        p[u][k] = abs(p[u][k]) + 0.05 if hi else -abs(p[u][k]) - 0.05
        if axis == 0: x[k] = NX if hi else 1
        if axis == 1: y[k] = NY if hi else 1
        if axis == 2: z[k] = NZ if hi else 1
    p["i"] = L.voxel(x, y, z, NX, NY, NZ)
    for c in ("dispx", "dispy", "dispz"):
        pm[c] = (0.3 * rng.uniform(-1, 1, n)).astype(np.float32)
    pm["i"] = np.arange(n)
    par = Params()
    ut_para, ut_perp = 0.11, 0.04
    par.ut_para[0], par.ut_perp[0] = ut_para, ut_perp
    l.new_mt_rng.restype = C.c_void_p
    l.mt_frand.restype = C.c_float
    l.mt_frandn.restype = C.c_float
    rng_a, rng_b = l.new_mt_rng(77), l.new_mt_rng(77)
    sp = l.ref_new_species(C.c_float(-1.0), 16, 16, 1)               # id 0
    inj = np.zeros(n, L.particle_injector_t)
    draws = np.zeros((n, 3), np.float32)
    fld = np.zeros(L.nv(NX, NY, NZ), L.field_t)
    acc = np.zeros(L.nv(NX, NY, NZ), L.accumulator_t)
    for k in range(n):
        ppi = C.c_void_p(inj.ctypes.data + k * inj.itemsize)
        l.maxwellian_reflux(C.byref(par), C.c_void_p(p.ctypes.data + k * p.itemsize), C.c_void_p(pm.ctypes.data + k * pm.itemsize),
                            R._p(fld), R._p(acc), R.V(g), R.V(sp), C.byref(ppi), R.V(rng_a), int(face[k]))
        assert ppi.value == inj.ctypes.data + (k + 1) * inj.itemsize
        draws[k] = [l.mt_frand(R.V(rng_b)), l.mt_frandn(R.V(rng_b)), l.mt_frandn(R.V(rng_b))]
    out = dict(dims=np.array([NX, NY, NZ]), box=np.array([LX, LY, LZ, DT]), p=p, pm=pm, face=face, draws=draws, inj=inj,
               ut=np.array([ut_para, ut_perp], np.float32))
    out.update(emitter(l, g))
    dst = os.path.join(ROOT, "tests", "golden", "reflux.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", n, "handler calls; first injector", inj[0])


class ChildLangmuir(C.Structure):               # emitter.h: child_langmuir_t
    _fields_ = [("n_emit_per_face", C.c_int), ("ut_perp", C.c_float), ("ut_para", C.c_float)]


def emitter(l, g):
    """The reference's child_langmuir (emitter.h) on a list of faces of all six orientations in a random interpolator;
    the six numbers each emitted particle consumed, read from a second generator in the model's draw order
    (mt_drand_c x 2, mt_drandn x 3, mt_drand_c0)."""
    rng = np.random.default_rng(31)
    nv = L.nv(NX, NY, NZ)
    fi = np.zeros(nv, L.interpolator_t)
    for c in ("ex", "ey", "ez"):
        fi[c] = rng.uniform(-2, 2, nv).astype(np.float32)
    types = {(-1, 0, 0): 12, (0, -1, 0): 10, (0, 0, -1): 4, (1, 0, 0): 14, (0, 1, 0): 16, (0, 0, 1): 22}
    comp = []
    for (dx_, dy_, dz_), t in types.items():
        for _ in range(12):
            x, y, z = rng.integers(2, NX), rng.integers(2, NY), rng.integers(2, NZ)      # interior cells: the first move stays local
            comp.append(int(L.voxel(x, y, z, NX, NY, NZ)) * 32 + t)
    comp.append(int(L.voxel(2, 2, 2, NX, NY, NZ)) * 32 + 13)                         # a cell body: not a face, emits nothing
    comp = np.array(comp, np.int32)
    n_emit, ut_perp, ut_para, q_m = 3, 0.05, 0.12, -1.0
    l.new_emitter.restype = C.c_void_p
    l.new_emitter.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    max_np = 4 * len(comp) * n_emit
    sp = l.ref_new_species(C.c_float(q_m), max_np, max_np, 1)
    e_list = C.c_void_p(None)
    em = l.new_emitter(b"cathode", R.V(sp), C.cast(l.child_langmuir, C.c_void_p), len(comp), C.byref(e_list))

    class Emitter(C.Structure):                 # emitter.h: emitter_t (head)
        _fields_ = [("component", C.POINTER(C.c_int)), ("n_component", C.c_int), ("max_component", C.c_int), ("sp", C.c_void_p),
                    ("emission_model", C.c_void_p), ("model_parameters", C.c_char * 1024)]
    E = Emitter.from_address(em)
    for k, c in enumerate(comp):
        E.component[k] = int(c)
    E.n_component = len(comp)
    C.memmove(C.addressof(E) + Emitter.model_parameters.offset, C.byref(ChildLangmuir(n_emit, ut_perp, ut_para)), C.sizeof(ChildLangmuir))
    l.new_mt_rng.restype = C.c_void_p
    for fn in ("mt_drand_c", "mt_drandn", "mt_drand_c0"):
        getattr(l, fn).restype = C.c_double
    rng_a, rng_b = l.new_mt_rng(91), l.new_mt_rng(91)
    f = np.zeros(nv, L.field_t)
    a = np.zeros(nv, L.accumulator_t)
    l.ref_species_set_counts(R.V(sp), 0, 0)
    l.child_langmuir(R.V(em), R._p(fi), R._p(f), R._p(a), R.V(g), R.V(rng_a))
    n_out, nm_out = l.ref_species_np(R.V(sp)), l.ref_species_nm(R.V(sp))
    p_out = np.frombuffer((C.c_char * (L.particle_t.itemsize * n_out)).from_address(l.ref_species_p(R.V(sp))), dtype=L.particle_t).copy()
    draws = np.zeros((n_out, 6))
    for k in range(n_out):
        draws[k] = [l.mt_drand_c(R.V(rng_b)), l.mt_drand_c(R.V(rng_b)), l.mt_drandn(R.V(rng_b)), l.mt_drandn(R.V(rng_b)),
                    l.mt_drandn(R.V(rng_b)), l.mt_drand_c0(R.V(rng_b))]
    assert nm_out == 0 and 0 < n_out < len(comp) * n_emit                        # some faces pull the species out, some do not
    print("emitter:", n_out, "particles from", len(comp), "components")
    return dict(emit_fi=fi, emit_component=comp, emit_par=np.array([n_emit, ut_perp, ut_para, q_m], np.float64), emit_draws=draws,
                emit_p=p_out, emit_rhob=f["rhob"].copy(), emit_a=a)


if __name__ == "__main__":
    main()
