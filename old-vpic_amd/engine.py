"""Host-side mirror of the reference's kernel-level interface over the resident HIP engine.

Method names and argument meaning follow the reference's L3 C functions
(src/species_advance/standard/spa.h:23-123, src/sf_interface/sf_interface.h:83-163,
field_advance_methods_t in src/field_advance/field_advance.h:185-302); arrays cross this boundary
in the reference's own array-of-struct layouts (layout.py).  Everything computes on the GPU
through libvpic_hip.so; errors raise VpicHipError (the reference would print and exit(1),
src/util/util_base.h:213-219).
"""
import ctypes as C

import numpy as np

from . import layout as L
from ._lib import lib


class VpicHipError(RuntimeError):
    pass


class GridDesc(C.Structure):
    """vpic_hip_grid_t (include/vpic_hip.h) -- what the kernels read from grid_t (src/grid/grid.h:112-167)."""
    _fields_ = [("dt", C.c_float), ("cvac", C.c_float), ("eps0", C.c_float), ("damp", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float),
                ("rdx", C.c_float), ("rdy", C.c_float), ("rdz", C.c_float),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("fbc", C.c_int32 * 6), ("pbc", C.c_int32 * 6), ("rank", C.c_int32)]

    @property
    def nv(self):
        return L.nv(self.nx, self.ny, self.nz)


def make_grid(nx, ny, nz, lx, ly, lz, dt, cvac=1.0, eps0=1.0, damp=0.0, fbc=None, pbc=None, rank=0):
    """A box domain.  Cell sizes are formed as partition_periodic_box does
    (src/grid/partition.c:60-66): double arithmetic, stored as float."""
    g = GridDesc()
    g.dt, g.cvac, g.eps0, g.damp = dt, cvac, eps0, damp
    g.dx, g.dy, g.dz = lx / nx, ly / ny, lz / nz
    g.rdx, g.rdy, g.rdz = nx / lx, ny / ly, nz / lz
    g.nx, g.ny, g.nz = nx, ny, nz
    g.rank = rank
    for f in range(6):
        g.fbc[f] = rank if fbc is None else fbc[f]
        g.pbc[f] = rank if pbc is None else pbc[f]
    return g


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One rectangular domain resident on one GPU."""

    device_type = "cuda"   # where exchange buffers handed to pack_*/unpack_*/inject must live

    def __init__(self, grid, device=-1):
        self._l = lib()
        self.grid = grid
        h = C.c_void_p()
        if self._l.vpic_hip_create(C.byref(h), C.byref(grid), device):
            raise VpicHipError(self._l.vpic_hip_last_error().decode())
        self._h = h
        self.nv = grid.nv

    def close(self):
        if getattr(self, "_h", None):
            self._l.vpic_hip_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        if rc:
            raise VpicHipError(self._l.vpic_hip_last_error().decode())

    def _arr(self, a, dtype, n=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        if n is not None and len(a) < n:
            raise VpicHipError(f"array of {len(a)} entries, need {n}")
        return a

    # ---- host mirrors ----
    def set_fields(self, f):
        f = self._arr(f, L.field_t, self.nv)
        self._ck(self._l.vpic_hip_set_fields(self._h, _ptr(f)))

    def get_fields(self):
        f = np.zeros(self.nv, L.field_t)
        self._ck(self._l.vpic_hip_get_fields(self._h, _ptr(f)))
        return f

    def set_interpolator(self, fi):
        fi = self._arr(fi, L.interpolator_t, self.nv)
        self._ck(self._l.vpic_hip_set_interpolator(self._h, _ptr(fi)))

    def get_interpolator(self):
        fi = np.zeros(self.nv, L.interpolator_t)
        self._ck(self._l.vpic_hip_get_interpolator(self._h, _ptr(fi)))
        return fi

    def set_accumulator(self, a):
        a = self._arr(a, L.accumulator_t, self.nv)
        self._ck(self._l.vpic_hip_set_accumulator(self._h, _ptr(a)))

    def get_accumulator(self):
        a = np.zeros(self.nv, L.accumulator_t)
        self._ck(self._l.vpic_hip_get_accumulator(self._h, _ptr(a)))
        return a

    def set_material_coefficients(self, m):
        m = self._arr(m, L.material_coefficient_t)
        self._ck(self._l.vpic_hip_set_material_coefficients(self._h, _ptr(m), len(m)))

    def set_vacuum(self):
        """One vacuum material (src/field_advance/standard/sfa.c:145-177 with eps=mu=1, sigma=0)."""
        m = np.zeros(1, L.material_coefficient_t)
        for n in ("decayx", "decayy", "decayz", "drivex", "drivey", "drivez", "rmux", "rmuy", "rmuz",
                  "nonconductive", "epsx", "epsy", "epsz"):
            m[n] = 1.0
        self.set_material_coefficients(m)

    # ---- species ----
    def new_species(self, q_m, max_np, max_nm):
        sp = self._l.vpic_hip_species_create(self._h, q_m, int(max_np), int(max_nm))
        if sp < 0:
            raise VpicHipError(self._l.vpic_hip_last_error().decode())
        return sp

    def set_particles(self, sp, p):
        p = self._arr(p, L.particle_t)
        self._ck(self._l.vpic_hip_species_set_particles(self._h, sp, _ptr(p), len(p)))

    def emit(self, sp, components, n_emit_per_face, ut_perp, ut_para, coef=32.0 / 81.0, thresh=0.0, seed=1):
        c = np.ascontiguousarray(components, np.int32)
        self._ck(self._l.vpic_hip_emit(self._h, sp, c.ctypes.data_as(C.c_void_p), len(c), n_emit_per_face, ut_perp, ut_para, coef, thresh, seed))

    def accumulate_rhob(self, p, q_scale=1.0):
        p = self._arr(p, L.particle_t)
        self._ck(self._l.vpic_hip_accumulate_rhob(self._h, _ptr(p), len(p), q_scale))

    def set_maxwellian_reflux(self, code, ut_para, ut_perp, seed=1):
        a, b = np.ascontiguousarray(ut_para, np.float32), np.ascontiguousarray(ut_perp, np.float32)
        self._ck(self._l.vpic_hip_set_maxwellian_reflux(self._h, int(code), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), len(a), int(seed)))

    def set_reflux_draws(self, draws):
        """Test mode: draws[k] = (uniform, normal, normal) for the particle at index k of its species' array."""
        d = np.ascontiguousarray(draws, np.float32).reshape(-1, 3)
        self._ck(self._l.vpic_hip_set_reflux_draws(self._h, d.ctypes.data_as(C.c_void_p), len(d)))

    def set_emit_draws(self, draws):
        """Test mode of emit(): draws[slot] = six numbers, slot = component index * n_emit_per_face + k."""
        d = np.ascontiguousarray(draws, np.float64).reshape(-1, 6)
        self._ck(self._l.vpic_hip_set_emit_draws(self._h, d.ctypes.data_as(C.c_void_p), len(d)))

    def set_movers(self, sp, pm):
        pm = self._arr(pm, L.particle_mover_t)
        self._ck(self._l.vpic_hip_species_set_movers(self._h, sp, _ptr(pm), len(pm)))

    def append_particles(self, sp, p):
        p = self._arr(p, L.particle_t)
        self._ck(self._l.vpic_hip_species_append_particles(self._h, sp, _ptr(p), len(p)))

    def get_particles(self, sp):
        n = self.np(sp)
        p = np.zeros(n, L.particle_t)
        self._ck(self._l.vpic_hip_species_get_particles(self._h, sp, _ptr(p), n))
        return p

    def get_particles_range(self, sp, first, count):
        p = np.zeros(count, L.particle_t)
        self._ck(self._l.vpic_hip_species_get_particles_range(self._h, sp, _ptr(p), int(first), int(count)))
        return p

    def load_maxwellian(self, sp, ppc, seed, q, u=(0.0, 0.0, 0.0), vth=0.0):
        """ppc synthetic particles in every interior cell (device-side loader)."""
        self._ck(self._l.vpic_hip_species_load_maxwellian(self._h, sp, int(ppc), int(seed), q, u[0], u[1], u[2], vth))

    def np(self, sp):
        return int(self._l.vpic_hip_species_np(self._h, sp))

    def nm(self, sp):
        return int(self._l.vpic_hip_species_nm(self._h, sp))

    def get_movers(self, sp):
        n = self.nm(sp)
        pm = np.zeros(n, L.particle_mover_t)
        self._ck(self._l.vpic_hip_species_get_movers(self._h, sp, _ptr(pm), n))
        return pm

    def get_partition(self, sp):
        part = np.zeros(self.nv + 1, np.int32)
        self._ck(self._l.vpic_hip_species_get_partition(self._h, sp, _ptr(part)))
        return part

    def get_tile_partition(self, sp):
        """tpart[64 * tiles + 1] of the engine's own order (include/vpic_hip.h, vpic_hip_species_get_tile_partition)."""
        n = C.c_int64()
        self._ck(self._l.vpic_hip_species_get_tile_partition(self._h, sp, None, C.byref(n)))
        t = np.zeros(n.value, np.int32)
        self._ck(self._l.vpic_hip_species_get_tile_partition(self._h, sp, _ptr(t), C.byref(n)))
        return t

    # ---- kernels (reference names) ----
    def load_interpolator(self):
        self._ck(self._l.vpic_hip_load_interpolator(self._h))

    def clear_accumulators(self):
        self._ck(self._l.vpic_hip_clear_accumulators(self._h))

    def reduce_accumulators(self):
        self._ck(self._l.vpic_hip_reduce_accumulators(self._h))

    def unload_accumulator(self):
        self._ck(self._l.vpic_hip_unload_accumulator(self._h))

    def set_push_mode(self, mode):
        """'exact' (default: the reference's scalar arithmetic, bit for bit) or 'fast' (include/vpic_hip.h)."""
        if mode == "exact" and not hasattr(self._l, "vpic_hip_set_push_mode"):
            return                                          # an older build of the ABI (VPIC_HIP_LIB, A/B timing): exact is all it has
        self._ck(self._l.vpic_hip_set_push_mode(self._h, {"exact": 0, "fast": 1}[mode]))

    def set_accumulation(self, mode, q_ref=0.0):
        """'float' (default) or 'deterministic' (64-bit fixed-point sums: bit-identical from run to run): include/vpic_hip.h."""
        self._ck(self._l.vpic_hip_set_accumulation(self._h, {"float": 0, "deterministic": 1}[mode], float(q_ref)))

    def advance_p(self, sp):
        """Returns the number of movers, like the reference's advance_p."""
        self._ck(self._l.vpic_hip_advance_p(self._h, sp))
        return self.nm(sp)

    # ---- the exchange that keeps its counts on the device (include/vpic_hip.h: vpic_hip_exchange_*) ----
    def advance_p_async(self, sp):
        self._ck(self._l.vpic_hip_advance_p_async(self._h, sp))

    def advance_p_phase(self, sp, phase):
        """phase 1: the tiles on the faces shared with other domains and the appended particles; 2: the other tiles."""
        self._ck(self._l.vpic_hip_advance_p_phase(self._h, sp, int(phase)))

    def capacity(self, sp):
        """(extent, max_np, max_nm): array slots in use (live particles + dead slots) and the capacities."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self._l.vpic_hip_species_capacity(self._h, sp, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def reserve(self, sp, max_np, max_nm):
        self._ck(self._l.vpic_hip_species_reserve(self._h, sp, int(max_np), int(max_nm)))

    def exchange_begin(self):
        self._ck(self._l.vpic_hip_exchange_begin(self._h))

    @staticmethod
    def exchange_message_bytes(cap):
        return 16 + 48 * int(cap)

    def exchange_pack(self, msg_ptrs, caps, mover_cap, species=None):
        """msg_ptrs / caps: per face 0..5, a device pointer (or 0) and the payload capacity of its message;
        species: the species whose movers are packed (default: all)."""
        m = (C.c_void_p * 6)(*[C.c_void_p(p or None) for p in msg_ptrs])
        c = (C.c_int32 * 6)(*[int(x) for x in caps])
        if species is None:
            self._ck(self._l.vpic_hip_exchange_pack(self._h, m, c, int(mover_cap)))
        else:
            mask = sum(1 << int(k) for k in species)
            self._ck(self._l.vpic_hip_exchange_pack_species(self._h, mask, m, c, int(mover_cap)))

    def exchange_inject(self, msg_ptr, cap):
        self._ck(self._l.vpic_hip_exchange_inject(self._h, C.c_void_p(msg_ptr), int(cap)))

    def exchange_finish(self, recv_ptrs):
        """The one synchronisation of the particle exchange.  Returns the headers of the received messages
        ([count, wanted, 0, 0] each); np / nm of every species are current on the host afterwards."""
        n = len(recv_ptrs)
        r = (C.c_void_p * max(n, 1))(*[C.c_void_p(p) for p in recv_ptrs])
        h = (C.c_int32 * (4 * max(n, 1)))()
        flags = C.c_int32()
        self._ck(self._l.vpic_hip_exchange_finish(self._h, r, n, h, C.byref(flags)))
        self.exchange_flags = flags.value
        return [list(h[4 * k:4 * k + 4]) for k in range(n)]

    def advance_e_part(self, part):
        self._ck(self._l.vpic_hip_advance_e_part(self._h, int(part)))

    def stream_wait_event(self, hip_event):
        self._ck(self._l.vpic_hip_stream_wait_event(self._h, C.c_void_p(hip_event)))

    def sort_p(self, sp):
        self._ck(self._l.vpic_hip_sort_p(self._h, sp))

    def sort_advance_p(self, sp):
        """sort_p then advance_p as one call (the push does the sort's moving when it can: vpic_hip_sort_advance_p)."""
        self._ck(self._l.vpic_hip_sort_advance_p(self._h, sp))
        return self.nm(sp)

    def energy_p(self, sp):
        e = C.c_double()
        self._ck(self._l.vpic_hip_energy_p(self._h, sp, C.byref(e)))
        return e.value

    def center_p(self, sp):
        self._ck(self._l.vpic_hip_center_p(self._h, sp))

    def uncenter_p(self, sp):
        self._ck(self._l.vpic_hip_uncenter_p(self._h, sp))

    # ---- hydro moments (sf_interface.h:83-163, spa.h:115-123) --------------------------------------
    def clear_hydro(self):
        self._ck(self._l.vpic_hip_clear_hydro(self._h))

    def accumulate_hydro_p(self, sp):
        self._ck(self._l.vpic_hip_accumulate_hydro_p(self._h, sp))

    def synchronize_hydro(self):
        self._ck(self._l.vpic_hip_synchronize_hydro(self._h))

    def set_hydro(self, h):
        h = np.ascontiguousarray(h, L.hydro_t)
        assert len(h) == self.nv
        self._ck(self._l.vpic_hip_set_hydro(self._h, h.ctypes.data_as(C.c_void_p)))

    def get_hydro(self):
        h = np.zeros(self.nv, L.hydro_t)
        self._ck(self._l.vpic_hip_get_hydro(self._h, h.ctypes.data_as(C.c_void_p)))
        return h

    def dump_gather(self, what, layout, words=(), strides=(1, 1, 1)):
        """Payload of field_dump (what 0) / hydro_dump (what 1) as uint32 words; layout 0 band
        [len(words), nz/sz+2, ny/sy+2, nx/sx+2], 1 interleaved records with the boundary entries,
        2 hydro_dump's interleaved shape (dump.cxx:1116-1552)."""
        W = 20 if what == 0 else 16
        no = [n // s for n, s in zip((self.grid.nx, self.grid.ny, self.grid.nz), strides)]
        dim = [n + (0 if layout == 2 else 2) for n in no]
        shape = (len(words), dim[2], dim[1], dim[0]) if layout == 0 else (dim[2], dim[1], dim[0], W)
        out = np.zeros(shape, np.uint32)
        w = np.asarray(words, np.int32)
        self._ck(self._l.vpic_hip_dump_gather(self._h, what, layout, w.ctypes.data_as(C.c_void_p), len(w), int(strides[0]),
                                              int(strides[1]), int(strides[2]), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out

    # ---- divergence cleaning family and charge densities (field_advance.h:242-302, spa.h:108-113) ----
    def clear_rhof(self):
        self._ck(self._l.vpic_hip_clear_rhof(self._h))

    def accumulate_rho_p(self, sp):
        self._ck(self._l.vpic_hip_accumulate_rho_p(self._h, sp))

    def synchronize_rho(self):
        self._ck(self._l.vpic_hip_synchronize_rho(self._h))

    def compute_rhob(self):
        self._ck(self._l.vpic_hip_compute_rhob(self._h))

    def compute_curl_b(self):
        self._ck(self._l.vpic_hip_compute_curl_b(self._h))

    def synchronize_tang_e_norm_b(self):
        err = C.c_double()
        self._ck(self._l.vpic_hip_synchronize_tang_e_norm_b(self._h, C.byref(err)))
        return err.value

    def compute_div_e_err(self):
        self._ck(self._l.vpic_hip_compute_div_e_err(self._h))

    def compute_rms_div_e_err(self):
        r = C.c_double()
        self._ck(self._l.vpic_hip_compute_rms_div_e_err(self._h, C.byref(r)))
        return r.value

    def clean_div_e(self):
        self._ck(self._l.vpic_hip_clean_div_e(self._h))

    def compute_div_b_err(self):
        self._ck(self._l.vpic_hip_compute_div_b_err(self._h))

    def compute_rms_div_b_err(self):
        r = C.c_double()
        self._ck(self._l.vpic_hip_compute_rms_div_b_err(self._h, C.byref(r)))
        return r.value

    def clean_div_b(self):
        self._ck(self._l.vpic_hip_clean_div_b(self._h))

    def clear_jf(self):
        self._ck(self._l.vpic_hip_clear_jf(self._h))

    def clear_jf_unload_accumulator(self):
        self._ck(self._l.vpic_hip_clear_jf_unload_accumulator(self._h))

    def synchronize_jf(self):
        self._ck(self._l.vpic_hip_synchronize_jf(self._h))

    def local_adjust_jf(self):
        self._ck(self._l.vpic_hip_local_adjust_jf(self._h))

    def synchronize_jf_self(self, axis):
        self._ck(self._l.vpic_hip_synchronize_jf_self(self._h, axis))

    def advance_b(self, frac):
        self._ck(self._l.vpic_hip_advance_b(self._h, frac))

    def advance_e(self):
        self._ck(self._l.vpic_hip_advance_e(self._h))

    def energy_f(self):
        en = np.zeros(6, np.float64)
        self._ck(self._l.vpic_hip_energy_f(self._h, _ptr(en)))
        return en

    def boundary_p_pack(self):
        self._ck(self._l.vpic_hip_boundary_p_pack(self._h))
        ns = (C.c_int32 * 6)()
        self._ck(self._l.vpic_hip_boundary_p_counts(self._h, ns))
        return list(ns)

    def send_buffer(self, face):
        return self._l.vpic_hip_boundary_p_send_buffer(self._h, face)

    def get_injectors(self, face, dev_ptr):
        self._ck(self._l.vpic_hip_boundary_p_get_injectors(self._h, face, C.c_void_p(dev_ptr)))

    def boundary_p_inject(self, dev_ptr, n):
        self._ck(self._l.vpic_hip_boundary_p_inject(self._h, C.c_void_p(dev_ptr), int(n)))

    def face_count(self, d):
        return self._l.vpic_hip_face_count(self._h, d)

    def pack_tang_b(self, d, dev_ptr):
        self._ck(self._l.vpic_hip_pack_tang_b(self._h, d, C.c_void_p(dev_ptr)))

    def unpack_tang_b(self, d, dev_ptr):
        self._ck(self._l.vpic_hip_unpack_tang_b(self._h, d, C.c_void_p(dev_ptr)))

    def pack_jf(self, d, dev_ptr):
        self._ck(self._l.vpic_hip_pack_jf(self._h, d, C.c_void_p(dev_ptr)))

    def unpack_jf(self, d, dev_ptr):
        self._ck(self._l.vpic_hip_unpack_jf(self._h, d, C.c_void_p(dev_ptr)))

    # pieces of the divergence-cleaning family for domains that share faces with other domains
    def local_adjust_rho(self):
        self._ck(self._l.vpic_hip_local_adjust_rho(self._h))

    def synchronize_rho_self(self, axis):
        self._ck(self._l.vpic_hip_synchronize_rho_self(self._h, axis))

    def rho_count(self, d):
        return self._l.vpic_hip_rho_count(self._h, d)

    def pack_rho(self, d, dev_ptr):
        self._ck(self._l.vpic_hip_pack_rho(self._h, d, C.c_void_p(dev_ptr)))

    def unpack_rho(self, d, dev_ptr):
        self._ck(self._l.vpic_hip_unpack_rho(self._h, d, C.c_void_p(dev_ptr)))

    def message_count(self, kind, d):
        return self._l.vpic_hip_face_message_count(self._h, kind, d)

    def pack_message(self, kind, d, dev_ptr):
        self._ck(self._l.vpic_hip_pack_face_message(self._h, kind, d, C.c_void_p(dev_ptr)))

    def unpack_message(self, kind, d, dev_ptr):
        """Returns the squared-difference sum of a kind-2 message, 0.0 otherwise."""
        err = C.c_double()
        self._ck(self._l.vpic_hip_unpack_face_message(self._h, kind, d, C.c_void_p(dev_ptr), C.byref(err)))
        return err.value

    def local_adjust_tang_e_norm_b(self):
        self._ck(self._l.vpic_hip_local_adjust_tang_e_norm_b(self._h))

    def synchronize_tang_e_norm_b_self(self, axis):
        err = C.c_double()
        self._ck(self._l.vpic_hip_synchronize_tang_e_norm_b_self(self._h, axis, C.byref(err)))
        return err.value

    def rms_div_e_err_local(self):
        l2 = (C.c_double * 2)()
        self._ck(self._l.vpic_hip_rms_div_e_err_local(self._h, l2))
        return l2[0], l2[1]

    def rms_div_b_err_local(self):
        l2 = (C.c_double * 2)()
        self._ck(self._l.vpic_hip_rms_div_b_err_local(self._h, l2))
        return l2[0], l2[1]

    def sort_due(self, sp, max_interval=0):
        """Adaptive sorting: should species sp be sorted before its next advance_p?"""
        d = C.c_int()
        self._ck(self._l.vpic_hip_sort_due(self._h, sp, int(max_interval), C.byref(d)))
        return bool(d.value)

    def set_sort_order(self, order):
        """'reference' (by voxel, the default) or 'engine' (tile order where it applies): include/vpic_hip.h."""
        self._ck(self._l.vpic_hip_set_sort_order(self._h, {"reference": 0, "engine": 1}[order]))

    def species_order(self, sp):
        """'none', 'voxel' (partition valid) or 'tile' (include/vpic_hip.h, vpic_hip_species_sort_order)."""
        o = C.c_int(0)
        self._ck(self._l.vpic_hip_species_sort_order(self._h, sp, C.byref(o)))
        return ("none", "voxel", "tile")[o.value]

    def species_stats(self, sp):
        """dict of what the engine knows about the species' last push and its sorts (include/vpic_hip.h, vpic_hip_species_stats)"""
        out = (C.c_int64 * 8)()
        self._ck(self._l.vpic_hip_species_stats(self._h, int(sp), out))
        return dict(zip(("crossed", "fullest_tile", "missed_runs", "sorts", "early_sorts", "too_clumped_for_tiles", "by_tile_only", "dead_slots"), [int(v) for v in out]))

    def measure_disorder(self, sp):
        f = C.c_double()
        self._ck(self._l.vpic_hip_measure_disorder(self._h, sp, C.byref(f)))
        return f.value

    def step(self, step, sort_interval=0):
        self._ck(self._l.vpic_hip_step(self._h, int(step), int(sort_interval)))

    def sync(self):
        self._ck(self._l.vpic_hip_sync(self._h))

    def stream(self):
        return self._l.vpic_hip_stream(self._h)

    def profile_enable(self, on=True):
        self._ck(self._l.vpic_hip_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, n, parts = C.c_double(), C.c_int64(), C.c_int64()
        self._ck(self._l.vpic_hip_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(parts)))
        return ms.value, n.value, parts.value

    def profile_read_species(self, sp):
        """the plain advance_p launches of one species: (ms, launches, particles)"""
        ms, n, parts = C.c_double(), C.c_int64(), C.c_int64()
        self._ck(self._l.vpic_hip_profile_read_species(self._h, int(sp), C.byref(ms), C.byref(n), C.byref(parts)))
        return ms.value, n.value, parts.value

    def profile_read_sorting(self):
        """the launches of advance_p that sorted the species as they pushed it (vpic_hip_step, fixed interval): booked apart"""
        ms, n, parts = C.c_double(), C.c_int64(), C.c_int64()
        self._ck(self._l.vpic_hip_profile_read_sorting(self._h, C.byref(ms), C.byref(n), C.byref(parts)))
        return ms.value, n.value, parts.value
