"""TEST-ONLY stand-in with the Engine interface (old-vpic_amd/engine.py) that computes with the CPU
oracle.  It exists so that the multi-domain exchange choreography of old-vpic_amd/domain.py can be
exercised with the gloo backend on a machine without a GPU.  Never imported by the product."""
import ctypes as C
import importlib

import numpy as np

from oracle import pyorc

L = importlib.import_module("old-vpic_amd.layout")


def _view(ptr, dtype, n):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class OracleEngine:
    device_type = "cpu"

    def __init__(self, grid):
        g = pyorc.Grid()
        for name, _ in grid._fields_:
            v = getattr(grid, name)
            if name in ("fbc", "pbc"):
                for k in range(6):
                    getattr(g, name)[k] = v[k]
            else:
                setattr(g, name, v)
        self.g, self.nv = g, g.nv
        self.f = np.zeros(self.nv, L.field_t)
        self.fi = np.zeros(self.nv, L.interpolator_t)
        self.a = np.zeros(self.nv, L.accumulator_t)
        self.m = pyorc.vacuum_coefficients()
        self.sp = []
        self.send = [np.zeros(0, L.particle_injector_t) for _ in range(6)]

    def set_vacuum(self):
        pass

    def sync(self):
        pass

    def new_species(self, q_m, max_np, max_nm):
        self.sp.append(dict(q_m=q_m, p=np.zeros(max_np, L.particle_t), np=0, pm=np.zeros(max_nm, L.particle_mover_t),
                            nm=0, part=np.zeros(self.nv + 1, np.int32)))
        return len(self.sp) - 1

    def set_particles(self, sp, p):
        s = self.sp[sp]
        s["p"][:len(p)] = p
        s["np"], s["nm"] = len(p), 0

    def get_particles(self, sp):
        s = self.sp[sp]
        return s["p"][:s["np"]].copy()

    def np(self, sp):
        return self.sp[sp]["np"]

    def nm(self, sp):
        return self.sp[sp]["nm"]

    def set_fields(self, f):
        self.f[:] = f

    def get_fields(self):
        return self.f.copy()

    def load_interpolator(self):
        pyorc.load_interpolator(self.fi, self.f, self.g)

    def clear_accumulators(self):
        pyorc.clear_accumulators(self.a, self.g)

    def reduce_accumulators(self):
        pass

    def unload_accumulator(self):
        pyorc.unload_accumulator(self.f, self.a, self.g)

    def sort_due(self, sp, max_interval=0):       # no timing on the CPU stand-in: the upper bound decides
        s = self.sp[sp]
        s["since_sort"] = s.get("since_sort", 10 ** 9) + 1
        return max_interval > 0 and s["since_sort"] >= max_interval

    def sort_p(self, sp):
        self.sp[sp]["since_sort"] = 0
        s = self.sp[sp]
        pyorc.sort_p(s["p"], s["np"], s["part"], self.g)

    def advance_p(self, sp):
        s = self.sp[sp]
        s["nm"] = pyorc.advance_p(s["p"], s["np"], s["q_m"], s["pm"], self.a, self.fi, self.g)
        return s["nm"]

    def energy_p(self, sp):
        s = self.sp[sp]
        return pyorc.energy_p(s["p"], s["np"], s["q_m"], self.fi, self.g)

    def energy_f(self):
        return pyorc.energy_f(self.f, self.m, self.g)

    def clear_jf(self):
        pyorc.clear_jf(self.f, self.g)

    def local_adjust_jf(self):
        pyorc.local_adjust_jf(self.f, self.g)

    def synchronize_jf_self(self, axis):
        pyorc.synchronize_jf_self(self.f, self.g, axis)

    def advance_b(self, frac):
        pyorc.advance_b(self.f, self.g, frac)

    def advance_e(self):
        pyorc.advance_e(self.f, self.m, self.g)

    def face_count(self, d):
        return pyorc.tang_b_count(self.g, d)

    def pack_tang_b(self, d, ptr):
        _view(ptr, np.float32, self.face_count(d))[:] = pyorc.pack_tang_b(self.f, self.g, d)

    def unpack_tang_b(self, d, ptr):
        pyorc.unpack_tang_b(self.f, _view(ptr, np.float32, self.face_count(d)).copy(), self.g, d)

    def pack_jf(self, d, ptr):
        _view(ptr, np.float32, self.face_count(d))[:] = pyorc.pack_jf(self.f, self.g, d)

    def unpack_jf(self, d, ptr):
        pyorc.unpack_jf(self.f, _view(ptr, np.float32, self.face_count(d)).copy(), self.g, d)

    # ---- divergence cleaning family ------------------------------------------------------------
    def clear_rhof(self):
        pyorc.clear_rhof(self.f, self.g)

    def accumulate_rho_p(self, sp):
        s = self.sp[sp]
        pyorc.accumulate_rho_p(self.f, s["p"], s["np"], self.g)

    def local_adjust_rho(self):
        pyorc.local_adjust_rho(self.f, self.g)

    def synchronize_rho_self(self, axis):
        pyorc.synchronize_rho_self(self.f, self.g, axis)

    def rho_count(self, d):
        return pyorc.rho_count(self.g, d)

    def pack_rho(self, d, ptr):
        _view(ptr, np.float32, self.rho_count(d))[:] = pyorc.pack_rho(self.f, self.g, d)

    def unpack_rho(self, d, ptr):
        pyorc.unpack_rho(self.f, _view(ptr, np.float32, self.rho_count(d)).copy(), self.g, d)

    def message_count(self, kind, d):
        return pyorc.msg_count(self.g, kind, d)

    def pack_message(self, kind, d, ptr):
        _view(ptr, np.float32, self.message_count(kind, d))[:] = pyorc.pack_msg(self.f, self.g, kind, d)

    def unpack_message(self, kind, d, ptr):
        return pyorc.unpack_msg(self.f, _view(ptr, np.float32, self.message_count(kind, d)).copy(), self.g, kind, d)

    def local_adjust_tang_e_norm_b(self):
        pyorc.local_adjust_tang_e_norm_b(self.f, self.g)

    def synchronize_tang_e_norm_b_self(self, axis):
        return pyorc.synchronize_tang_e_norm_b_self(self.f, self.g, axis)

    def compute_rhob(self):
        pyorc.compute_rhob(self.f, self.m, self.g)

    def compute_curl_b(self):
        pyorc.compute_curl_b(self.f, self.m, self.g)

    def compute_div_e_err(self):
        pyorc.compute_div_e_err(self.f, self.m, self.g)

    def clean_div_e(self):
        pyorc.clean_div_e(self.f, self.m, self.g)

    def compute_div_b_err(self):
        pyorc.compute_div_b_err(self.f, self.g)

    def clean_div_b(self):
        pyorc.clean_div_b(self.f, self.g)

    def rms_div_e_err_local(self):
        return pyorc.rms_local(self.f, self.g, "e")

    def rms_div_b_err_local(self):
        return pyorc.rms_local(self.f, self.g, "b")

    def boundary_p_pack(self):
        outs = [[] for _ in range(6)]
        for k, s in enumerate(self.sp):
            if s["nm"]:
                s["np"], per_face = pyorc.boundary_p_pack(s["p"], s["np"], s["pm"], s["nm"], k, self.f, self.g, s["nm"])
                for f in range(6):
                    outs[f].append(per_face[f])
            s["nm"] = 0
        self.send = [np.concatenate(o) if o else np.zeros(0, L.particle_injector_t) for o in outs]
        return [len(x) for x in self.send]

    def get_injectors(self, face, ptr):
        n = len(self.send[face])
        _view(ptr, L.particle_injector_t, n)[:] = self.send[face]

    def boundary_p_inject(self, ptr, n):
        inj = _view(ptr, L.particle_injector_t, n).copy()
        for k, s in enumerate(self.sp):
            mine = inj[inj["sp_id"] == k]
            if len(mine):
                s["np"], s["nm"] = pyorc.boundary_p_inject(s["p"], s["np"], s["pm"], s["nm"], mine, self.a, self.g)


class ResidentOracleEngine(OracleEngine):
    """The same stand-in speaking the device-resident exchange protocol of the HIP engine (include/vpic_hip.h:
    vpic_hip_advance_p_phase, vpic_hip_exchange_*): fixed-capacity messages {int32 header[4]; injectors}, counts in the
    headers, one read-back.  Lets the CPU tests run SlabDomain.push_and_exchange -- the order in which species are pushed,
    packed, put on the wire and landed -- over gloo.  A stand-in: phase 1 pushes the whole species (what the HIP engine
    does for a species that is not in tile order), a full message is an error (the HIP engine parks the movers)."""

    def advance_p_async(self, sp):
        self.advance_p(sp)

    def advance_p_phase(self, sp, phase):
        if phase == 1:
            self.advance_p(sp)
        self.calls = getattr(self, "calls", []) + [("push", sp, phase)]

    def exchange_begin(self):
        self.exchange_flags = 0

    def capacity(self, sp):
        s = self.sp[sp]
        return s["np"], len(s["p"]), len(s["pm"])

    def reserve(self, sp, max_np, max_nm):
        s = self.sp[sp]
        if max_np > len(s["p"]):
            p = np.zeros(max_np, L.particle_t); p[:len(s["p"])] = s["p"]; s["p"] = p
        if max_nm > len(s["pm"]):
            pm = np.zeros(max_nm, L.particle_mover_t); pm[:len(s["pm"])] = s["pm"]; s["pm"] = pm

    def exchange_pack(self, ptrs, caps, mover_cap, species=None):
        which = range(len(self.sp)) if species is None else species
        outs = [[] for _ in range(6)]
        for k in which:
            s = self.sp[k]
            if s["nm"]:
                s["np"], per_face = pyorc.boundary_p_pack(s["p"], s["np"], s["pm"], s["nm"], k, self.f, self.g, s["nm"])
                for f in range(6):
                    outs[f].append(per_face[f])
            s["nm"] = 0
        self.calls = getattr(self, "calls", []) + [("pack", tuple(which))]
        for f in range(6):
            if not ptrs[f]:
                assert not any(len(o) for o in outs[f]), "a mover for a face without a message"
                continue
            inj = np.concatenate(outs[f]) if outs[f] else np.zeros(0, L.particle_injector_t)
            assert len(inj) <= caps[f], "message overflow (the stand-in cannot park movers)"
            hdr = _view(ptrs[f], np.int32, 4)
            hdr[:] = (len(inj), len(inj), 0, 0)
            _view(ptrs[f] + 16, L.particle_injector_t, caps[f])[:len(inj)] = inj

    def exchange_inject(self, ptr, cap):
        n = int(_view(ptr, np.int32, 4)[0])
        self.calls = getattr(self, "calls", []) + [("inject", n)]
        if n:
            self.boundary_p_inject(ptr + 16, n)

    def exchange_finish(self, ptrs):
        self.exchange_flags = 0
        return [list(_view(p, np.int32, 4)) for p in ptrs]
