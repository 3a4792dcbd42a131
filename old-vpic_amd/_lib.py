"""Loader of libvpic_hip.so -- the HIP engine's C ABI (include/vpic_hip.h).

There is no CPU fallback: if the library is missing it is built with hipcc; if that fails, or no
HIP device is present when an engine is created, the error is raised to the caller.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libvpic_hip.so")
CSRC = os.path.join(HERE, "csrc")
_lib = None


def build(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", CSRC, "-j4"])
    if not os.path.exists(SO):
        raise RuntimeError("building libvpic_hip.so failed")
    return SO


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    return any(os.path.getmtime(s) > t for s in srcs)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if _stale() and os.path.exists("/opt/rocm/bin/hipcc") and not os.environ.get("VPIC_HIP_NO_REBUILD"):
        build()
    try:
        # torch ships its own libamdhip64 (same SONAME): when both live in one process the HIP
        # runtime must be loaded once, so let torch load it first.
        import torch  # noqa: F401
    except Exception:
        pass
    # VPIC_HIP_LIB: load another build of the same ABI (A/B timing of kernel variants on one GPU box)
    L = C.CDLL(os.environ.get("VPIC_HIP_LIB", SO))
    L.vpic_hip_last_error.restype = C.c_char_p
    L.vpic_hip_stream.restype = C.c_void_p
    L.vpic_hip_boundary_p_send_buffer.restype = C.c_void_p
    L.vpic_hip_species_np.restype = C.c_int64
    L.vpic_hip_species_nm.restype = C.c_int64
    L.vpic_hip_species_create.argtypes = [C.c_void_p, C.c_float, C.c_int64, C.c_int64]
    L.vpic_hip_species_set_particles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.vpic_hip_species_get_particles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.vpic_hip_species_get_particles_range.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64]
    if hasattr(L, "vpic_hip_species_reserve"):
        L.vpic_hip_species_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64]
        L.vpic_hip_species_capacity.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vpic_hip_exchange_pack_species.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    if hasattr(L, "vpic_hip_set_accumulation"):
        L.vpic_hip_set_accumulation.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.vpic_hip_push_plan.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.vpic_hip_set_reflux_draws.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.vpic_hip_set_emit_draws.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.vpic_hip_species_append_particles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.vpic_hip_emit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32]
    L.vpic_hip_inject_aged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.vpic_hip_accumulate_rhob.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
    L.vpic_hip_set_maxwellian_reflux.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32]
    L.vpic_hip_species_get_movers.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.vpic_hip_species_np.argtypes = [C.c_void_p, C.c_int]
    L.vpic_hip_species_load_maxwellian.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32] + [C.c_float] * 5
    L.vpic_hip_species_nm.argtypes = [C.c_void_p, C.c_int]
    L.vpic_hip_advance_b.argtypes = [C.c_void_p, C.c_float]
    L.vpic_hip_step.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.vpic_hip_boundary_p_send_buffer.argtypes = [C.c_void_p, C.c_int]
    L.vpic_hip_boundary_p_inject.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.vpic_hip_boundary_p_get_injectors.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.vpic_hip_synchronize_jf_self.argtypes = [C.c_void_p, C.c_int]
    L.vpic_hip_exchange_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.vpic_hip_exchange_inject.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.vpic_hip_exchange_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.vpic_hip_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
    L.vpic_hip_advance_e_part.argtypes = [C.c_void_p, C.c_int]
    for n in ("vpic_hip_pack_tang_b", "vpic_hip_unpack_tang_b", "vpic_hip_pack_jf", "vpic_hip_unpack_jf"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.vpic_hip_face_count.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "vpic_hip_comm_create"):
        L.vpic_hip_comm_unique_id.argtypes = [C.c_void_p]
        L.vpic_hip_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.vpic_hip_comm_destroy.argtypes = [C.c_void_p]
        L.vpic_hip_comm_start.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vpic_hip_comm_finish.argtypes = [C.c_void_p, C.c_int]
        L.vpic_hip_comm_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vpic_hip_comm_timing.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vpic_hip_dump_gather.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_size_t]
    _lib = L
    return L


EXPORTS = """vpic_hip_clear_jf_unload_accumulator vpic_hip_set_emit_draws vpic_hip_set_reflux_draws vpic_hip_set_host_access_hook vpic_hip_last_error vpic_hip_device_count vpic_hip_create vpic_hip_destroy vpic_hip_sync
vpic_hip_stream vpic_hip_nv vpic_hip_set_fields vpic_hip_get_fields vpic_hip_set_interpolator
vpic_hip_get_interpolator vpic_hip_set_accumulator vpic_hip_get_accumulator
vpic_hip_set_material_coefficients vpic_hip_species_create vpic_hip_species_set_particles vpic_hip_species_append_particles vpic_hip_set_maxwellian_reflux vpic_hip_accumulate_rhob vpic_hip_inject_aged vpic_hip_emit
vpic_hip_species_get_particles vpic_hip_species_load_maxwellian vpic_hip_species_np vpic_hip_species_nm vpic_hip_species_get_movers
vpic_hip_species_get_partition vpic_hip_load_interpolator vpic_hip_clear_accumulators
vpic_hip_reduce_accumulators vpic_hip_unload_accumulator vpic_hip_advance_p vpic_hip_sort_p
vpic_hip_set_push_mode vpic_hip_species_get_particles_range vpic_hip_push_plan vpic_hip_species_sort_hint vpic_hip_set_accumulation vpic_hip_advance_p_async vpic_hip_advance_p_phase vpic_hip_exchange_pack_species vpic_hip_species_capacity vpic_hip_species_reserve vpic_hip_exchange_begin vpic_hip_exchange_pack vpic_hip_exchange_inject vpic_hip_exchange_finish vpic_hip_advance_e_part vpic_hip_stream_wait_event vpic_hip_energy_p vpic_hip_center_p vpic_hip_uncenter_p vpic_hip_clear_jf vpic_hip_sort_due vpic_hip_species_sort_order vpic_hip_set_sort_order vpic_hip_measure_disorder vpic_hip_species_set_movers vpic_hip_device_alloc vpic_hip_device_free vpic_hip_copy_to_host vpic_hip_copy_from_host vpic_hip_clear_hydro vpic_hip_accumulate_hydro_p vpic_hip_synchronize_hydro vpic_hip_local_adjust_hydro vpic_hip_synchronize_hydro_self vpic_hip_hydro_count vpic_hip_pack_hydro vpic_hip_unpack_hydro vpic_hip_set_hydro vpic_hip_get_hydro vpic_hip_dump_gather vpic_hip_clear_rhof vpic_hip_accumulate_rho_p vpic_hip_synchronize_rho vpic_hip_local_adjust_rho vpic_hip_synchronize_rho_self vpic_hip_rho_count vpic_hip_pack_rho vpic_hip_unpack_rho vpic_hip_compute_rhob vpic_hip_compute_curl_b vpic_hip_synchronize_tang_e_norm_b vpic_hip_face_message_count vpic_hip_pack_face_message vpic_hip_unpack_face_message vpic_hip_local_adjust_tang_e_norm_b vpic_hip_synchronize_tang_e_norm_b_self vpic_hip_compute_div_e_err vpic_hip_clean_div_e vpic_hip_compute_div_b_err vpic_hip_clean_div_b vpic_hip_rms_div_e_err_local vpic_hip_rms_div_b_err_local vpic_hip_compute_rms_div_e_err vpic_hip_compute_rms_div_b_err vpic_hip_synchronize_jf vpic_hip_advance_b vpic_hip_advance_e
vpic_hip_energy_f vpic_hip_boundary_p_pack vpic_hip_boundary_p_counts vpic_hip_boundary_p_send_buffer
vpic_hip_boundary_p_inject vpic_hip_boundary_p_get_injectors vpic_hip_local_adjust_jf
vpic_hip_synchronize_jf_self vpic_hip_face_count vpic_hip_pack_tang_b vpic_hip_unpack_tang_b
vpic_hip_pack_jf vpic_hip_unpack_jf vpic_hip_step vpic_hip_profile_enable vpic_hip_profile_read vpic_hip_profile_read_sorting vpic_hip_sort_advance_p vpic_hip_species_get_tile_partition vpic_hip_profile_read_species vpic_hip_comm_unique_id vpic_hip_comm_create vpic_hip_comm_destroy vpic_hip_comm_start vpic_hip_comm_finish vpic_hip_comm_stats vpic_hip_comm_timing vpic_hip_species_stats""".split()
