"""ctypes binding of libmundy_hip.so (include/mundy_hip.h).  Plumbing only: torch supplies device memory and the
current HIP stream; every computation happens in the hand-written HIP library.  There is no CPU fallback: if the
library is missing `load()` raises, and on a box without a GPU every compute entry point fails in HIP.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmundy_hip.so")

SUCCESS, ERR_INVALID_ARGUMENT, ERR_LOGIC, ERR_RUNTIME, ERR_HIP, ERR_NO_DEVICE = range(6)
SEARCH_SPHERES, SEARCH_AABB = 0, 1
SEARCH_METHOD_AUTO, SEARCH_METHOD_GRID, SEARCH_METHOD_MORTON_LBVH = 0, 1, 2
SPACE_UNCONSTRAINED, SPACE_LOWER_BOUND, SPACE_UPPER_BOUND, SPACE_BOUNDED = 0, 1, 2, 3
RESIDUAL_PROJECTED_DIFF, RESIDUAL_PROJECTED_GRADIENT = 0, 1


class MhipError(RuntimeError):
    """HIP runtime failure or missing device (MHIP_ERR_HIP / MHIP_ERR_NO_DEVICE)."""


class BroadphaseConfig(C.Structure):
    _fields_ = [("search_kind", C.c_int), ("symmetric", C.c_int), ("buffer", C.c_double), ("periodic", C.c_int),
                ("box", C.c_double * 3), ("method", C.c_int), ("include_self", C.c_int), ("cell", C.c_double * 9)]


class Space(C.Structure):
    _fields_ = [("kind", C.c_int), ("lower_bound", C.c_double), ("upper_bound", C.c_double)]


class PgdConfig(C.Structure):
    _fields_ = [("max_iters", C.c_uint), ("tol", C.c_double), ("residual_kind", C.c_int)]


class SolveResult(C.Structure):
    _fields_ = [("num_iters", C.c_uint), ("residual", C.c_double), ("converged", C.c_int)]


class VelocityHalo(C.Structure):
    """mhip_velocity_halo: the per-iteration ghost-velocity exchange of one rank (host lists + one device index list)"""
    _fields_ = [("velocity", C.c_void_p), ("num_send_peers", C.c_int), ("send_peer", C.POINTER(C.c_int)),
                ("send_rows", C.POINTER(C.c_size_t)), ("send_index", C.c_void_p), ("num_recv_peers", C.c_int),
                ("recv_peer", C.POINTER(C.c_int)), ("recv_first_row", C.POINTER(C.c_size_t)),
                ("recv_rows", C.POINTER(C.c_size_t))]


class GhostLayout(C.Structure):
    _fields_ = [("num_ghost_lo", C.c_size_t), ("num_owned", C.c_size_t), ("num_ghost_hi", C.c_size_t),
                ("num_sent", C.c_size_t), ("halo", VelocityHalo)]


class DistProfile(C.Structure):
    _fields_ = [("body_ms", C.c_double), ("constraint_ms", C.c_double), ("halo_wait_ms", C.c_double),
                ("timed_iterations", C.c_size_t), ("halo_post_ms", C.c_double), ("record_ms", C.c_double),
                ("halo_path", C.c_int), ("record_path", C.c_int)]


# host-callback transport (mhip_comm_create_host): device pointers arrive as integers
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_size_t))
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
COMM_ID_BYTES = 128

_vp, _sz, _d, _i = C.c_void_p, C.c_size_t, C.c_double, C.c_int

# name -> argtypes; every function returns int status.  Must list every symbol include/mundy_hip.h declares
# (tests/test_capi_symbols.py checks the header against this table and against the built library).
SIGNATURES = {
    "mhip_device_info": [C.POINTER(C.c_int), C.c_char_p, _sz],
    "mhip_malloc": [C.POINTER(_vp), _sz],
    "mhip_free": [_vp],
    "mhip_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "mhip_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "mhip_stream_synchronize": [_vp],
    "mhip_compute_aabb_spheres": [_sz, _vp, _vp, _vp, _vp],
    "mhip_compute_aabb_spherocylinders": [_sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_compute_aabb_ellipsoids": [_sz, _vp, _vp, _vp, _vp, _vp],
    "mhip_compute_aabb_ellipsoids_conservative": [_sz, _vp, _vp, _vp, _vp, _vp],
    "mhip_compute_aabb_segments": [_sz, _vp, _vp, _vp],
    "mhip_bounding_radius_spherocylinders": [_sz, _vp, _vp, _vp, _vp],
    "mhip_bounding_radius_ellipsoids": [_sz, _vp, _vp, _vp],
    "mhip_spherocylinder_segments": [_sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_distance_sphere_sphere": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_distance_point_segment": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_distance_point_sphere": [_sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_distance_segment_sphere": [_sz] + [_vp] * 9,
    "mhip_distance_segment_segment": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_distance_ellipsoid_ellipsoid": [_sz] + [_vp] * 12,
    "mhip_distance_point_ellipsoid": [_sz] + [_vp] * 8,
    "mhip_contact_ellipsoids": [_sz] + [_vp] * 11,
    "mhip_compute_aabb_mixed": [_sz] + [_vp] * 7,
    "mhip_compute_aabb_mixed_conservative": [_sz] + [_vp] * 7,
    "mhip_contact_mixed": [_sz] + [_vp] * 11 + [C.POINTER(_sz), _vp],
    "mhip_contact_mixed_set_contraction": [_i],
    "mhip_contact_mixed_set_sphere_ellipsoid_route": [_i],
    "mhip_contact_mixed_last_evaluations": [C.POINTER(C.c_ulonglong), _vp],
    "mhip_ellipsoid_last_evaluations": [C.POINTER(C.c_ulonglong), _vp],
    "mhip_contact_mixed_periodic": [_sz] + [_vp] * 5 + [C.POINTER(_d)] + [_vp] * 6 + [C.POINTER(_sz), _vp],
    "mhip_contact_spherocylinders_periodic": [_sz, _vp, _vp, _vp, C.POINTER(_d)] + [_vp] * 9,
    "mhip_contact_spheres": [_sz, _vp, _vp, _vp, C.POINTER(_d), _vp, _vp, _vp],
    "mhip_contact_spherocylinders": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_broadphase_create": [C.POINTER(_vp)],
    "mhip_broadphase_destroy": [_vp],
    "mhip_broadphase_build": [_vp, C.POINTER(BroadphaseConfig), _sz, _vp, _vp, _vp, C.POINTER(_sz), _vp],
    "mhip_broadphase_get_pairs": [_vp, _vp, _vp, _vp, _vp],
    "mhip_broadphase_needs_rebuild": [_vp, _sz, _vp, C.POINTER(_i), _vp],
    "mhip_select_contacts": [_sz, _vp, _d, _vp, C.POINTER(_sz), _vp],
    "mhip_curve_keys": [_sz, _vp, C.POINTER(_d), C.POINTER(_d), _i, _vp, _vp, _vp],
    "mhip_sort_by_key_u64": [_sz, _vp, _vp, _vp],
    "mhip_broadphase_set_sets": [_vp, _sz, _vp, _vp, _vp],
    "mhip_broadphase_set_exclusions": [_vp, _sz, _vp, _vp, _sz, _vp],
    "mhip_broadphase_set_identities": [_vp, _sz, _vp, _vp, _vp],
    "mhip_broadphase_get_ident_pairs": [_vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_broadphase_method_used": [_vp, C.POINTER(_i)],
    "mhip_broadphase_minimum_image_complete": [_vp, C.POINTER(_i)],
    "mhip_links_export_coo": [_vp, C.c_uint64, _i, _i, _vp, _vp, _vp, _vp],
    "mhip_links_export_crs": [_vp, C.c_uint64, C.c_uint, _vp, _vp, _vp, _vp, _vp],
    "mhip_deep_copy": [_sz, _vp, _vp, _vp],
    "mhip_fill": [_sz, _vp, _d, _vp],
    "mhip_axpby": [_sz, _d, _vp, _d, _vp, _vp],
    "mhip_wrapped_axpbyz": [_sz, _d, _vp, _d, _vp, _vp, C.POINTER(Space), _vp],
    "mhip_diff_dot2": [_sz, _vp, _vp, C.POINTER(_d), _vp],
    "mhip_diff_dot4": [_sz, _vp, _vp, _vp, _vp, C.POINTER(_d), _vp],
    "mhip_residual": [_sz, _i, _vp, _vp, C.POINTER(Space), C.POINTER(_d), _vp],
    "mhip_bb_step": [_sz, _vp, _vp, _vp, _vp, C.POINTER(_d), _vp],
    "mhip_gemv": [_sz, _vp, _vp, _vp, _vp],
    "mhip_contact_op_create": [C.POINTER(_vp), _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp],
    "mhip_contact_op_create_rods": [C.POINTER(_vp), _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp],
    "mhip_contact_op_refresh": [_vp, _vp, _vp, _vp, _vp],
    "mhip_contact_op_refresh_rods": [_vp, _vp, _vp, _vp, _vp, _vp],
    "mhip_contact_op_destroy": [_vp],
    "mhip_contact_op_apply": [_vp, _vp, _vp, _vp],
    "mhip_contact_op_body_velocity": [_vp, C.POINTER(_vp)],
    "mhip_contact_op_set_profiling": [_vp, _i],
    "mhip_contact_op_set_tiering": [_vp, _i],
    "mhip_contact_op_tier_stats": [_vp, C.POINTER(_sz), C.POINTER(_d), C.POINTER(_sz), C.POINTER(_sz)],
    "mhip_contact_op_set_work_mapping": [_vp, _i, _i],
    "mhip_bbpgd_stage_snapshot_active": [_vp, _vp],
    "mhip_contact_op_get_profile": [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_sz)],
    "mhip_bbpgd_solve_dense": [_sz, _vp, _vp, C.POINTER(Space), C.POINTER(PgdConfig), _vp, _vp, _vp, _vp,
                               C.POINTER(SolveResult), _vp],
    "mhip_bbpgd_solve_contact": [_vp, _vp, C.POINTER(Space), C.POINTER(PgdConfig), _vp, _vp, _vp, _vp,
                                 C.POINTER(SolveResult), _vp],
    "mhip_bbpgd_solve_contact_friction": [_vp, _vp, _d, C.POINTER(PgdConfig), _vp, _vp, C.POINTER(SolveResult), _vp],
    "mhip_apgd_solve_contact_friction": [_vp, _vp, _d, C.POINTER(PgdConfig), _vp, _vp, C.POINTER(SolveResult), _vp],
    "mhip_solve_small_cqpp_batch": [_sz, _i, _vp, _vp, C.POINTER(Space), C.POINTER(PgdConfig), _vp, _vp, _vp, _vp,
                                    _vp, _vp],
    "mhip_scrap_bbpgd_solve_contact": [_vp, _vp, _d, C.c_uint, _vp, _vp, _vp, _vp, C.POINTER(SolveResult),
                                       C.POINTER(_d), _vp],
    "mhip_contact_op_set_partition": [_vp, _sz, _sz, _vp, _vp],
    "mhip_bbpgd_stage_begin": [_vp, _vp, C.POINTER(Space), C.POINTER(PgdConfig), _vp, _vp, _vp, _vp, _vp],
    "mhip_bbpgd_stage_body": [_vp, _i, _vp],
    "mhip_bbpgd_stage_constraint": [_vp, _i, _vp, _vp],
    "mhip_bbpgd_stage_constraint_range": [_vp, _i, _sz, _sz, _vp],
    "mhip_bbpgd_stage_reduce": [_vp, _i, _vp, _vp],
    "mhip_bbpgd_stage_finalize": [_vp, _i, _vp, _i, _vp],
    "mhip_bbpgd_stage_poll": [_vp, C.POINTER(SolveResult), C.POINTER(_i), _vp],
    "mhip_bbpgd_stage_end": [_vp, C.POINTER(SolveResult), _vp],
    "mhip_filter_pairs_owned": [_sz, _vp, _sz, _sz, _vp, _vp, C.POINTER(_sz), _vp],
    "mhip_partition_pairs_owned": [_sz, _vp, _sz, _sz, _vp, _vp, C.POINTER(_sz), C.POINTER(_sz), _vp],
    "mhip_select_aabb_overlap": [_sz, _vp, _d, C.POINTER(_d), _vp, C.POINTER(_sz), _vp],
    "mhip_aabb_chunk_bounds": [_sz, _vp, _d, _i, _vp, _vp],
    "mhip_select_aabb_overlap_any": [_sz, _vp, _d, _i, _vp, _vp, C.POINTER(_sz), _vp],
    "mhip_aabb_bounds": [_sz, _vp, _d, C.POINTER(_d), _vp],
    "mhip_bbpgd_solve_contact_unfused": [_vp, _vp, C.POINTER(Space), C.POINTER(PgdConfig), _vp, _vp, _vp, _vp,
                                         C.POINTER(SolveResult), _vp],
    "mhip_periodic_sep": [_sz, C.POINTER(_d), _vp, _vp, _vp, _vp],
    "mhip_wrap_rigid": [_sz, C.POINTER(_d), _vp, _vp],
    "mhip_unit_cell_inverse": [C.POINTER(_d), C.POINTER(_d)],
    "mhip_periodic_sep_triclinic": [_sz, C.POINTER(_d), _vp, _vp, _vp, _vp],
    "mhip_wrap_rigid_triclinic": [_sz, C.POINTER(_d), _vp, _vp],
    "mhip_shift_image_triclinic": [_sz, C.POINTER(_d), _vp, _vp, _vp, _vp],
    "mhip_contact_spheres_triclinic": [_sz, _vp, _vp, _vp, C.POINTER(_d), _vp, _vp, _vp],
    "mhip_integrate_euler": [_sz, _d, _vp, _vp, _vp, _vp],
    "mhip_morton_order": [_sz, _vp, C.POINTER(_d), _d, _vp, _vp],
    "mhip_set_tracing": [_i],
    "mhip_curve_order": [_sz, _vp, C.POINTER(_d), C.POINTER(_d), _i, _vp, _vp, _vp],
    "mhip_gather_rows": [_sz, _sz, _vp, _vp, _vp, _vp],
    "mhip_copy_strided": [_sz, _sz, _vp, _sz, _vp, _sz, _vp],
    "mhip_contact_op_sizes": [_vp, C.POINTER(_sz), C.POINTER(_sz)],
    "mhip_contact_op_set_drift_source": [_vp, _i],
    "mhip_contact_op_get_drift_source": [_vp, C.POINTER(_i)],
    "mhip_comm_unique_id": [C.c_char_p],
    "mhip_comm_create_rccl": [C.POINTER(_vp), C.c_char_p, _i, _i],
    "mhip_comm_create_host": [C.POINTER(_vp), _i, _i, EXCHANGE_FN, ALL_GATHER_FN, _vp],
    "mhip_comm_destroy": [_vp],
    "mhip_comm_info": [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "mhip_comm_mailbox_open": [_vp, C.POINTER(_i), _vp],
    "mhip_comm_mailbox_close": [_vp],
    "mhip_comm_halo_ipc_enable": [_vp, _i],
    "mhip_comm_set_exchange_timeout": [_vp, C.c_double],
    "mhip_comm_inject_fault": [_vp, C.c_uint],
    "mhip_comm_halo_ipc_active": [_vp, C.POINTER(_i)],
    "mhip_comm_all_gather": [_vp, _vp, _sz, _vp, _vp],
    "mhip_comm_exchange_start": [_vp, _i, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_sz), _i, C.POINTER(_i),
                                 C.POINTER(_vp), C.POINTER(_sz), _vp],
    "mhip_comm_exchange_finish": [_vp, _vp],
    "mhip_ghost_layout_from_counts": [_i, _i, _sz, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i),
                                      C.POINTER(_i), C.POINTER(_sz), C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz),
                                      C.POINTER(_sz)],
    "mhip_ghost_plan": [_vp, _sz, _vp, _d, C.POINTER(GhostLayout), _vp],
    "mhip_ghost_exchange": [_vp, _sz, _vp, _vp, _vp],
    "mhip_hilbert_key_table": [_i, _vp],
    "mhip_body_work_weights": [_sz, _vp, _sz, _sz, _vp, _vp],
    "mhip_compose_keys_u64": [_sz, _vp, _vp, _i, _vp, _vp],
    "mhip_fill_sequence": [_sz, _d, _vp, _vp],
    "mhip_curve_cut": [_vp, _sz, _vp, _vp, _sz, _vp, _vp],
    "mhip_migrate_plan": [_vp, _sz, _vp, _vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), _vp],
    "mhip_migrate_exchange": [_vp, _sz, _vp, _vp, _vp],
    "mhip_bbpgd_solve_contact_distributed": [_vp, _vp, C.POINTER(VelocityHalo), _sz, _vp, C.POINTER(Space),
                                             C.POINTER(PgdConfig), _vp, _vp, _vp, _vp, C.c_uint,
                                             C.POINTER(SolveResult), C.POINTER(DistProfile), _vp],
}
OTHER_SYMBOLS = {"mhip_last_error": ([], C.c_char_p), "mhip_version": ([], C.c_int),
                 "mhip_release_cached_workspaces": ([], C.c_int)}

_lib = None


def load():
    """dlopen the in-tree library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MhipError("%s is missing: run `python -m mundy_amd.build` (or __graft_entry__.build()); "
                            "mundy_amd has no CPU fallback" % LIB_PATH)
        # torch ships its own libamdhip64; it must be mapped first so that torch's allocator/streams and this
        # library share ONE HIP runtime (loading the system runtime first leaves the process with a runtime that
        # cannot see torch's device context)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, (argtypes, restype) in OTHER_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib


def check(status):
    """status -> the exception type the reference throws for that class of error (throw_assert.hpp:135-203)."""
    if status == SUCCESS:
        return
    msg = load().mhip_last_error().decode()
    if status == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)  # std::invalid_argument
    if status == ERR_LOGIC:
        raise AssertionError(msg)  # std::logic_error
    if status == ERR_RUNTIME:
        raise RuntimeError(msg)  # std::runtime_error
    raise MhipError(msg)
