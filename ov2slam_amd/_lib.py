"""ctypes binding of libov2hip.so (the C ABI declared in include/ov2slam_hip.h).

The product path is the HIP library and nothing else: if the shared object is missing or a symbol
is absent this module raises; there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OV2SLAM_HIP_LIB selects another build of the same library (kernel experiments); the default is the in-tree one
LIB_PATH = os.environ.get("OV2SLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libov2hip.so")

vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)
ip = C.POINTER(C.c_int)
fp = C.POINTER(C.c_float)

# name -> (restype, argtypes).  Must list every symbol of include/ov2slam_hip.h (tests check this).
SIGNATURES = {
    "ov2_ctx_create": (C.c_int, [C.c_int, vpp]),
    "ov2_ctx_create_ex": (C.c_int, [C.c_int, C.c_int, vpp]),
    "ov2_ctx_destroy": (None, [vp]),
    "ov2_last_error": (C.c_char_p, [vp]),
    "ov2_status_string": (C.c_char_p, [C.c_int]),
    "ov2_ctx_synchronize": (C.c_int, [vp]),
    "ov2_timer_start": (C.c_int, [vp]),
    "ov2_timer_stop": (C.c_int, [vp, fp]),
    "ov2_ktime_enable": (C.c_int, [vp, C.c_int]),
    "ov2_ktime_report": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.POINTER(C.c_longlong), ip]),
    "ov2_dev_alloc": (C.c_int, [vp, C.c_size_t, vpp]),
    "ov2_dev_free": (C.c_int, [vp, vp]),
    "ov2_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ov2_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ov2_memcpy_d2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ov2_images_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vpp]),
    "ov2_images_upload": (C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
    "ov2_images_destroy": (None, [vp]),
    "ov2_pyramid_build": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_int, C.c_int, vpp]),
    "ov2_pyramid_build_images": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vpp]),
    "ov2_pyr_retain": (None, [vp]),
    "ov2_pyr_release": (None, [vp]),
    "ov2_pyr_release_from": (None, [vp, vp]),
    "ov2_pyr_batch": (C.c_int, [vp]),
    "ov2_pyr_nlevels": (C.c_int, [vp]),
    "ov2_pyr_level_size": (C.c_int, [vp, C.c_int, ip, ip, ip]),
    "ov2_pyr_download_level": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "ov2_klt_track_fb": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                   C.c_int, vp, vp, vp]),
    "ov2_klt_track_fb_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_int, vp, vp, vp, vp, vp]),
    "ov2_klt_tracking_frame_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                             C.c_float, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ov2_klt_set_lanes": (C.c_int, [vp, C.c_int]),
    "ov2_klt_set_yield": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "ov2_line_min_sad": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "ov2_line_min_sad_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "ov2_stereo_matching": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                      vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]),
    "ov2_stereo_matching_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                          vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]),
    "ov2_detect_grid": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, vp, vp, C.c_int, ip, vp]),
    "ov2_detect_grid_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_double), ip, vp, vp, C.c_int, ip, vp, C.c_int]),
    "ov2_detect_grid_batch_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int]),
    "ov2_ba_default_options": (None, [vp, C.c_float]),
    "ov2_ba_solve": (C.c_int, [vp, vp, vp, vp]),
    "ov2_ba_solve_batch": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "ov2_ba_solve_batch_dev": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "ov2_pnp_solve_batch_dev": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int,
                                          vp, vp, vp, vp]),
    "ov2_map_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "ov2_map_destroy": (None, [vp]),
    "ov2_map_add_keyframe": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]),
    "ov2_map_set_landmarks": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "ov2_map_set_poses": (C.c_int, [vp, C.c_int, vp, vp]),
    "ov2_map_remove_obs": (C.c_int, [vp, C.c_int, vp, vp]),
    "ov2_map_set_obs_stereo": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "ov2_map_remove_landmarks": (C.c_int, [vp, C.c_int, vp]),
    "ov2_map_remove_keyframe": (C.c_int, [vp, C.c_int]),
    "ov2_pose_graph_solve": (C.c_int, [vp, vp, vp, vp]),
    "ov2_map_compact": (C.c_int, [vp, ip, ip]),
    "ov2_map_obs_rows": (C.c_int, [vp, ip, ip, ip]),
    "ov2_map_local_ba_setup": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ov2_map_setup_device_view": (C.c_int, [vp, vp, vp]),
    "ov2_map_local_ba_setup_batch": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ov2_map_local_ba_update_batch": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "ov2_map_save_state": (C.c_int, [vp]),
    "ov2_map_restore_state_batch": (C.c_int, [vp, C.c_int, vp]),
    "ov2_map_download": (C.c_int, [vp, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ov2_triangulate_pairs": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float,
                                        vp, vp, vp, vp]),
    "ov2_triangulate_pairs_dev": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float,
                                            vp, vp, vp, vp]),
    "ov2_dbg_rowload16": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, vp]),
    "ov2_describe_brief": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "ov2_describe_brief_dev": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp]),
    "ov2_match_to_map": (C.c_int, [vp, vp, C.c_float, C.c_float, vp, vp]),
    "ov2_pnp_solve_batch": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int,
                                      vp, vp, vp]),
}

_lib = None


class Ov2Error(RuntimeError):
    pass


def load():
    """dlopen libov2hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Ov2Error(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Ov2Error(f"libov2hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
