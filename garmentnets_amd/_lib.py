"""ctypes binding of libgarmentnets_hip.so (C ABI: include/garmentnets_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C garmentnets_amd/csrc``.  There is NO CPU
fallback: if the shared object is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GARMENTNETS_HIP_LIB: another build of the same C ABI (deployment layouts; A/B runs of two builds in two processes)
LIB_PATH = os.environ.get("GARMENTNETS_HIP_LIB") or os.path.join(_HERE, "libgarmentnets_hip.so")

GN_OK, GN_EINVAL, GN_ELAUNCH, GN_ECAP = 0, -1, -2, -3

_vp, _i32, _i64, _f32, _f64, _sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double,
                                   ctypes.c_size_t)

# name -> argtypes (every function returns int unless listed in _RESTYPES)
PROTOTYPES = {
    "gn_version": [],
    "gn_device_info": [_vp, _vp],
    "gn_segment_ptr": [_vp, _i64, _i32, _vp, _vp],
    "gn_fps": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "gn_fps_nested": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "gn_ball_query": [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp, _vp],
    "gn_sa_gather": [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp],
    "gn_segment_max": [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_sa_fused_supported": [_i32, _i32, _i32, _i32],
    "gn_sa_fused": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_sa_fused_scoped": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_sa_gather_scoped": [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp],
    "gn_global_max_pool": [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _vp],
    "gn_knn_interpolate": [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_linear": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _i32, _vp],
    "gn_nocs_head": [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "gn_grid_features": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp],
    "gn_grid_scatter_workspace_bytes": [_i64, _i32, _i32],
    "gn_grid_scatter": [_vp, _i32, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _sz, _i32, _vp],
    "gn_channel_stats": [_vp, _i32, _i64, _i32, _vp, _vp, _vp],
    "gn_groupnorm_affine": [_vp, _vp, _i32, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    "gn_conv3d_gcr": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "gn_conv3d_gcr_split": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _sz, _vp],
    "gn_conv3d_occupancy_workspace_bytes": [_i32, _i32, _i32, _i32],
    "gn_conv_affine_pack_bytes": [_i32, _i32, _i32],
    "gn_conv_affine_pack": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "gn_conv3d_gcr_split_persample": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _sz, _vp],
    "gn_conv_affine_pack_wino_bytes": [_i32, _i32, _i32],
    "gn_conv_affine_pack_wino": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "gn_conv3d_gcr_split_wino": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp],
    "gn_conv3d_gcr_split_wino_partial": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "gn_affine_act": [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "gn_upconv_partial": [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "gn_grid_tile_flags": [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "gn_maxpool3d_2": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "gn_grid_stats": [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "gn_trilinear_sample": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i64, _i64, _vp, _i32, _vp],
    "gn_trilinear_sample_batch": [_vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _vp],
    "gn_implicit_decode": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32,
                           _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp],
    "gn_ggm3d": [_vp, _i32, _i32, _i32, _f64, _vp, _vp, _vp],
    "gn_minmax": [_vp, _i64, _vp, _vp],
    "gn_ggm3d_batch": [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _vp],
    "gn_ggm3d_range_workspace_bytes": [_i32, _i32, _i32, _i32],
    "gn_ggm3d_batch_ex": [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _i32, _vp, _vp, _sz, _vp],
    "gn_minmax_batch": [_vp, _i32, _i64, _vp, _vp],
    "gn_mc33_batch_workspace_bytes": [_i32, _i32, _i32, _i32],
    "gn_mc33_batch": [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _sz, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "gn_mc33_batch_profiled": [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _sz, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "gn_mc33_workspace_bytes": [_i32, _i32, _i32],
    "gn_mc33": [_vp, _i32, _i32, _i32, _f64, _vp, _sz, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "gn_gather_nn": [_vp, _i32, _i32, _i32, _vp, _i64, _f64, _vp, _vp],
    "gn_gather_nn_batch": [_vp, _i32, _i32, _i32, _i32, _vp, _i64, _f64, _vp, _vp],
    "gn_mesh_compact_workspace_bytes": [_i64, _i64],
    "gn_mesh_largest_component_workspace_bytes": [_i64],
    "gn_mesh_largest_component": [_vp, _i64, _i64, _vp, _sz, _vp, _vp, _vp, _vp],
    "gn_mesh_compact": [_vp, _i32, _vp, _vp, _i64, _i64, _vp, _sz, _vp, _vp, _vp, _vp],
    "gn_scale_verts": [_vp, _i64, _f64, _vp, _vp],
    "gn_implicit_decode_split": [_vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_implicit_decode_split_batch": [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_implicit_decode_batch": [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp],
    "gn_implicit_decode_lattice_split": [_vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "gn_decoder_input_scale": [_vp, _i64, _i32, _i32, _f32, _vp, _vp],
    "gn_nearest_neighbor": [_vp, _i64, _vp, _i64, _vp, _vp, _vp],
}
_RESTYPES = {"gn_conv_affine_pack_bytes": _sz, "gn_conv_affine_pack_wino_bytes": _sz, "gn_conv3d_occupancy_workspace_bytes": _sz, "gn_mc33_workspace_bytes": _sz, "gn_ggm3d_range_workspace_bytes": _sz, "gn_mc33_batch_workspace_bytes": _sz, "gn_grid_scatter_workspace_bytes": _sz, "gn_mesh_compact_workspace_bytes": _sz, "gn_mesh_largest_component_workspace_bytes": _sz}

_lib = None


class GarmentNetsHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and declare every prototype of include/garmentnets_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GarmentNetsHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C garmentnets_amd/csrc`). garmentnets_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.gn_last_error.restype = ctypes.c_char_p
    lib.gn_last_error.argtypes = []
    lib.gn_last_kernel.restype = ctypes.c_char_p
    lib.gn_last_kernel.argtypes = []
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != GN_OK:
        msg = lib.gn_last_error().decode(errors="replace")
        if rc == GN_EINVAL:
            raise ValueError(f"{name}: {msg}")
        raise GarmentNetsHipError(f"{name} failed (code {rc}): {msg}")
    return rc
