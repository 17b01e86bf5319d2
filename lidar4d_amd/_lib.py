"""ctypes binding of the C ABI declared in include/lidar4d_hip.h (liblidar4d_hip.so, gfx950).

The product path has no CPU fallback: if the shared library is missing or a tensor is not on a HIP
device, these wrappers raise.  (The CPU restatement lives in oracle/ and is test infrastructure.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L4D_LIB", os.path.join(_HERE, "liblidar4d_hip.so"))  # L4D_LIB: ablation builds (tools/)

L4D_MAX_LEVELS = 16
L4D_MAX_TIME_SLICES = 8
L4D_MAX_PLANE_SCALES = 8
ABI_VERSION = 2


class GridDesc(C.Structure):
    _fields_ = [
        ("n_dims", C.c_int32),
        ("n_features", C.c_int32),
        ("n_levels", C.c_int32),
        ("hashed_mask", C.c_uint32),
        ("scale", C.c_float * L4D_MAX_LEVELS),
        ("res", C.c_uint32 * L4D_MAX_LEVELS),
        ("size", C.c_uint32 * L4D_MAX_LEVELS),
        ("offset", C.c_uint32 * L4D_MAX_LEVELS),
    ]


class FieldDesc(C.Structure):
    _fields_ = [
        ("hash_static", GridDesc),
        ("hash_static_table", C.c_void_p),
        ("hash_dynamic", GridDesc * 3),
        ("hash_dynamic_tables", (C.c_void_p * L4D_MAX_TIME_SLICES) * 3),
        ("hash_dynamic_pairs", C.c_void_p * 3),
        ("n_slices", C.c_int32),
        ("n_scales", C.c_int32),
        ("plane_channels", C.c_int32),
        ("plane_res", C.c_int32 * (L4D_MAX_PLANE_SCALES * 4)),
        ("plane_off", C.c_int64 * (L4D_MAX_PLANE_SCALES * 6)),
        ("planes_cl", C.c_void_p),
    ]


class FieldGrads(C.Structure):
    _fields_ = [
        ("hash_static_table", C.c_void_p),
        ("hash_dynamic_tables", (C.c_void_p * L4D_MAX_TIME_SLICES) * 3),
        ("planes_cl", C.c_void_p),
    ]


P, I32, I64, F32, F64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
GD, FD, FG = C.POINTER(GridDesc), C.POINTER(FieldDesc), C.POINTER(FieldGrads)
PI32 = C.POINTER(C.c_int32)
PI64 = C.POINTER(C.c_int64)
PP = C.POINTER(C.c_void_p)

# name -> argtypes (all return int status, except where noted); mirrors include/lidar4d_hip.h
SIGNATURES = {
    "l4d_hashgrid_fwd": [GD, P, I64, I32, PI32, P, P, I32, P],
    "l4d_hashgrid_fwd_workspace": [GD, I64],
    "l4d_hashgrid_fwd_ws": [GD, P, I64, I32, PI32, P, P, I32, P, P],
    "l4d_hashgrid_bwd": [GD, P, I64, I32, PI32, P, I32, I32, F32, P, P],
    "l4d_hashgrid_t_fwd": [GD, P, I64, I32, PI32, PP, I32, P, P, I32, I32, P],
    "l4d_hashgrid_t_fwd_workspace": [GD, I64],
    "l4d_hashgrid_t_fwd_ws": [GD, P, I64, I32, PI32, PP, I32, P, P, I32, I32, P, P],
    "l4d_hashgrid_t_bwd": [GD, P, I64, I32, PI32, I32, P, P, I32, I32, F32, PP, P, P, P],
    "l4d_hashgrid_t_bwd_workspace": [GD, I64],
    "l4d_planes_relayout": [PP, PI32, I32, I32, P, PI64, I32, P],
    "l4d_planes_fwd": [P, PI64, PI32, I32, I32, P, I64, I32, P, P, P],
    "l4d_planes_bwd": [P, PI64, PI32, I32, I32, P, I64, I32, P, P, P, P, P],
    "l4d_freq_fwd": [P, I64, I32, I32, P, I32, P],
    "l4d_mlp_fwd": [P, I64, P, I32, I32, P, P, P, P],
    "l4d_mlp_bwd": [P, P, P, I64, P, I32, I32, P, P, P, F32, P, I32, I32, P],
    "l4d_sample_rays": [P, P, P, P, I64, I32, F32, F32, F32, P, P, P],
    "l4d_sample_rays_xt": [P, P, P, P, P, I64, I32, F32, F32, F32, P, P, P],
    "l4d_composite_fwd": [P, P, I64, I32, F32, F32, I32, P, P, P, P, P, P, P],
    "l4d_composite_image": [P, P, I64, I32, I32, P, P],
    "l4d_composite_bwd": [P, P, P, P, I64, I32, I32, F32, F32, I32, P, P, P, P, P, P, P],
    "l4d_attr_gather": [P, P, I64, I32, P, I32, P, I32, P, I32, P],
    "l4d_attr_scatter": [P, P, I64, P, P, P, P, P],
    "l4d_attr_scatter_bwd": [P, P, I64, P, P, F32, P, P, P],
    "l4d_attr_gather_bwd": [P, P, I64, P, P, I32, I32, I32, P, I32, P],
    "l4d_attr_mlp_fwd": [P, P, I64, I32, P, I32, P, I32, I32, I32, P, P, P, P, P, P, I32, P],
    "l4d_mlp_fwd_sigma": [P, I64, I32, I32, P, P, P, P, P],
    "l4d_attr_mlp_bwd": [P, P, I64, I32, I32, I32, I32, P, P, P, P, P, F32, P],
    "l4d_attr_mlp_bwd_gathered": [P, P, I64, I32, P, I32, P, I32, I32, I32, P, P, P, P, P, F32, P, P, I32, F32, P, I32, P],
    "l4d_sigma_from_h": [P, I64, P, P],
    "l4d_sigma_bwd": [P, P, I64, F32, P, P],
    "l4d_sigma_bwd_rows": [P, P, I64, F32, P, P],
    "l4d_time_setup": [P, I32, P, P],
    "l4d_density_encode_fwd": [FD, P, P, P, I64, P, I32, P, P, P],
    "l4d_density_encode_sigma_fwd": [FD, P, P, P, I64, P, I32, P, P, P, I32, P, P, P, P],
    "l4d_plane_rows_workspace": [FD],
    "l4d_density_encode_fwd_workspace": [FD, I64],
    "l4d_density_encode_bwd": [FD, FG, P, P, P, I64, P, I32, F32, P, I32, P, P, P, P, I32, P],
    "l4d_density_encode_bwd_workspace": [FD, I64],
    "l4d_field_width": [FD],
    "l4d_dyn_pairs_build": [PP, I32, I64, P, P],
    "l4d_chamfer_workspace": [I32, I32, I32],
    "l4d_chamfer_fwd": [P, P, I32, I32, I32, P, P, P, P, P, P],
    "l4d_chamfer_bwd": [P, P, I32, I32, I32, P, P, P, P, P, P, P],
    "l4d_lidar_ray_batch": [P, P, I32, P, F32, F32, I32, I32, P, P, P, P, P, P],
    "l4d_glue_workspace": [I32],
    "l4d_lidar_losses": [P, P, P, P, I32, F32, F32, F32, F32, F32, P, P, P, P, P, P],
    "l4d_ray_chamfer_grad": [P, P, P, P, P, P, P, I32, F32, F32, P, P, P, P],
    "l4d_scale_buffers": [P, P, I64, P, P, I64, P, P],
    "l4d_flow_xt": [P, I32, P, F32, P, P],
    "l4d_flow_warp": [P, P, I32, I32, PI32, P, P, P],  # (col0 / step: small HOST arrays)
    "l4d_flow_chamfer_grad": [P, I32, P, I32, P, P, P, P, F32, I32, P, P, P],
    "l4d_flow_loss_finish": [P, I32, P, I32, F32, P, P, I64, P, P, P],
    "l4d_flow_dy16": [P, I32, P, P, P, P, P],
    "l4d_axpy_dev": [P, P, I64, P, P],
    "l4d_pano_to_lidar_workspace": [I32, I32],
    "l4d_pano_to_lidar": [P, P, I32, I32, F64, F64, P, P, P, P],
    "l4d_lidar_to_pano_workspace": [I32, I32],
    "l4d_lidar_to_pano": [P, I64, I32, I32, F64, F64, F32, P, P, P, P],
    "l4d_cast_f32_to_f16": [P, P, I64, P],
    "l4d_adam_step": [P, P, P, P, P, I64, F32, F32, F32, F32, F32, F32, F32, P],
    "l4d_adam_step_ranges": [P, P, P, P, P, I32, PI64, PI64, P, PI32, P, P, P, F32, F32, F32, F32, P, F32, P],
    "l4d_grad_nonfinite_check": [P, I64, P, P],
    "l4d_absmax_f32": [P, I64, P, P],
    "l4d_scaler_update": [P, F32, F32, I32, P],
    "l4d_mark_time_slices": [P, I32, P, P],
    "l4d_profile_enable": [I32],
    "l4d_profile_count": [],
    "l4d_profile_get": [I32, P, P],
    "l4d_streams_config": [I32],
    "l4d_streams_mask": [],
    "l4d_streams_join": [P],
    "l4d_side_fork": [P, I32],      # returns the side stream (void*)
    "l4d_side_join": [P, I32],
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def lib():
    """Load liblidar4d_hip.so (once).  Raises HipExtensionError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C lidar4d_amd/csrc).  lidar4d_amd has no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    l.l4d_version.restype = C.c_int
    l.l4d_last_error.restype = C.c_char_p
    if l.l4d_version() != ABI_VERSION:
        raise HipExtensionError(f"ABI mismatch: library {l.l4d_version()} != binding {ABI_VERSION}; rebuild")
    for name, args in SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = C.c_int64 if name.endswith("_workspace") else (C.c_void_p if name == "l4d_side_fork" else C.c_int)
    _lib = l
    return l


def check(status, name):
    if status != 0:
        raise HipExtensionError(f"{name} failed: {lib().l4d_last_error().decode()}")


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def profile_start():
    """Per-kernel HIP-event timing inside the library (csrc/common.h L4D_LAUNCH): from now on every kernel launch is
    bracketed by events on its launch stream.  bench.py's per-kernel pass."""
    check(lib().l4d_profile_enable(1), "l4d_profile_enable")


def profile_stop():
    """-> [(kernel name, ms)] in launch order; switches the recording off."""
    l = lib()
    out = []
    name, ms = C.c_char_p(), C.c_float()
    for i in range(l.l4d_profile_count()):
        check(l.l4d_profile_get(i, C.byref(name), C.byref(ms)), "l4d_profile_get")
        out.append((name.value.decode().strip("()"), ms.value))
    check(l.l4d_profile_enable(0), "l4d_profile_enable")
    return out
