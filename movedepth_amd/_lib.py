"""ctypes binding of libmovedepth_hip.so (the C ABI declared in include/movedepth_hip.h).

There is no CPU fallback: if the library is missing or a call fails this raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C movedepth_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOVEDEPTH_HIP_LIB") or os.path.join(_HERE, "libmovedepth_hip.so")  # override: kernel A/B builds

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong
_sz = ctypes.c_size_t

# md_costvol_fwd / _bwd (and their _bf16 / _f16 twins): ..., B, C, G, h, w, D, feat_cl, ...
_CV_FWD = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _ll, _ll, _ll, ctypes.c_uint, _vp]   # (ABI 17: ..., out strides, flags, stream)
_u = ctypes.c_uint
# (ABI 17: d_ref, d_src, flags, census, shares, n_shares, cost, stream)
_CV_BWD = [_vp, _ll, _ll, _ll, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _u, _vp, _vp, _i, _vp, _vp]
CV_GATHER_TABLE = 1   # MD_CV_GATHER_TABLE (md_costvol_bwd*)
CV_FINE_SLICES = 2    # MD_CV_FINE_SLICES (md_costvol_fwd*)

# name -> (restype, argtypes); must list every symbol the header declares (tests/test_cabi.py checks)
SIGNATURES = {
    "md_last_error": (ctypes.c_char_p, []),
    "md_abi_version": (_i, []),
    "md_schedule_depth_range": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "md_costvol_fwd": (_i, _CV_FWD),
    "md_costvol_bwd": (_i, _CV_BWD),
    "md_costvol_bwd_plan": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "md_costvol_fwd_bf16": (_i, _CV_FWD),
    "md_costvol_bwd_bf16": (_i, _CV_BWD),
    "md_costvol_fwd_f16": (_i, _CV_FWD),
    "md_costvol_bwd_f16": (_i, _CV_BWD),
    "md_fuse_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _vp, _vp, _vp]),
    "md_fuse_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _vp, _vp]),
    "md_warp_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "md_warp_bwd_ws_bytes": (_sz, [_i, _i, _i]),
    "md_warp_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "md_disp_to_depth_up_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    "md_disp_to_depth_up_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    "md_ssim": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "md_reproj_loss_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "md_reproj_loss_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "md_masked_min_ws_bytes": (_sz, [_i, _i, _i]),
    "md_masked_min_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "md_masked_min_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "md_photo_fwd_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "md_photo_fwd": (_i, [_vp, _vp, _vp]),
    "md_photo_bwd_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "md_photo_bwd": (_i, [_vp, _vp, _vp]),
    "md_photo_desc_bytes": (_sz, []),
    "md_pack_rgbx": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "md_smooth_multi_ws_bytes": (_sz, [_i, _i]),
    "md_smooth_multi_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "md_smooth_multi_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "md_smooth_ws_bytes": (_sz, [_i, _i, _i]),
    "md_smooth_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "md_smooth_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "md_softmax_entropy_localmax_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "md_softmax_entropy_localmax_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "md_convex_upsample_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "md_convex_upsample_bwd_ws_bytes": (_sz, [_i, _i, _i]),
    "md_convex_upsample_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "md_conv3d_c1_fwd": (_i, [_vp, _vp, _ll, _ll, _vp, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_c1_bwd_data": (_i, [_vp, _vp, _ll, _ll, _vp, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_c1_bwd_weight_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "md_conv3d_c1_bwd_weight": (_i, [_vp, _vp, _vp, _ll, _ll, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_c16_fwd": (_i, [_vp, _i, _vp, _ll, _ll, _ll, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_c16_bwd_data": (_i, [_vp, _vp, _ll, _ll, _ll, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_c16_bwd_weight_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "md_conv3d_c16_bwd_weight": (_i, [_vp, _i, _vp, _vp, _ll, _ll, _ll, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_cb_fwd": (_i, [_vp, _vp, _ll, _ll, _ll, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_cb_bwd_data": (_i, [_vp, _vp, _ll, _ll, _ll, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "md_conv3d_cb_bwd_weight_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "md_conv3d_cb_bwd_weight": (_i, [_vp, _vp, _vp, _ll, _ll, _ll, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]),
    "md_bn_relu_ws_bytes": (_sz, []),
    "md_bn_relu_stats": (_i, [_vp, _ll, _i, _vp, _vp, _vp]),
    "md_bn_relu_finalize": (_i, [_vp, _ll, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "md_bn_relu_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp]),
    "md_bn_relu_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp]),
    "md_bn_relu_bwd_dx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _i, _vp, _vp]),
    "md_bn_ws_bytes": (_sz, []),
    "md_bn_stats": (_i, [_vp, _i, _ll, _i, _vp, _vp, _vp]),
    "md_bn_apply": (_i, [_vp, _i, _vp, _ll, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _ll, _i, _vp]),
    "md_bn_eval": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _ll, _i, _vp]),
    "md_bn_bwd_reduce": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp]),
    "md_bn_bwd_dx": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _ll, _ll, _i, _vp, _vp]),
    "md_pose_matrix_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "md_pose_matrix_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "md_backproject": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "md_project3d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "md_kernel_timing_enable": (_i, [_i]),
    "md_costvol_stats": (_i, [_i, ctypes.POINTER(ctypes.c_ulonglong)]),
    "md_costvol_stats_wg": (_i, [ctypes.POINTER(ctypes.c_ulonglong), _i]),
    "md_kernel_timing_list": (_i, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), _i]),
    "md_kernel_timing_read": (_i, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(_i)]),
}

PHOTO_MAX_FRAMES = 4
PHOTO_MAX_SCALES = 4


class PhotoDesc(ctypes.Structure):
    """md_photo_desc of include/movedepth_hip.h (field for field)."""
    _fields_ = [
        ("B", _i), ("H", _i), ("W", _i), ("F", _i), ("S", _i),
        ("is_disp", _i), ("identity", _i), ("mvs_mode", _i), ("no_ssim", _i),
        ("ssim_w", _f), ("min_depth", _f), ("max_depth", _f),
        ("dh", _i * PHOTO_MAX_SCALES), ("dw", _i * PHOTO_MAX_SCALES),
        ("target", _vp), ("src", _vp * PHOTO_MAX_FRAMES), ("T", _vp * PHOTO_MAX_FRAMES), ("K", _vp), ("invK", _vp),
        ("dz", _vp * PHOTO_MAX_SCALES), ("ident_min", _vp), ("noise", _vp), ("ext_mask", _vp),
        ("warped", (_vp * PHOTO_MAX_FRAMES) * PHOTO_MAX_SCALES), ("pix", (_vp * PHOTO_MAX_FRAMES) * PHOTO_MAX_SCALES),
        ("oob", _vp * PHOTO_MAX_FRAMES), ("depth_out", _vp * PHOTO_MAX_SCALES), ("mn", _vp * PHOTO_MAX_SCALES),
        ("mask", _vp * PHOTO_MAX_SCALES), ("sel", _vp * PHOTO_MAX_SCALES), ("loss", _vp),
        ("gloss", _vp * PHOTO_MAX_SCALES), ("d_dz", _vp * PHOTO_MAX_SCALES), ("d_T", _vp * PHOTO_MAX_FRAMES),
    ]


_lib = None


class MovedepthHipError(RuntimeError):
    pass


def load():
    """Loads the library once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own HIP runtime; if libmovedepth_hip.so is loaded before torch, the process
    # ends up with two runtimes and kernel launches from this library fail with "no ROCm-capable device is detected"
    # (seen with __graft_entry__.build() followed by smoke() in one process)
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise MovedepthHipError(
            "libmovedepth_hip.so not found at %s: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C movedepth_amd/csrc` (hipcc --offload-arch=gfx950)." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.md_photo_desc_bytes() != ctypes.sizeof(PhotoDesc):
        raise MovedepthHipError("md_photo_desc is %d bytes in the library, %d in the binding: stale build?"
                                % (lib.md_photo_desc_bytes(), ctypes.sizeof(PhotoDesc)))
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        msg = load().md_last_error()
        raise MovedepthHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    check(getattr(load(), name)(*args), name)
