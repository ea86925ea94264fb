"""ctypes binding of libfresco_hip.so (the C ABI declared in include/fresco_hip.h).

The product path has no CPU or PyTorch-eager fallback: if the shared library is missing or does not
load, every operator of this package raises (``FrescoHipError``) instead of computing elsewhere.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRESCO_HIP_LIB: an alternative build of the same library (A/B measurements of kernel variants, tools/ab_variants.sh);
# it must exist and export every symbol like the default one -- there is no fallback either way
LIB_PATH = os.environ.get("FRESCO_HIP_LIB") or os.path.join(_HERE, "lib", "libfresco_hip.so")

OK = 0
ERRORS = {
    -1: "FRESCO_EINVAL (null pointer / bad size / inconsistent arguments)",
    -2: "FRESCO_EUNSUPPORTED (shape outside what the kernels are built for)",
    -3: "FRESCO_EWORKSPACE (workspace too small)",
    -4: "FRESCO_ELAUNCH (HIP launch failed)",
}
F16, F32 = 0, 1


class FrescoHipError(RuntimeError):
    pass


_c = ctypes
_vp, _i, _f, _sz, _i64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_int64

# name -> (restype, argtypes); must list every symbol include/fresco_hip.h declares
SIGNATURES = {
    "fresco_version": (_c.c_char_p, []),
    "fresco_last_error": (_c.c_char_p, []),
    "fresco_prof_enable": (_i, [_i]),
    "fresco_prof_disable": (_i, []),
    "fresco_prof_read": (_i, [_i, _vp, _vp, _vp]),
    "fresco_attn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fresco_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i64, _f, _f, _vp]),
    "fresco_attn_fwd_ld": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i64, _f, _f, _i64, _i64, _vp]),
    "fresco_attn_kvproj_supported": (_i, [_i, _i, _i]),
    "fresco_attn_fwd_kvproj": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _f, _i64, _vp]),
    "fresco_temporal_attn_ld": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i64, _i64, _i64, _vp]),
    "fresco_temporal_attn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_temporal_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _vp]),
    "fresco_temporal_attn_packed": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_temporal_unpack": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fresco_flow_warp": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fresco_resize_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "fresco_max_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "fresco_dilate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "fresco_linear": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i,
                           _vp]),
    "fresco_linear_rows": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i,
                                _i, _vp]),
    "fresco_attn_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_attn_f32_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fresco_attn_f32_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_attn_f32_guarded": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_attn_f32_guarded_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_fn_gemm": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64] + [_i] * 4 + [_f, _f] + [_i] * 7 + [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "fresco_fn_colstats_finish": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "fresco_fn_colstats_workspace_bytes": (_sz, [_i, _i, _i]),
    "fresco_fn_colstats": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _vp]),
    "fresco_fn_prep": (_i, [_vp] * 7 + [_i64, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "fresco_fn_layernorm": (_i, [_vp] * 7 + [_i64, _i64, _i64, _i, _f, _f, _vp, _vp]),
    "fresco_fn_conv7_rgb": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "fresco_fn_convex_upsample": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fresco_flow_occlusion": (_i, [_vp] * 5 + [_i, _i, _i, _i, _f, _f, _f, _vp]),
    "fresco_warp_fuse_chain": (_i, [_vp] * 8 + [_i, _i, _i, _i, _i, _vp]),
    "fresco_adain": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _i, _vp]),
    "fresco_chan_mean_std": (_i, [_vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "fresco_opt_workspace_bytes": (_sz, [_i] * 7),
    "fresco_opt_run": (_i, [_vp] * 7 + [_sz, _i, _i, _i, _i, _i, _f, _i, _f, _f, _f, _f, _vp]),
    "fresco_ctx_create": (_i, [_vp]),
    "fresco_ctx_destroy": (_i, [_vp]),
    "fresco_opt_run_ctx": (_i, [_vp] * 8 + [_sz, _i, _i, _i, _i, _i, _f, _i, _f, _f, _f, _f, _vp]),
    "fresco_opt_loss_grad": (_i, [_vp] * 9 + [_sz, _i, _i, _i, _i, _i, _f, _vp]),
    "fresco_opt_sharded_workspace_bytes": (_sz, [_i] * 7),
    "fresco_opt_sharded_begin": (_i, [_vp] * 5 + [_sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fresco_opt_sharded_step": (_i, [_vp] * 9 + [_sz, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _f, _f, _vp]),
    "fresco_opt_sharded_step_part": (_i, [_vp] * 9 + [_sz, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _f, _f, _i, _vp]),
    "fresco_mapping_workspace_bytes": (_sz, [_i, _i, _i]),
    "fresco_mapping_ind": (_i, [_vp] * 7 + [_sz, _i, _i, _i, _f, _vp]),
    "fresco_ddpm_x0": (_i, [_vp] * 5 + [_i64, _f, _f, _f, _i, _vp]),
    "fresco_ddpm_prev": (_i, [_vp] * 4 + [_i64, _i64, _f, _f, _f, _i, _vp]),
    "fresco_gram_target": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load the library (once) and return the ctypes handle; raises FrescoHipError if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FrescoHipError(
            "fresco_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fresco_amd/csrc`. There is no CPU / eager fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise FrescoHipError("fresco_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise FrescoHipError("fresco_amd: %s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != OK:
        detail = ""
        if rc == -4:
            detail = ": " + load().fresco_last_error().decode()
        raise FrescoHipError("%s failed: %s%s" % (what, ERRORS.get(rc, "error %d" % rc), detail))
