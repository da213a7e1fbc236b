"""ctypes binding of libcnerf_hip.so (include/cnerf.h).  There is NO fallback: if the library is
missing or a call fails, this raises — the product path never routes through CPU code."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CNERF_LIB_PATH: load an experiment build of the same ABI instead (kernel ablations, scripts/kvariants.sh)
LIB_PATH = os.environ.get("CNERF_LIB_PATH") or os.path.join(_HERE, "libcnerf_hip.so")
MAX_TENSORS = 48


class CnerfError(RuntimeError):
    pass


class Net(C.Structure):
    """struct cnerf_net"""
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("multires", C.c_int32), ("multires_views", C.c_int32),
                ("use_viewdirs", C.c_int32), ("output_ch", C.c_int32), ("skip", C.c_int32)]


class Ptrs(C.Structure):
    """struct cnerf_ptrs"""
    _fields_ = [("p", C.c_void_p * MAX_TENSORS)]


class RenderCfg(C.Structure):
    """struct cnerf_render_cfg"""
    _fields_ = [("Nc", C.c_int32), ("Nf", C.c_int32), ("lindisp", C.c_int32), ("white_bkgd", C.c_int32),
                ("ray_stride", C.c_int32), ("train", C.c_int32)]


class RayGen(C.Structure):
    """struct cnerf_raygen"""
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("c2w", C.c_float * 12), ("near", C.c_float), ("far", C.c_float), ("use_viewdirs", C.c_int32),
                ("ndc", C.c_int32), ("ndc_ax", C.c_float), ("ndc_ay", C.c_float), ("first", C.c_int64)]


class RenderOut(C.Structure):
    """struct cnerf_render_out"""
    _fields_ = [(n, C.c_void_p) for n in ("rgb_map", "disp_map", "acc_map", "depth_map", "rgb0", "disp0", "acc0",
                                          "depth0", "z_std", "raw", "z_vals", "weights")]


class RenderGrads(C.Structure):
    """struct cnerf_render_grads"""
    _fields_ = [(n, C.c_void_p) for n in ("g_rgb_map", "g_disp_map", "g_acc_map", "g_depth_map", "g_rgb0", "g_disp0",
                                          "g_acc0", "g_depth0")]


class Rng(C.Structure):
    """struct cnerf_rng"""
    _fields_ = [("seed", C.c_uint64), ("offset", C.c_uint64), ("state_dev", C.c_void_p), ("row0", C.c_int64)]


class Closs(C.Structure):
    """struct cnerf_closs"""
    _fields_ = [("target", C.c_void_p), ("mask", C.c_void_p), ("prior", C.c_void_p), ("far", C.c_float), ("seg_row", C.c_int64)]


class ClossTail(C.Structure):
    """struct cnerf_closs_sum"""
    _fields_ = [("ws_last", C.c_void_p), ("ws_coarse", C.c_void_p), ("B", C.c_int64), ("counts", C.c_void_p),
                ("coef", C.c_float), ("far", C.c_float), ("rgb_w", C.c_float), ("depth_w", C.c_float), ("patch_w", C.c_float),
                ("has_depth", C.c_int32), ("depth_last", C.c_void_p), ("depth_coarse", C.c_void_p), ("mono", C.c_void_p),
                ("P", C.c_int32), ("n", C.c_int32)]


class SsWarp(C.Structure):
    """struct cnerf_ss_warp"""
    _fields_ = [("ref", RayGen), ("w2c", C.c_float * 12), ("flip", C.c_int32), ("image_ch", C.c_int32), ("thr0", C.c_float)]


class PixelBatch(C.Structure):
    """struct cnerf_pixel_batch"""
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("c2w", C.c_float * 12), ("near", C.c_float), ("far", C.c_float), ("use_viewdirs", C.c_int32), ("ndc", C.c_int32),
                ("ndc_ax", C.c_float), ("ndc_ay", C.c_float), ("crop_r0", C.c_int32), ("crop_c0", C.c_int32),
                ("crop_h", C.c_int32), ("crop_w", C.c_int32), ("n_patches", C.c_int32), ("patch_size", C.c_int32),
                ("patch_start", (C.c_int32 * 2) * 16), ("n_rand", C.c_int64), ("image_ch", C.c_int32), ("n_extras", C.c_int32)]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_ClossP = C.POINTER(Closs)
_RngP = C.POINTER(Rng)
_NetP, _PtrsP = C.POINTER(Net), C.POINTER(Ptrs)

# name -> (restype, argtypes); every symbol include/cnerf.h declares
SIGNATURES = {
    "cnerf_abi_version": (_i, []),
    "cnerf_strerror": (C.c_char_p, [_i]),
    "cnerf_device_info": (_i, [_i, C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "cnerf_num_tensors": (_i, [_NetP]),
    "cnerf_tensor_shape": (_i, [_NetP, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "cnerf_packed_floats": (_i64, [_NetP]),
    "cnerf_pack_weights": (_i, [_NetP, _PtrsP, _vp, _vp]),
    "cnerf_pack_weights_pair": (_i, [_NetP, _PtrsP, _vp, _NetP, _PtrsP, _vp, _vp]),
    "cnerf_coarse_z": (_i, [_vp, _i, _i64, _i, _vp, _vp, _i, _vp, _vp]),
    "cnerf_uniform_rng": (_i, [_RngP, _i64, _i, _vp, _vp]),
    "cnerf_coarse_z_rng": (_i, [_vp, _i, _i64, _i, _vp, _RngP, _i, _vp, _vp]),
    "cnerf_resample_rng": (_i, [_vp, _vp, _RngP, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_composite_mse_ws_floats": (_i64, [_i64]),
    "cnerf_composite_mse_counter_words": (_i64, []),
    "cnerf_composite_mse_max_rays": (_i64, []),
    "cnerf_composite_fwd_mse": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_composite_bwd_mse": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_closs_ws_floats": (_i64, [_i64]),
    "cnerf_composite_fwd_closs": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _ClossP, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_closs_finish": (_i, [C.POINTER(ClossTail), _vp, _vp, _vp, _vp]),
    "cnerf_closs_finish_ss": (_i, [C.POINTER(ClossTail), C.POINTER(C.c_int32), _vp, _vp, _vp]),
    "cnerf_closs_finish_ss2": (_i, [C.POINTER(ClossTail), C.POINTER(C.c_int32), _i64, _vp, _vp, _vp, _vp]),
    "cnerf_composite_bwd_closs": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _ClossP, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _i64,
                                       _vp, _vp]),
    "cnerf_sample_pixels": (_i, [C.POINTER(PixelBatch), _vp, _RngP, _vp, C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_embed": (_i, [_vp, _i64, _i, _vp, _vp]),
    "cnerf_mlp_stash_floats": (_i64, [_NetP, _i64]),
    "cnerf_mlp_fwd": (_i, [_NetP, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cnerf_mlp_fwd_live": (_i, [_NetP, _vp, _vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "cnerf_mlp_fwd_embedded": (_i, [_NetP, _vp, _vp, _i64, _vp, _vp, _vp]),
    "cnerf_packed_bf_bytes": (_i64, [_NetP, _i]),
    "cnerf_pack_weights_bf": (_i, [_NetP, _PtrsP, _i, _vp, _vp]),
    "cnerf_mlp_fwd_bf": (_i, [_NetP, _vp, _i, _vp, _vp, _i, _vp, _vp, _i64, _i, _vp, _vp]),
    "cnerf_mlp_fwd_bf_train": (_i, [_NetP, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cnerf_mlp_dgrad_bf": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cnerf_mlp_dgrad_bf_pair": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _NetP, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cnerf_mlp_bwd_ws_floats": (_i64, [_NetP, _i64]),
    "cnerf_mlp_bwd": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _PtrsP, _i, _vp]),
    "cnerf_mlp_bwd_pair": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _PtrsP, _NetP, _vp, _vp, _i64, _i, _vp, _vp, _PtrsP, _i, _vp]),
    "cnerf_mlp_dgrad_pair_live": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _NetP, _vp, _vp, _i64, _i, _vp, _vp, _vp, _i64, _i64, _vp]),
    "cnerf_mlp_wgrad_pair_live": (_i, [_NetP, _i64, _i, _vp, _vp, _PtrsP, _NetP, _i64, _i, _vp, _vp, _PtrsP, _i, _vp, _i64, _i64, _vp]),
    "cnerf_mlp_bwd_live": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _PtrsP, _i, _vp, _vp]),
    "cnerf_mlp_bwd_pair_live": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _PtrsP, _NetP, _vp, _vp, _i64, _i, _vp, _vp, _PtrsP, _i, _vp,
                                     _i64, _i64, _vp]),
    "cnerf_mlp_dgrad_pair": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _NetP, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cnerf_mlp_wgrad_pair": (_i, [_NetP, _i64, _i, _vp, _vp, _PtrsP, _NetP, _i64, _i, _vp, _vp, _PtrsP, _i, _vp]),
    "cnerf_mlp_dgrad": (_i, [_NetP, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cnerf_mlp_wgrad": (_i, [_NetP, _i64, _i, _vp, _vp, _PtrsP, _i, _vp]),
    "cnerf_mlp_wgrad_bf": (_i, [_NetP, _i64, _i, _vp, _vp, _PtrsP, _i, _vp]),
    "cnerf_mlp_wgrad_bf_pair": (_i, [_NetP, _i64, _i, _vp, _vp, _PtrsP, _NetP, _i64, _i, _vp, _vp, _PtrsP, _i, _vp]),
    "cnerf_composite_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_composite_bwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_sample_pdf": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _i, _vp, _vp, _vp]),
    "cnerf_resample": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_render_ws_floats": (_i64, [_NetP, _NetP, C.POINTER(RenderCfg), _i64]),
    "cnerf_render_fwd": (_i, [_NetP, _vp, _NetP, _vp, _vp, _i64, C.POINTER(RenderCfg), _vp, _vp, _vp, _i64, _vp, _vp,
                              C.POINTER(RenderOut), _vp, _vp]),
    "cnerf_render_fwd_cam": (_i, [_NetP, _vp, _NetP, _vp, C.POINTER(RayGen), _i64, C.POINTER(RenderCfg), _vp, _vp, _vp, _i64, _vp,
                                  _vp, C.POINTER(RenderOut), _vp, _vp]),
    "cnerf_render_bwd": (_i, [_NetP, _vp, _NetP, _vp, _vp, _i64, C.POINTER(RenderCfg), _vp, _vp, C.POINTER(RenderGrads),
                              _vp, _PtrsP, _PtrsP, _i, _vp]),
    "cnerf_gen_rays": (_i, [_i, _i, _f, _f, _f, _f, C.POINTER(_f), _f, _f, _i, _i, _f, _f, _vp, _vp]),
    "cnerf_pack_rays": (_i, [_vp, _vp, _i64, _f, _f, _i, _i, _f, _f, _vp, _vp]),
    "cnerf_warp_points": (_i, [_vp, _i64, C.POINTER(_f), _f, _f, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_hard_mask_pair": (_i, [_i, _i, _f, _f, _f, _f, C.POINTER(_f), C.POINTER(_f), _vp, _vp, _f, _i, _vp,
                                  _vp, _vp]),
    "cnerf_ss_ref_rays": (_i, [C.POINTER(SsWarp), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_ss_batch": (_i, [C.POINTER(SsWarp), _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp]),
    "cnerf_mse": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "cnerf_mse_ws_floats": (_i64, [_i64]),
    "cnerf_soft_lp_loss": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp]),
    "cnerf_mse_ws": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "cnerf_loss_ws_floats": (_i64, []),
    "cnerf_masked_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "cnerf_patch_depth_loss": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "cnerf_adam_hyper": (_i, [_i, C.c_double, C.c_double, C.c_double, C.c_double, _f, _f, _vp]),
    "cnerf_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "cnerf_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _i, C.c_double, C.c_double, C.c_double, C.c_double, _f, _f, _vp]),
}

_lib = None


def load():
    """Load (once) and type the library.  torch is imported first so that our NEEDED libamdhip64.so.7
    resolves to the HIP runtime torch already mapped (one runtime per process: streams and device
    pointers are shared)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must precede the dlopen, see docstring)
    if not os.path.exists(LIB_PATH):
        raise CnerfError(
            f"{LIB_PATH} is missing: build it with `python -m consistentnerf_amd.build` "
            "(hipcc --offload-arch=gfx950).  consistentnerf_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CnerfError(f"libcnerf_hip.so does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.cnerf_abi_version() != 6:
        raise CnerfError("libcnerf_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cnerf_strerror(rc).decode()
        raise CnerfError(f"{what} failed: {msg} (code {rc})")
