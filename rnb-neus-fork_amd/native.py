"""ctypes binding of librnbneus_hip.so (include/rnbneus.h).  There is no CPU fallback: if the library
is missing or a symbol is absent, loading fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librnbneus_hip.so")
ABI_VERSION = 5
MAX_LIN = 16

MODE_CORE = 0
MODE_MVPS = 1
FLAG_RELU_SHADING = 2
FLAG_NO_ALBEDO = 4
FLAG_LIGHT_PER_RAY = 8
FLAG_FORWARD_ONLY = 16

# rnb_model_desc.variant bits (include/rnbneus.h)
VARIANT_BF16 = 1
VARIANT_DETERMINISTIC = 2
VARIANT_GENERIC = 4
VARIANT_DW_LDS = 8
VARIANT_DW_STAGED = 16
VARIANT_X3 = 32
VARIANT_F32_MFMA = 64
VARIANT_BWD_TI_SHIFT, VARIANT_BWD_NW_SHIFT, VARIANT_FWD_TI_SHIFT, VARIANT_FWD_NW_SHIFT = 8, 10, 12, 14
VARIANT_REG_TILE = 1 << 16
VARIANT_LDS_TILE = 1 << 17
VARIANT_X2H = 1 << 18
VARIANT_NO_X2H = 1 << 19

c_float_p = C.c_void_p  # device pointers are passed as integers


class ModelDesc(C.Structure):
    _fields_ = [
        ("sdf_d_in", C.c_int32), ("sdf_d_out", C.c_int32), ("sdf_d_hidden", C.c_int32),
        ("sdf_n_layers", C.c_int32), ("sdf_skip_in", C.c_int32), ("sdf_multires", C.c_int32),
        ("sdf_scale", C.c_float), ("sdf_weight_norm", C.c_int32),
        ("col_d_feature", C.c_int32), ("col_d_in", C.c_int32), ("col_d_out", C.c_int32),
        ("col_d_hidden", C.c_int32), ("col_n_layers", C.c_int32), ("col_multires_view", C.c_int32),
        ("col_squeeze_out", C.c_int32), ("col_weight_norm", C.c_int32),
        ("n_samples", C.c_int32), ("n_importance", C.c_int32), ("up_sample_steps", C.c_int32),
        ("variant", C.c_int32),
    ]


class MlpParams(C.Structure):
    _fields_ = [("n_lin", C.c_int32), ("pad_", C.c_int32),
                ("g", C.c_void_p * MAX_LIN), ("v", C.c_void_p * MAX_LIN), ("b", C.c_void_p * MAX_LIN)]


MlpGrads = MlpParams  # identical layout (non-const pointers)


class RenderArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int64), ("S", C.c_int32), ("n_lights", C.c_int32), ("flags", C.c_int32),
        ("cos_anneal_ratio", C.c_float),
        ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("z_vals", C.c_void_p), ("lights_dir", C.c_void_p),
        ("background_rgb", C.c_void_p), ("variance", C.c_void_p),
        ("color_fine", C.c_void_p), ("weights", C.c_void_p), ("cdf_fine", C.c_void_p),
        ("gradients", C.c_void_p), ("inside_sphere", C.c_void_p), ("weight_sum", C.c_void_p),
        ("weight_max", C.c_void_p), ("s_val", C.c_void_p), ("gradient_error", C.c_void_p),
        ("sdf", C.c_void_p), ("sampled_albedo", C.c_void_p),
        ("gerr_partial", C.c_void_p), ("gerr_den_global", C.c_void_p),
    ]


class GridDesc(C.Structure):
    _fields_ = [("bound_min", C.c_float * 3), ("bound_max", C.c_float * 3), ("resolution", C.c_int32),
                ("x_begin", C.c_int32), ("x_end", C.c_int32), ("out_scale", C.c_float)]


class RenderGrads(C.Structure):
    _fields_ = [("color_fine", C.c_void_p), ("weights", C.c_void_p), ("cdf_fine", C.c_void_p),
                ("gradients", C.c_void_p), ("weight_sum", C.c_void_p), ("weight_max", C.c_void_p),
                ("s_val", C.c_void_p), ("gradient_error", C.c_void_p)]


_P = C.POINTER
_SIGNATURES = {
    "rnb_abi_version": (C.c_int, []),
    "rnb_last_error_string": (C.c_char_p, []),
    "rnb_build_id": (C.c_char_p, []),
    "rnb_packed_floats": (C.c_int, [_P(ModelDesc), _P(C.c_int64)]),
    "rnb_weightnorm_fwd": (C.c_int, [_P(ModelDesc), _P(MlpParams), _P(MlpParams), C.c_void_p, C.c_void_p]),
    "rnb_weightnorm_bwd": (C.c_int, [_P(ModelDesc), _P(MlpParams), _P(MlpParams), C.c_void_p, _P(MlpGrads),
                                     _P(MlpGrads), C.c_void_p]),
    "rnb_points_workspace_bytes": (C.c_int, [_P(ModelDesc), C.c_int64, _P(C.c_int64)]),
    "rnb_sdf_forward": (C.c_int, [_P(ModelDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnb_sdf_gradient": (C.c_int, [_P(ModelDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnb_color_forward": (C.c_int, [_P(ModelDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnb_sdf_grid_workspace_bytes": (C.c_int, [_P(ModelDesc), _P(GridDesc), _P(C.c_int64)]),
    "rnb_sdf_grid": (C.c_int, [_P(ModelDesc), C.c_void_p, _P(GridDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_void_p]),
    "rnb_marching_cubes_workspace_bytes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P(C.c_int64)]),
    "rnb_marching_cubes_count": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_void_p]),
    "rnb_marching_cubes_emit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                          C.c_size_t, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rnb_up_sample_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "rnb_gather_sdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p]),
    "rnb_sample_workspace_bytes": (C.c_int, [_P(ModelDesc), C.c_int64, _P(C.c_int64)]),
    "rnb_sample_rays": (C.c_int, [_P(ModelDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnb_render_workspace_bytes": (C.c_int, [_P(ModelDesc), C.c_int64, C.c_int32, C.c_int32, _P(C.c_int64)]),
    "rnb_render_fwd": (C.c_int, [_P(ModelDesc), C.c_void_p, _P(RenderArgs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnb_render_bwd": (C.c_int, [_P(ModelDesc), C.c_void_p, _P(RenderArgs), _P(RenderGrads), C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnb_render_range": (C.c_int, [_P(ModelDesc), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p]),
    "rnb_algorithmic_flops": (C.c_int, [_P(ModelDesc), C.c_int64, C.c_int32, _P(C.c_double), _P(C.c_double)]),
    "rnb_algorithmic_bytes": (C.c_int, [_P(ModelDesc), C.c_int64, C.c_int32, _P(C.c_double)]),
    "rnb_gen_rays_at_view": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "rnb_loss_rnb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                               C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "rnb_loss_rnb_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                     C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rnb_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_double, C.c_int64, C.c_void_p]),
    "rnb_profile_enable": (C.c_int, [C.c_int]),
    "rnb_profile_collect": (C.c_int, [_P(C.c_double), _P(C.c_int64), _P(C.c_double)]),
    "rnb_profile_report": (C.c_int64, [C.c_char_p, C.c_int64]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib = None


class NativeError(RuntimeError):
    pass


def load():
    """Loads the shared library once and checks every symbol declared in include/rnbneus.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "There is no CPU fallback for the renderer.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    v = lib.rnb_abi_version()
    if v != ABI_VERSION:
        raise NativeError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def build_id() -> str:
    """rnb_build_id(): the hash of sources + flags this library was compiled from (buildid.source_build_id of that tree)."""
    return load().rnb_build_id().decode()


def check(rc: int):
    if rc != 0:
        msg = load().rnb_last_error_string()
        raise NativeError(f"librnbneus_hip error {rc}: {msg.decode() if msg else '?'}")


def variant_bits(bf16=False, deterministic=False, generic=False, dw_lds=False, dw_staged=False, bwd_ti=0, bwd_nw=0,
                 fwd_ti=0, fwd_nw=0, x3=False, f32_mfma=False, reg_tile=False, lds_tile=False, x2h=None) -> int:
    """rnb_model_desc.variant from keyword switches (tile heights: 0/1/2; waves: 0/4/8; x2h: None = the library's default
    (on with the x3 arithmetic), True / False force the fp16 three-term forward sweeps on / off)."""
    nw = {0: 0, 4: 1, 8: 2}
    return ((VARIANT_BF16 if bf16 else 0) | (VARIANT_DETERMINISTIC if deterministic else 0)
            | (VARIANT_GENERIC if generic else 0) | (VARIANT_DW_LDS if dw_lds else 0)
            | (VARIANT_DW_STAGED if dw_staged else 0) | (VARIANT_X3 if x3 else 0) | (VARIANT_F32_MFMA if f32_mfma else 0)
            | (VARIANT_REG_TILE if reg_tile else 0) | (VARIANT_LDS_TILE if lds_tile else 0) | (0 if x2h is None else VARIANT_X2H if x2h else VARIANT_NO_X2H)
            | (int(bwd_ti) << VARIANT_BWD_TI_SHIFT) | (nw[int(bwd_nw)] << VARIANT_BWD_NW_SHIFT)
            | (int(fwd_ti) << VARIANT_FWD_TI_SHIFT) | (nw[int(fwd_nw)] << VARIANT_FWD_NW_SHIFT))


class on_device:
    """Context for one native call: makes the tensors' device the current HIP device (the library launches on
    the stream it is given and never calls hipSetDevice) and yields that device's current torch stream as the
    `rnb_stream_t` argument.  Without it a model on cuda:k with another current device would have its kernels
    enqueued on the wrong device's stream."""

    def __init__(self, device):
        import torch
        if isinstance(device, torch.Tensor):
            device = device.device
        if device.type != "cuda":
            raise RuntimeError("librnbneus_hip.so works on GPU tensors only (there is no CPU path)")
        self._torch = torch
        self.device = device
        self._ctx = torch.cuda.device(device)

    def __enter__(self):
        self._ctx.__enter__()
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def same_device(*tensors):
    """Raises unless every given tensor (None skipped) lives on one GPU; returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("librnbneus_hip.so works on GPU tensors only (there is no CPU path)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors of one native call live on different devices ({dev} and {t.device})")
    return dev


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensors must be contiguous fp32/int32."""
    if t is None:
        return None
    assert t.is_contiguous(), "native call needs contiguous tensors"
    return C.c_void_p(t.data_ptr())
