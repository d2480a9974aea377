"""Thin host runtime over the C ABI: weight packing, workspaces, point-wise network calls.
Everything here is plumbing (torch owns the memory and the stream); all arithmetic is native."""
from __future__ import annotations

import ctypes as C

import torch

from . import native
from .fields import RenderingNetwork, SDFNetwork, _mlp_struct, model_desc


def _require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the renderer has no CPU path "
                           "(librnbneus_hip.so is the only implementation)")


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def packed_floats(desc) -> int:
    n = C.c_int64()
    native.check(native.load().rnb_packed_floats(C.byref(desc), C.byref(n)))
    return n.value


def pack_weights(desc, sdf: SDFNetwork | None, color: RenderingNetwork | None, device) -> torch.Tensor:
    """rnb_weightnorm_fwd: effective weights of both MLPs in the library's packed layout."""
    lib = native.load()
    packed = torch.empty(packed_floats(desc), dtype=torch.float32, device=device)
    sp = _mlp_struct(sdf.lins(), sdf.weight_norm) if sdf is not None else None
    cp = _mlp_struct(color.lins(), color.weight_norm) if color is not None else None
    with native.on_device(packed) as stream:
        native.check(lib.rnb_weightnorm_fwd(C.byref(desc), C.byref(sp) if sp is not None else None,
                                            C.byref(cp) if cp is not None else None, native.ptr(packed), stream))
    return packed


def points_workspace(desc, n, device):
    b = C.c_int64()
    native.check(native.load().rnb_points_workspace_bytes(C.byref(desc), n, C.byref(b)))
    return torch.empty(b.value, dtype=torch.uint8, device=device)


def sdf_forward(desc, packed, pts, with_feature):
    _require_cuda(pts, "points")
    pts = _f32c(pts).reshape(-1, 3)
    n = pts.shape[0]
    sdf = torch.empty(n, 1, dtype=torch.float32, device=pts.device)
    feat = torch.empty(n, desc.sdf_d_out - 1, dtype=torch.float32, device=pts.device) if with_feature else None
    if n > 0:
        native.same_device(packed, pts)
        ws = points_workspace(desc, n, pts.device)
        with native.on_device(pts) as stream:
            native.check(native.load().rnb_sdf_forward(C.byref(desc), native.ptr(packed), native.ptr(pts), n,
                                                       native.ptr(sdf), native.ptr(feat), native.ptr(ws), ws.numel(),
                                                       stream))
    return torch.cat([sdf, feat], dim=-1) if with_feature else sdf


def sdf_gradient(desc, packed, pts):
    _require_cuda(pts, "points")
    pts = _f32c(pts).reshape(-1, 3)
    n = pts.shape[0]
    grad = torch.empty(n, 3, dtype=torch.float32, device=pts.device)
    if n > 0:
        native.same_device(packed, pts)
        ws = points_workspace(desc, n, pts.device)
        with native.on_device(pts) as stream:
            native.check(native.load().rnb_sdf_gradient(C.byref(desc), native.ptr(packed), native.ptr(pts), n,
                                                        native.ptr(grad), None, native.ptr(ws), ws.numel(), stream))
    return grad


def color_forward(desc, packed, pts, normals, feats):
    _require_cuda(pts, "points")
    pts = _f32c(pts).reshape(-1, 3)
    normals = _f32c(normals).reshape(-1, 3)
    feats = _f32c(feats).reshape(pts.shape[0], -1)
    n = pts.shape[0]
    out = torch.empty(n, desc.col_d_out, dtype=torch.float32, device=pts.device)
    if n > 0:
        native.same_device(packed, pts, normals, feats)
        ws = points_workspace(desc, n, pts.device)
        with native.on_device(pts) as stream:
            native.check(native.load().rnb_color_forward(C.byref(desc), native.ptr(packed), native.ptr(pts),
                                                         native.ptr(normals), native.ptr(feats), n, native.ptr(out),
                                                         native.ptr(ws), ws.numel(), stream))
    return out


class StandaloneSDF:
    """Context for calling an SDFNetwork outside a renderer (exp_runner.py:607-610 style)."""

    def __init__(self, sdf: SDFNetwork):
        self.sdf = sdf
        self.desc = model_desc(sdf, None)

    def _packed(self):
        dev = self.sdf.lin0.bias.device
        _require_cuda(self.sdf.lin0.bias, "SDFNetwork parameters")
        return pack_weights(self.desc, self.sdf, None, dev)

    def sdf_forward(self, x, with_feature):
        return sdf_forward(self.desc, self._packed(), x, with_feature)

    def sdf_gradient(self, x):
        return sdf_gradient(self.desc, self._packed(), x)


def standalone_color(color: RenderingNetwork, pts, normals, feats):
    """RenderingNetwork.forward without a renderer: a minimal placeholder SDF shape completes the
    descriptor (only the albedo rows of the packed buffer are written and read)."""
    d = native.ModelDesc()
    d.sdf_d_in, d.sdf_d_out, d.sdf_d_hidden, d.sdf_n_layers = 3, color.d_feature + 1, 32, 1
    d.sdf_skip_in, d.sdf_multires, d.sdf_scale, d.sdf_weight_norm = -1, 0, 1.0, 1
    d.col_d_feature, d.col_d_in, d.col_d_out = color.d_feature, color.d_in, color.d_out
    d.col_d_hidden, d.col_n_layers, d.col_multires_view = color.d_hidden, color.n_layers, color.multires_view
    d.col_squeeze_out, d.col_weight_norm = int(bool(color.squeeze_out)), int(color.weight_norm)
    d.n_samples, d.n_importance, d.up_sample_steps = 64, 0, 1
    _require_cuda(color.lin0.bias, "RenderingNetwork parameters")
    packed = pack_weights(d, None, color, color.lin0.bias.device)
    return color_forward(d, packed, pts, normals, feats)
