"""Drop-in replacements for the reference's neural fields (models/fields.py): same constructor
keywords, parameter names / order (`linN.bias`, `linN.weight_g`, `linN.weight_v`, `variance`) and
initial values, so that `state_dict()`s and Adam states are interchangeable.  The arithmetic runs in
librnbneus_hip.so; these classes only own the parameters and marshal pointers.

  SDFNetwork            <- models/fields.py:8-127
  RenderingNetwork      <- models/fields.py:131-215   (mode "no_view_dir")
  SingleVarianceNetwork <- models/fields.py:317-325
  NeRF                  <- models/fields.py:219-314   (parameter container only: every shipped config
                           has n_outside = 0, so it is constructed and checkpointed but never evaluated)
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import native
from .embedder import get_embedder


class _WNLinear(nn.Module):
    """Parameter holder with the attribute layout nn.utils.weight_norm(nn.Linear) leaves behind:
    named_parameters() yields bias, weight_g, weight_v (in that order)."""

    def __init__(self, lin: nn.Linear, weight_norm: bool):
        super().__init__()
        self.in_features = lin.in_features
        self.out_features = lin.out_features
        w = lin.weight.detach().clone()
        if weight_norm:
            self.bias = nn.Parameter(lin.bias.detach().clone())
            self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
            self.bias = nn.Parameter(lin.bias.detach().clone())

    def leaves(self):
        if hasattr(self, "weight_g"):
            return [self.bias, self.weight_g, self.weight_v]
        return [self.weight, self.bias]


def _forward_only_guard(module: nn.Module, what: str, *inputs):
    """The direct network calls run the native forward sweeps and return tensors WITHOUT a grad_fn; the reference's are
    ordinary autograd modules (models/fields.py:82-127, :177-215; `gradient` even builds a graph, create_graph=True).  A loss
    written on them would train nothing without being told — so under grad mode with anything trainable in sight the call
    fails loudly instead of detaching silently."""
    if not torch.is_grad_enabled():
        return
    if any(p.requires_grad for p in module.parameters()) or any(torch.is_tensor(t) and t.requires_grad for t in inputs):
        raise RuntimeError(
            f"{what} is forward-only in rnb_neus_fork_amd: it returns a tensor without grad_fn, so gradients would silently "
            "not flow.  Training gradients flow through NeuSRenderer.render / render_rnb / render_rnb_warmup (whose backward is "
            "native); for evaluation wrap the call in torch.no_grad() (or freeze the parameters with requires_grad_(False)).")


def _mlp_struct(lins, weight_norm, grads=None):
    """rnb_mlp_params / rnb_mlp_grads for a list of _WNLinear (grads: dict leaf -> tensor)."""
    s = native.MlpParams()
    s.n_lin = len(lins)
    for i, lin in enumerate(lins):
        if weight_norm:
            g, v, b = lin.weight_g, lin.weight_v, lin.bias
        else:
            g, v, b = None, lin.weight, lin.bias
        if grads is not None:
            g = grads[id(g)] if g is not None else None
            v, b = grads[id(v)], grads[id(b)]
        for t in (g, v, b):
            if t is not None:
                assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda, "parameters must be fp32 on the GPU"
        s.g[i] = g.data_ptr() if g is not None else None
        s.v[i] = v.data_ptr()
        s.b[i] = b.data_ptr()
    return s


class SDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        if d_in != 3:
            raise ValueError("SDFNetwork: d_in must be 3")
        skip_in = tuple(skip_in)
        if len(skip_in) > 1:
            raise ValueError("SDFNetwork: at most one skip connection is supported")
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embed_fn_fine = None
        if multires > 0:
            embed_fn, input_ch = get_embedder(multires, input_dims=d_in)
            self.embed_fn_fine = embed_fn
            dims[0] = input_ch
        self.num_layers = len(dims)
        self.skip_in = skip_in
        self.scale = scale
        self.d_in, self.d_out, self.d_hidden, self.n_layers = d_in, d_out, d_hidden, n_layers
        self.multires = multires
        self.weight_norm = bool(weight_norm)
        # same construction order / RNG consumption as models/fields.py:40-74
        for l in range(0, self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                with torch.no_grad():
                    if l == self.num_layers - 2:
                        mean = np.sqrt(np.pi) / np.sqrt(dims[l])
                        if not inside_outside:
                            torch.nn.init.normal_(lin.weight, mean=mean, std=0.0001)
                            torch.nn.init.constant_(lin.bias, -bias)
                        else:
                            torch.nn.init.normal_(lin.weight, mean=-mean, std=0.0001)
                            torch.nn.init.constant_(lin.bias, bias)
                    elif multires > 0 and l == 0:
                        torch.nn.init.constant_(lin.bias, 0.0)
                        torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                        torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    elif multires > 0 and l in self.skip_in:
                        torch.nn.init.constant_(lin.bias, 0.0)
                        torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                        torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                    else:
                        torch.nn.init.constant_(lin.bias, 0.0)
                        torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, "lin" + str(l), _WNLinear(lin, self.weight_norm))

    # -- plumbing ------------------------------------------------------------------------------------
    def lins(self):
        return [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]

    def leaves(self):
        out = []
        for lin in self.lins():
            out += lin.leaves()
        return out

    def _standalone(self):
        from .runtime import StandaloneSDF
        return StandaloneSDF(self)

    # -- reference API (forward only, and loud about it; training gradients flow through NeuSRenderer.render*) -----------
    def forward(self, inputs):
        """[N,3] -> [N,d_out] = [sdf, feature]  (models/fields.py:82-104).  Forward only: raises under grad mode."""
        _forward_only_guard(self, "SDFNetwork.forward", inputs)
        ctx = self._standalone()
        return ctx.sdf_forward(inputs, with_feature=True)

    def sdf(self, x):
        """[N,3] -> [N,1]  (models/fields.py:106-108).  Forward only: raises under grad mode."""
        _forward_only_guard(self, "SDFNetwork.sdf", x)
        ctx = self._standalone()
        return ctx.sdf_forward(x, with_feature=False)

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        """[N,3] -> [N,1,3] = d sdf / d x  (models/fields.py:114-127), analytic reverse sweep.  Forward only (the
        reference returns a differentiable graph, create_graph=True): raises under grad mode."""
        _forward_only_guard(self, "SDFNetwork.gradient", x)
        ctx = self._standalone()
        return ctx.sdf_gradient(x).unsqueeze(1)


class RenderingNetwork(nn.Module):
    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True):
        super().__init__()
        if mode != "no_view_dir":
            raise NotImplementedError(
                f"RenderingNetwork mode '{mode}': only 'no_view_dir' (every shipped conf) is implemented")
        self.mode = mode
        self.squeeze_out = squeeze_out
        self.d_feature, self.d_in, self.d_out, self.d_hidden, self.n_layers = d_feature, d_in, d_out, d_hidden, n_layers
        self.multires_view = multires_view
        self.weight_norm = bool(weight_norm)
        dims = [d_in + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embedview_fn = None
        if multires_view > 0:
            embedview_fn, input_ch = get_embedder(multires_view)
            self.embedview_fn = embedview_fn
            dims[0] += 2 * (input_ch - 3)
        self.num_layers = len(dims)
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            setattr(self, "lin" + str(l), _WNLinear(lin, self.weight_norm))

    def lins(self):
        return [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]

    def leaves(self):
        out = []
        for lin in self.lins():
            out += lin.leaves()
        return out

    def forward(self, points, normals, view_dirs, feature_vectors):
        """models/fields.py:177-215 (view_dirs are encoded and discarded by the reference in this mode).  Forward only:
        raises under grad mode."""
        _forward_only_guard(self, "RenderingNetwork.forward", points, normals, feature_vectors)
        from .runtime import standalone_color
        return standalone_color(self, points, normals, feature_vectors)


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))

    def forward(self, x):
        """models/fields.py:323-325: ones([len(x),1]) * exp(10 * variance)."""
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)


class NeRF(nn.Module):
    """Background network of the reference (models/fields.py:219-314).  With n_outside = 0 (all shipped
    configs) it is never evaluated; this container keeps the reference's parameter names so that
    checkpoints and the optimizer's positional state line up (exp_runner.py:105, :361-379)."""

    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=[4],
                 use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch = d_in * (1 + 2 * multires) if multires > 0 else 3
        self.input_ch_view = d_in_view * (1 + 2 * multires_view) if multires_view > 0 else 3
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] +
            [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    def forward(self, input_pts, input_views):
        raise NotImplementedError("NeRF background (n_outside > 0) is outside the accelerated path")


def model_desc(sdf: SDFNetwork, color: RenderingNetwork, n_samples=64, n_importance=64, up_sample_steps=4):
    d = native.ModelDesc()
    d.sdf_d_in = sdf.d_in
    d.sdf_d_out = sdf.d_out
    d.sdf_d_hidden = sdf.d_hidden
    d.sdf_n_layers = sdf.n_layers
    d.sdf_skip_in = sdf.skip_in[0] if len(sdf.skip_in) else -1
    d.sdf_multires = sdf.multires
    d.sdf_scale = float(sdf.scale)
    d.sdf_weight_norm = int(sdf.weight_norm)
    if color is not None:
        d.col_d_feature = color.d_feature
        d.col_d_in = color.d_in
        d.col_d_out = color.d_out
        d.col_d_hidden = color.d_hidden
        d.col_n_layers = color.n_layers
        d.col_multires_view = color.multires_view
        d.col_squeeze_out = int(bool(color.squeeze_out))
        d.col_weight_norm = int(color.weight_norm)
    else:  # a structurally valid placeholder; the albedo rows of the packed buffer stay unused
        d.col_d_feature = sdf.d_out - 1
        d.col_d_in = 6
        d.col_d_out = 3
        d.col_d_hidden = 32
        d.col_n_layers = 1
        d.col_multires_view = 0
        d.col_squeeze_out = 1
        d.col_weight_norm = 1
    d.n_samples = n_samples
    d.n_importance = n_importance
    d.up_sample_steps = up_sample_steps
    return d
