"""rnb-neus-fork_amd: MI355X-native (gfx950) volumetric SDF renderer for RNb-NeuS.

Drop-in for the reference's `models` package on the renderer hot path:

    from rnb_neus_fork_amd.fields import RenderingNetwork, SDFNetwork, SingleVarianceNetwork, NeRF
    from rnb_neus_fork_amd.renderer import NeuSRenderer

(exp_runner.py:13-15 are the only lines of the reference's runner that change.)  All arithmetic runs in
`librnbneus_hip.so` (hand-written HIP, C ABI in include/rnbneus.h); there is no CPU fallback.
"""
from __future__ import annotations

import torch

from . import native  # noqa: F401
from .embedder import get_embedder  # noqa: F401
from .fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork, model_desc  # noqa: F401
from .renderer import NeuSRenderer  # noqa: F401
from .losses import rnb_loss  # noqa: F401
from .optim import FlatAdam  # noqa: F401
from .raygen import DeviceRays  # noqa: F401
from .mcubes import marching_cubes  # noqa: F401

__all__ = ["NeuSRenderer", "SDFNetwork", "RenderingNetwork", "SingleVarianceNetwork", "NeRF", "get_embedder",
           "native", "build_from_named_params", "rnb_loss", "FlatAdam", "DeviceRays", "marching_cubes"]


def build_from_named_params(mc, params, device):
    """Builds (sdf_network, deviation_network, color_network, renderer) from a configuration object with
    `.sdf`, `.color`, `.render`, `.init_val` attribute groups (the constructor keywords of
    confs/wmask_rnb.conf:53-90) and loads a flat `{ 'sdf.lin0.weight_v': tensor, ... }` dict
    (reference state_dict names prefixed by sdf./color./dev.)."""
    s, c, r = mc.sdf, mc.color, mc.render
    sdf = SDFNetwork(d_in=s.d_in, d_out=s.d_out, d_hidden=s.d_hidden, n_layers=s.n_layers, skip_in=tuple(s.skip_in),
                     multires=s.multires, bias=s.bias, scale=s.scale, geometric_init=s.geometric_init,
                     weight_norm=s.weight_norm)
    dev = SingleVarianceNetwork(mc.init_val)
    col = RenderingNetwork(d_feature=c.d_feature, mode=c.mode, d_in=c.d_in, d_out=c.d_out, d_hidden=c.d_hidden,
                           n_layers=c.n_layers, weight_norm=c.weight_norm, multires_view=c.multires_view,
                           squeeze_out=c.squeeze_out)
    if params is not None:
        sdf.load_state_dict({k[4:]: v for k, v in params.items() if k.startswith("sdf.")})
        col.load_state_dict({k[6:]: v for k, v in params.items() if k.startswith("color.")})
        dev.load_state_dict({"variance": params["dev.variance"]})
    sdf, dev, col = sdf.to(device), dev.to(device), col.to(device)
    ren = NeuSRenderer(None, sdf, dev, col, n_samples=r.n_samples, n_importance=r.n_importance,
                       n_outside=r.n_outside, up_sample_steps=r.up_sample_steps, perturb=r.perturb)
    return sdf, dev, col, ren


build_from_oracle_params = build_from_named_params
