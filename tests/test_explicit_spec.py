"""The explicit forward/backward statement (oracle/explicit.py, the kernel specification) must agree
with autograd through the pinned oracle.  float64 so that agreement is ~1e-9, CPU only."""
import pytest
import torch

from oracle import rnb_oracle as O
from oracle.explicit import FinePass
from tests.golden_util import Golden


def _to64(d):
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}


@pytest.mark.parametrize("name,scale", [("tiny_warmup_sharp", 1.0), ("tiny_main_sharp", 1.0),
                                        ("tiny_main_noalbedo", 1.0), ("tiny_render_bg", 1.0),
                                        ("tiny_main_sharp", 1.7)])
def test_explicit_matches_autograd(name, scale):
    g = Golden(name)
    g.mc.sdf.scale = scale
    p = _to64(g.params())
    for v in p.values():
        v.requires_grad_(True)
    b = _to64(g.batch)
    z_vals = g.steps[-1]["z_out"].double()
    torch.manual_seed(0)
    if g.api == "render":
        bg = b.get("background_rgb")
        out = O.render(p, g.mc, b["rays_o"], b["rays_d"], b["near"], b["far"], z_vals=z_vals,
                       background_rgb=bg, cos_anneal_ratio=g.cos_anneal_ratio)
        fp = FinePass({k: v.detach() for k, v in p.items()}, g.mc)
        mine = fp.forward(b["rays_o"], b["rays_d"], z_vals, None, cos_anneal_ratio=g.cos_anneal_ratio,
                          relu_shading=False, no_albedo=False, mvps=False, background_rgb=bg)
    else:
        out = O.render_rnb(p, g.mc, b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"],
                           z_vals=z_vals, cos_anneal_ratio=g.cos_anneal_ratio, no_albedo=g.no_albedo,
                           warmup=(g.api == "render_rnb_warmup"))
        fp = FinePass({k: v.detach() for k, v in p.items()}, g.mc)
        mine = fp.forward(b["rays_o"], b["rays_d"], z_vals, b["lights_dir"],
                          cos_anneal_ratio=g.cos_anneal_ratio, relu_shading=(g.api == "render_rnb_warmup"),
                          no_albedo=g.no_albedo, mvps=True)
    keys = ["color_fine", "weights", "weight_sum", "weight_max", "gradients", "gradient_error", "cdf_fine",
            "s_val", "inside_sphere"]
    for k in keys:
        torch.testing.assert_close(mine[k], out[k].detach().double(), rtol=1e-9, atol=1e-10, msg=lambda m: f"{k}: {m}")
    # a loss touching every differentiable output with random cotangents
    gen = torch.Generator().manual_seed(5)
    cot = {k: torch.randn(out[k].shape, generator=gen, dtype=torch.float64)
           for k in keys if k != "inside_sphere"}
    loss = sum((out[k] * cot[k]).sum() for k in cot)
    loss.backward()
    grads = fp.backward(cot)
    for k, v in p.items():
        if v.grad is None:
            assert k not in grads or float(grads[k].abs().max()) == 0.0, k
            continue
        ref = v.grad
        got = grads[k].reshape(ref.shape)
        err = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
        assert err < 1e-8, f"{k}: rel {err:.3e}"
