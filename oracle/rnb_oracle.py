"""CPU oracle for the RNb-NeuS volumetric SDF renderer hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain PyTorch (CPU, fp32, autograd)
restatement of the reference algorithm.  It may be imported by `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` and by
nothing else; the product path (`rnb-neus-fork_amd/`) never routes through it.

Parity status: PINNED.  Every function here is checked in
`tests/test_oracle_golden.py` against golden vectors that
`oracle/gen_golden.py` produced by importing the reference's own
`models/{embedder,fields,renderer}.py` in the build container (the reference
ships no tests or known-answer vectors of its own).

Each function cites the reference lines it restates (paths relative to the
reference repository root).  Parameters are carried as a flat dict
``{name: tensor}`` using the reference's ``state_dict`` names prefixed by
``sdf.``, ``color.`` and ``dev.`` so fixtures are plain named arrays.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# configuration (values of confs/wmask_rnb.conf:54-89 are the defaults)
# --------------------------------------------------------------------------------------
@dataclass
class SDFConf:
    d_in: int = 3
    d_out: int = 257
    d_hidden: int = 256
    n_layers: int = 8
    skip_in: Sequence[int] = (4,)
    multires: int = 6
    bias: float = 0.5
    scale: float = 1.0
    geometric_init: bool = True
    weight_norm: bool = True
    inside_outside: bool = False

    def dims(self):
        d0 = self.d_in * (1 + 2 * self.multires) if self.multires > 0 else self.d_in
        return [d0] + [self.d_hidden] * self.n_layers + [self.d_out]

    def layer_shapes(self):
        """(out, in) of every linear layer — models/fields.py:24, :42-49."""
        dims = self.dims()
        shapes = []
        for l in range(len(dims) - 1):
            out = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            shapes.append((out, dims[l]))
        return shapes


@dataclass
class ColorConf:
    d_feature: int = 256
    mode: str = "no_view_dir"
    d_in: int = 6
    d_out: int = 3
    d_hidden: int = 256
    n_layers: int = 2
    weight_norm: bool = True
    multires_view: int = 4
    squeeze_out: bool = True

    def dims(self):
        d0 = self.d_in + self.d_feature
        if self.multires_view > 0:
            pe = 3 * (1 + 2 * self.multires_view)
            if self.mode == "no_view_dir":
                d0 += 2 * (pe - 3)
        return [d0] + [self.d_hidden] * self.n_layers + [self.d_out]

    def layer_shapes(self):
        dims = self.dims()
        return [(dims[l + 1], dims[l]) for l in range(len(dims) - 1)]


@dataclass
class RenderConf:
    n_samples: int = 64
    n_importance: int = 64
    n_outside: int = 0
    up_sample_steps: int = 4
    perturb: float = 1.0


@dataclass
class ModelConf:
    sdf: SDFConf = field(default_factory=SDFConf)
    color: ColorConf = field(default_factory=ColorConf)
    render: RenderConf = field(default_factory=RenderConf)
    init_val: float = 0.3


# --------------------------------------------------------------------------------------
# R1  positional encoding — models/embedder.py:34 (freqs), :40-46 (order), :53-55
# --------------------------------------------------------------------------------------
def embed(x: torch.Tensor, multires: int) -> torch.Tensor:
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^{L-1} x), cos(2^{L-1} x)]."""
    if multires <= 0:
        return x
    parts = [x]
    freqs = 2.0 ** torch.linspace(0.0, multires - 1, multires)
    for f in freqs:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, dim=-1)


# --------------------------------------------------------------------------------------
# weight norm — torch.nn.utils.weight_norm(dim=0) as applied at models/fields.py:72-74,:168-170
# --------------------------------------------------------------------------------------
def effective_weight(p: Params, prefix: str) -> torch.Tensor:
    if prefix + ".weight" in p:
        return p[prefix + ".weight"]
    g = p[prefix + ".weight_g"]
    v = p[prefix + ".weight_v"]
    # w = g * v / ||v||_row; torch's own fused primitive keeps the oracle bit-identical to
    # nn.utils.weight_norm's forward (the summation order of the row norm matters at 1 ulp,
    # which the up-sampling loop amplifies into different sample indices).
    return torch._weight_norm(v, g, 0)


def softplus100(x: torch.Tensor) -> torch.Tensor:
    """nn.Softplus(beta=100) — models/fields.py:80 (PyTorch threshold 20)."""
    return F.softplus(x, beta=100.0, threshold=20.0)


# --------------------------------------------------------------------------------------
# R2  SDF network — models/fields.py:82-104 (forward), :106-108 (sdf), :114-127 (gradient)
# --------------------------------------------------------------------------------------
def sdf_forward(p: Params, conf: SDFConf, pts: torch.Tensor) -> torch.Tensor:
    inputs = pts * conf.scale
    inputs = embed(inputs, conf.multires)
    x = inputs
    n_lin = conf.n_layers + 1
    for l in range(n_lin):
        if l in conf.skip_in:
            x = torch.cat([x, inputs], dim=1) / math.sqrt(2.0)
        w = effective_weight(p, f"sdf.lin{l}")
        x = F.linear(x, w, p[f"sdf.lin{l}.bias"])
        if l < n_lin - 1:
            x = softplus100(x)
    return torch.cat([x[:, :1] / conf.scale, x[:, 1:]], dim=-1)


def sdf_only(p: Params, conf: SDFConf, pts: torch.Tensor) -> torch.Tensor:
    return sdf_forward(p, conf, pts)[:, :1]


def sdf_gradient(p: Params, conf: SDFConf, pts: torch.Tensor, create_graph: bool = True) -> torch.Tensor:
    """d sdf / d x via autograd, as the reference does (second forward + grad)."""
    x = pts.detach().requires_grad_(True)
    with torch.enable_grad():
        y = sdf_only(p, conf, x)
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph, retain_graph=True)
    return g


# --------------------------------------------------------------------------------------
# R3  albedo network — models/fields.py:177-215
# --------------------------------------------------------------------------------------
def color_forward(p: Params, conf: ColorConf, points, normals, view_dirs, feats) -> torch.Tensor:
    if conf.multires_view > 0:
        points = embed(points, conf.multires_view)
        normals = embed(normals, conf.multires_view)
        view_dirs = embed(view_dirs, conf.multires_view)
    if conf.mode == "no_view_dir":
        x = torch.cat([points, normals, feats], dim=-1)
    elif conf.mode == "idr":
        x = torch.cat([points, view_dirs, normals, feats], dim=-1)
    elif conf.mode == "no_normal":
        x = torch.cat([points, view_dirs, feats], dim=-1)
    else:
        raise ValueError(conf.mode)
    n_lin = conf.n_layers + 1
    for l in range(n_lin):
        w = effective_weight(p, f"color.lin{l}")
        x = F.linear(x, w, p[f"color.lin{l}.bias"])
        if l < n_lin - 1:
            x = torch.relu(x)
    if conf.squeeze_out:
        x = torch.sigmoid(x)
    return x


# --------------------------------------------------------------------------------------
# R4  variance scalar — models/fields.py:323-325, clip at models/renderer.py:503
# --------------------------------------------------------------------------------------
def inv_s_of(p: Params) -> torch.Tensor:
    return torch.exp(p["dev.variance"] * 10.0).clip(1e-6, 1e6)


# --------------------------------------------------------------------------------------
# R7  importance sampling — models/renderer.py:39-69, :132-176, :178-192
# --------------------------------------------------------------------------------------
def sample_pdf_det(bins, weights, n_new, trace: Optional[dict] = None):
    """Deterministic inverse-CDF sampling (det=True branch), renderer.py:39-69."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, steps=n_new)
    u = u.expand(list(cdf.shape[:-1]) + [n_new]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_lo = torch.gather(cdf, 1, below)
    cdf_hi = torch.gather(cdf, 1, above)
    bin_lo = torch.gather(bins, 1, below)
    bin_hi = torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    if trace is not None:
        trace["inds"] = inds
    return bin_lo + t * (bin_hi - bin_lo)


def up_sample(rays_o, rays_d, z_vals, sdf, n_new, inv_s, trace: Optional[dict] = None):
    """renderer.py:132-176 — section-wise alpha at a fixed inv_s, then sample_pdf."""
    B, n = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
    radius = torch.linalg.norm(pts, ord=2, dim=-1)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    sdf = sdf.reshape(B, n)
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z_vals[:, :-1], z_vals[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
    prev_cos = torch.cat([torch.zeros([B, 1]), cos_val[:, :-1]], dim=-1)
    cos_val = torch.minimum(prev_cos, cos_val)
    cos_val = cos_val.clip(-1e3, 0.0) * inside
    dist = next_z - prev_z
    prev_esti = mid_sdf - cos_val * dist * 0.5
    next_esti = mid_sdf + cos_val * dist * 0.5
    prev_cdf = torch.sigmoid(prev_esti * inv_s)
    next_cdf = torch.sigmoid(next_esti * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
    trans = torch.cumprod(torch.cat([torch.ones([B, 1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    weights = alpha * trans
    return sample_pdf_det(z_vals, weights, n_new, trace).detach()


def cat_z_vals(p, conf: SDFConf, rays_o, rays_d, z_vals, new_z, sdf, last, trace: Optional[dict] = None):
    """renderer.py:178-192 — merge new depths (torch.sort) and carry the SDF along."""
    B, n = z_vals.shape
    n_new = new_z.shape[1]
    pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., :, None]
    z_cat = torch.cat([z_vals, new_z], dim=-1)
    z_sorted, index = torch.sort(z_cat, dim=-1)
    if trace is not None:
        trace["sort_index"] = index
    if not last:
        new_sdf = sdf_only(p, conf, pts.reshape(-1, 3)).reshape(B, n_new)
        sdf = torch.gather(torch.cat([sdf, new_sdf], dim=-1), 1, index)
    return z_sorted, sdf


def sample_rays(p: Params, mc: ModelConf, rays_o, rays_d, near, far, t_rand, perturb: float,
                trace: Optional[dict] = None):
    """Common prologue of render / render_rnb / render_rnb_warmup — renderer.py:557-608.

    `t_rand` is the [B,1] uniform draw (`torch.rand([B,1])`, renderer.py:572); the
    caller supplies it so that the oracle and the device path consume identical
    randomness.  Returns z_vals [B, n_samples + n_importance].
    """
    rc = mc.render
    B = rays_o.shape[0]
    z = torch.linspace(0.0, 1.0, rc.n_samples)
    z = near + (far - near) * z[None, :]
    if perturb > 0:
        z = z + (t_rand - 0.5) * 2.0 / rc.n_samples
    if rc.n_importance > 0:
        with torch.no_grad():
            pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., :, None]
            sdf = sdf_only(p, mc.sdf, pts.reshape(-1, 3)).reshape(B, rc.n_samples)
            if trace is not None:
                trace["coarse_sdf"] = sdf.clone()
                trace["steps"] = []
            for i in range(rc.up_sample_steps):
                st = {} if trace is not None else None
                new_z = up_sample(rays_o, rays_d, z, sdf, rc.n_importance // rc.up_sample_steps,
                                  64 * 2 ** i, st)
                if st is not None:
                    st["z_in"] = z.clone()
                    st["sdf_in"] = sdf.clone()
                    st["new_z"] = new_z.clone()
                z, sdf = cat_z_vals(p, mc.sdf, rays_o, rays_d, z, new_z, sdf,
                                    last=(i + 1 == rc.up_sample_steps), trace=st)
                if st is not None:
                    st["z_out"] = z.clone()
                    st["sdf_out"] = sdf.clone()
                    trace["steps"].append(st)
    return z


# --------------------------------------------------------------------------------------
# R5 / R6  render cores — models/renderer.py:194-285 and :466-554
# --------------------------------------------------------------------------------------
def _core_common(p: Params, mc: ModelConf, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio):
    B, S = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], sample_dist)], -1)
    mid_z = z_vals + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(B, S, 3).reshape(-1, 3)

    out = sdf_forward(p, mc.sdf, pts)
    sdf = out[:, :1]
    feat = out[:, 1:]
    grads = sdf_gradient(p, mc.sdf, pts, create_graph=True)
    color = color_forward(p, mc.color, pts, grads, dirs, feat).reshape(B, S, mc.color.d_out)

    inv_s = inv_s_of(p).reshape(1, 1).expand(B * S, 1)
    true_cos = (dirs * grads).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio)
                 + F.relu(-true_cos) * cos_anneal_ratio)
    est_next = sdf + iter_cos * dists.reshape(-1, 1) * 0.5
    est_prev = sdf - iter_cos * dists.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    pp = prev_cdf - next_cdf
    cc = prev_cdf
    alpha = ((pp + 1e-5) / (cc + 1e-5)).reshape(B, S).clip(0.0, 1.0)

    pts_norm = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).reshape(B, S)
    inside = (pts_norm < 1.0).float().detach()
    relax = (pts_norm < 1.2).float().detach()

    trans = torch.cumprod(torch.cat([torch.ones([B, 1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    weights = alpha * trans
    g3 = grads.reshape(B, S, 3)
    gerr = (torch.linalg.norm(g3, ord=2, dim=-1) - 1.0) ** 2
    gerr = (relax * gerr).sum() / (relax.sum() + 1e-5)
    return {
        "sdf": sdf, "dists": dists, "gradients": g3, "s_val": 1.0 / inv_s, "mid_z_vals": mid_z,
        "weights": weights, "cdf": cc.reshape(B, S), "gradient_error": gerr,
        "inside_sphere": inside, "sampled": color,
    }


def render_core(p, mc, rays_o, rays_d, z_vals, sample_dist, background_rgb=None, cos_anneal_ratio=0.0):
    """renderer.py:194-285 (n_outside == 0 path)."""
    r = _core_common(p, mc, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio)
    sampled_color = r.pop("sampled")[:, :, :3]
    w = r["weights"]
    color = (sampled_color * w[:, :, None]).sum(dim=1)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - w.sum(dim=-1, keepdim=True))
    r["color"] = color
    return r


def render_core_mvps(p, mc, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio=0.0):
    """renderer.py:466-554."""
    r = _core_common(p, mc, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio)
    r["sampled_albedo"] = r.pop("sampled")
    r["sampled_normal"] = r["gradients"]
    return r


# --------------------------------------------------------------------------------------
# R8  wrappers — models/renderer.py:556-648, :828-930, :932-1033
# --------------------------------------------------------------------------------------
def _pack(ret, color_fine, B, S):
    w = ret["weights"]
    return {
        "color_fine": color_fine,
        "s_val": ret["s_val"].reshape(B, S).mean(dim=-1, keepdim=True),
        "cdf_fine": ret["cdf"],
        "weight_sum": w.sum(dim=-1, keepdim=True),
        "weight_max": torch.max(w, dim=-1, keepdim=True)[0],
        "gradients": ret["gradients"],
        "weights": w,
        "gradient_error": ret["gradient_error"],
        "inside_sphere": ret["inside_sphere"],
    }


def render(p, mc: ModelConf, rays_o, rays_d, near, far, t_rand=None, perturb_overwrite=-1,
           background_rgb=None, cos_anneal_ratio=0.0, trace=None, z_vals=None):
    perturb = mc.render.perturb if perturb_overwrite < 0 else perturb_overwrite
    if z_vals is None:
        z_vals = sample_rays(p, mc, rays_o, rays_d, near, far, t_rand, perturb, trace)
    B, S = z_vals.shape
    ret = render_core(p, mc, rays_o, rays_d, z_vals, 2.0 / mc.render.n_samples,
                      background_rgb=background_rgb, cos_anneal_ratio=cos_anneal_ratio)
    out = _pack(ret, ret["color"], B, S)
    out["z_vals"] = z_vals
    return out


def render_rnb(p, mc: ModelConf, rays_o, rays_d, near, far, lights_dir, t_rand=None, perturb_overwrite=-1,
               cos_anneal_ratio=0.0, no_albedo=False, warmup=False, trace=None, z_vals=None):
    """render_rnb (warmup=False, renderer.py:932-1033) / render_rnb_warmup (True, :828-930)."""
    perturb = mc.render.perturb if perturb_overwrite < 0 else perturb_overwrite
    if z_vals is None:
        z_vals = sample_rays(p, mc, rays_o, rays_d, near, far, t_rand, perturb, trace)
    B, S = z_vals.shape
    ret = render_core_mvps(p, mc, rays_o, rays_d, z_vals, 2.0 / mc.render.n_samples,
                           cos_anneal_ratio=cos_anneal_ratio)
    albedo = ret["sampled_albedo"]
    if no_albedo:
        albedo = torch.ones_like(albedo)
    normal = ret["sampled_normal"]
    w = ret["weights"]
    directions = lights_dir * torch.ones(lights_dir.shape[0], B, S, 3)
    shading = (normal[None] * directions).sum(dim=-1, keepdim=True)
    if warmup:
        shading = torch.relu(shading)
    color_fine = (albedo[None] * w[None, :, :, None] * shading).sum(dim=2)
    out = _pack(ret, color_fine, B, S)
    out["z_vals"] = z_vals
    out["sampled_albedo"] = ret["sampled_albedo"]
    out["sdf"] = ret["sdf"]
    return out


# --------------------------------------------------------------------------------------
# R9  the train_rnb loss — exp_runner.py:187-194, :241-256 (weights confs/wmask_rnb.conf:37-38)
# --------------------------------------------------------------------------------------
def rnb_loss(render_out, true_rgb, mask, igr_weight=0.1, mask_weight=0.1):
    n_lights = true_rgb.shape[0]
    if mask_weight > 0.0:
        mask = (mask > 0.5).to(render_out["weight_sum"].dtype)   # `.float()` in the reference (fp32 run)
    else:
        mask = torch.ones_like(mask)
    mask_sum = mask.sum() + 1e-5
    err = ((render_out["color_fine"] - true_rgb) * mask[None, :, :]).reshape(-1, true_rgb.shape[-1])
    color_loss = F.l1_loss(err, torch.zeros_like(err), reduction="sum") / (mask_sum * n_lights)
    eik = render_out["gradient_error"]
    mask_loss = F.binary_cross_entropy(render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    loss = color_loss + eik * igr_weight + mask_loss * mask_weight
    return loss, {"color_loss": color_loss, "eikonal_loss": eik, "mask_loss": mask_loss}


# --------------------------------------------------------------------------------------
# R2i  parameter construction (geometric init) — models/fields.py:40-74, :161-172, :320-321
# --------------------------------------------------------------------------------------
def init_params(mc: ModelConf) -> Params:
    """Builds parameters with the same RNG call sequence as constructing the reference's
    SDFNetwork, SingleVarianceNetwork and RenderingNetwork in exp_runner.py:95-100 order
    (NeRF, constructed first there, is not on the path; callers wanting bit-identical
    streams seed immediately before this call, as the golden generator does)."""
    p: Params = {}
    sc = mc.sdf
    dims = sc.dims()
    n_lin = len(dims) - 1
    for l, (out_dim, in_dim) in enumerate(sc.layer_shapes()):
        lin = torch.nn.Linear(in_dim, out_dim)
        with torch.no_grad():
            if sc.geometric_init:
                if l == n_lin - 1:
                    mean = math.sqrt(math.pi) / math.sqrt(in_dim)
                    if not sc.inside_outside:
                        torch.nn.init.normal_(lin.weight, mean=mean, std=0.0001)
                        torch.nn.init.constant_(lin.bias, -sc.bias)
                    else:
                        torch.nn.init.normal_(lin.weight, mean=-mean, std=0.0001)
                        torch.nn.init.constant_(lin.bias, sc.bias)
                elif sc.multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, math.sqrt(2) / math.sqrt(out_dim))
                elif sc.multires > 0 and l in sc.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, math.sqrt(2) / math.sqrt(out_dim))
        _store_linear(p, f"sdf.lin{l}", lin, sc.weight_norm)
    p["dev.variance"] = torch.tensor(mc.init_val)
    cc = mc.color
    for l, (out_dim, in_dim) in enumerate(cc.layer_shapes()):
        lin = torch.nn.Linear(in_dim, out_dim)
        _store_linear(p, f"color.lin{l}", lin, cc.weight_norm)
    return p


def _store_linear(p: Params, prefix: str, lin: torch.nn.Linear, weight_norm: bool):
    w = lin.weight.detach().clone()
    p[prefix + ".bias"] = lin.bias.detach().clone()
    if weight_norm:
        p[prefix + ".weight_g"] = w.norm(dim=1, keepdim=True)
        p[prefix + ".weight_v"] = w
    else:
        p[prefix + ".weight"] = w


def param_order(mc: ModelConf, no_albedo: bool = False):
    """Order in which exp_runner.py:105-112 hands leaves to Adam (nerf omitted: no grads)."""
    names = []
    for l in range(mc.sdf.n_layers + 1):
        names += _leaf_names(f"sdf.lin{l}", mc.sdf.weight_norm)
    names.append("dev.variance")
    if not no_albedo:
        for l in range(mc.color.n_layers + 1):
            names += _leaf_names(f"color.lin{l}", mc.color.weight_norm)
    return names


def _leaf_names(prefix, weight_norm):
    if weight_norm:
        return [prefix + ".bias", prefix + ".weight_g", prefix + ".weight_v"]
    return [prefix + ".weight", prefix + ".bias"]


# --------------------------------------------------------------------------------------
# synthetic workload of SURVEY.md 8(d) (no dataset exists in the container or on the GPU box)
# --------------------------------------------------------------------------------------
def near_far_from_sphere(rays_o, rays_d):
    """models/dataset.py:448-458."""
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


def synthetic_batch(B: int, n_lights: int = 3, seed: int = 0, step: int = 0, n_views: int = 20,
                    warmup: bool = False):
    g = torch.Generator("cpu").manual_seed(seed * 1000003 + step)
    gv = torch.Generator("cpu").manual_seed(seed)
    centres = torch.randn(n_views, 3, generator=gv)
    centres = 3.0 * centres / centres.norm(dim=-1, keepdim=True)
    o = centres[step % n_views][None, :].expand(B, 3).contiguous()
    tgt = torch.randn(B, 3, generator=g)
    tgt = tgt / tgt.norm(dim=-1, keepdim=True) * (0.9 * torch.rand(B, 1, generator=g) ** (1.0 / 3.0))
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    near, far = near_far_from_sphere(o, d)
    t_rand = torch.rand(B, 1, generator=g)
    if warmup:
        tilt = torch.deg2rad(torch.tensor([0.0, 120.0, 240.0]))[:n_lights]
        slant = math.radians(30.0)
        L = -torch.stack([math.sin(slant) * torch.cos(tilt), math.sin(slant) * torch.sin(tilt),
                          math.cos(slant) * torch.ones_like(tilt)], dim=-1)
        lights = L.reshape(n_lights, 1, 1, 3).contiguous()
    else:
        L = torch.randn(n_lights, B, 1, 3, generator=g)
        lights = (L / L.norm(dim=-1, keepdim=True)).contiguous()
    true_rgb = torch.rand(n_lights, B, 3, generator=g)
    closest = o + d * (-(o * d).sum(-1, keepdim=True))
    mask = (closest.norm(dim=-1, keepdim=True) < 0.5).float()
    return {"rays_o": o, "rays_d": d, "near": near, "far": far, "t_rand": t_rand,
            "lights_dir": lights, "true_rgb": true_rgb, "mask": mask}


def sphere_scene_batch(B: int, n_lights: int = 3, seed: int = 0, step: int = 0, n_views: int = 20,
                       warmup: bool = False, radius: float = 0.5, albedo=(0.7, 0.5, 0.3)):
    """A learnable synthetic capture for the convergence test (SURVEY 8d has no dataset to lean on): the same
    ray stream as `synthetic_batch`, but the targets are renderings of an analytic Lambertian sphere of the given
    radius and constant albedo, `true_rgb[l] = albedo * max(n . l, 0)` inside the silhouette and 0 outside — the
    image model the reference's photometric-stereo inputs follow (models/dataset.py:255-300 builds `images` from
    normal and albedo maps this way).  Lights: the three warm-up directions (models/dataset.py:257-266) or, in main
    mode, per-ray unit vectors in the hemisphere around the true normal."""
    b = synthetic_batch(B, n_lights=n_lights, seed=seed, step=step, n_views=n_views, warmup=warmup)
    o, d = b["rays_o"], b["rays_d"]
    tca = -(o * d).sum(-1, keepdim=True)
    closest = o + d * tca
    c2 = (closest * closest).sum(-1, keepdim=True)
    hit = c2 < radius * radius
    thc = torch.sqrt(torch.clamp(radius * radius - c2, min=0.0))
    p_hit = o + d * (tca - thc)
    n = p_hit / radius
    g = torch.Generator("cpu").manual_seed(seed * 7919 + step + 17)
    if warmup:
        # the three warm-up lights live in CAMERA space (dataset.py:257-266, u = -[sin s cos t, sin s sin t, cos s])
        # and are rotated into the world by the view's pose: a camera at c looking at the origin
        fwd = -o[0] / o[0].norm()
        up0 = torch.tensor([0.0, 0.0, 1.0]) if abs(float(fwd[2])) < 0.9 else torch.tensor([1.0, 0.0, 0.0])
        right = torch.linalg.cross(up0, fwd)
        right = right / right.norm()
        up = torch.linalg.cross(fwd, right)
        u = b["lights_dir"].reshape(n_lights, 3)                  # camera-space directions
        lw = u[:, 0:1] * right[None] + u[:, 1:2] * up[None] + u[:, 2:3] * fwd[None]
        lights = lw.reshape(n_lights, 1, 1, 3).contiguous()
        ldir = lw.reshape(n_lights, 1, 3).expand(n_lights, B, 3)
    else:
        r = torch.randn(n_lights, B, 3, generator=g)
        r = r / r.norm(dim=-1, keepdim=True)
        r = torch.where(((r * n[None]).sum(-1, keepdim=True) < 0), -r, r)   # flip into the normal's hemisphere
        ldir = torch.where(hit[None].expand(n_lights, B, 1), r, -d[None].expand(n_lights, B, 3))
        ldir = ldir / ldir.norm(dim=-1, keepdim=True)
        lights = ldir.reshape(n_lights, B, 1, 3).contiguous()
    shade = torch.clamp((ldir * n[None]).sum(-1, keepdim=True), min=0.0)          # [L,B,1]
    alb = torch.tensor(albedo, dtype=torch.float32).reshape(1, 1, 3)
    rgb = torch.where(hit[None], alb * shade, torch.zeros(1))
    out = dict(b)
    out.update({"lights_dir": lights, "true_rgb": rgb.contiguous(), "mask": hit.float()})
    return out


def lr_factor(iter_step: int, warm_up_end: int, end_iter: int, alpha: float) -> float:
    """Learning-rate schedule of exp_runner.py:320-332 (linear warm-up, then cosine decay to alpha)."""
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


# --------------------------------------------------------------------------------------
# 8f-2  per-step ray / target generation — models/dataset.py:351-376, exp_runner.py:214-220
# --------------------------------------------------------------------------------------
def gen_rays_at_view(ds: dict, img_idx: int, pixels_x: torch.Tensor, pixels_y: torch.Tensor):
    """Restatement of Dataset.ps_gen_random_rays_at_view_on_all_lights (dataset.py:359-376) for given pixel
    draws, plus the light gather of exp_runner.py:218 and near_far_from_sphere (dataset.py:448-458).
    `ds` holds the Dataset tensors: images, images_warmup, masks, light_directions, intrinsics_all_inv, pose_all."""
    images_warmup = ds["images_warmup"][img_idx, :, pixels_y, pixels_x, :]          # dataset.py:359
    images = ds["images"][img_idx, :, pixels_y, pixels_x, :]                        # :360
    mask = ds["masks"][img_idx][(pixels_y, pixels_x)]                               # :363
    p = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], dim=-1).float()   # :365
    p = torch.matmul(ds["intrinsics_all_inv"][img_idx, None, :3, :3], p[:, :, None]).squeeze()   # :367
    rays_v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)                  # :369
    rays_v = torch.matmul(ds["pose_all"][img_idx, None, :3, :3], rays_v[:, :, None]).squeeze()   # :371
    rays_o = ds["pose_all"][img_idx, None, :3, 3].expand(rays_v.shape)              # :373
    data = torch.cat([rays_o, rays_v, mask[:, :1]], dim=-1)                         # :376
    lights_dir = ds["light_directions"][img_idx, :, pixels_y, pixels_x, :]          # exp_runner.py:218
    near, far = near_far_from_sphere(data[:, :3], data[:, 3:6])
    return {"data": data, "images_warmup": images_warmup, "images": images, "lights_dir": lights_dir,
            "near": near, "far": far}
