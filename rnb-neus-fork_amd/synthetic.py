"""Synthetic ray batches with the shape and statistics of a DiLiGenT-MV training batch: what
`Dataset.ps_gen_random_rays_at_view_on_all_lights` (models/dataset.py:400-446) hands to `train_rnb` — rays from
one of 20 cameras on a radius-3 sphere towards the unit ball, near/far from the unit sphere
(models/dataset.py:448-458), 3 light directions per ray, target colours and a foreground mask.  Used by bench.py
(the dataset itself needs image files that exist nowhere here).  Pure host-side tensor construction."""
from __future__ import annotations

import math

import torch


def near_far_from_sphere(rays_o, rays_d):
    """models/dataset.py:448-458."""
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


def synthetic_batch(n_rays, n_lights=3, seed=0, step=0, n_views=20, warmup=False):
    g = torch.Generator("cpu").manual_seed(seed * 1000003 + step)
    gv = torch.Generator("cpu").manual_seed(seed)
    centres = torch.randn(n_views, 3, generator=gv)
    centres = 3.0 * centres / centres.norm(dim=-1, keepdim=True)
    o = centres[step % n_views][None, :].expand(n_rays, 3).contiguous()
    tgt = torch.randn(n_rays, 3, generator=g)
    tgt = tgt / tgt.norm(dim=-1, keepdim=True) * (0.9 * torch.rand(n_rays, 1, generator=g) ** (1.0 / 3.0))
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    near, far = near_far_from_sphere(o, d)
    t_rand = torch.rand(n_rays, 1, generator=g)
    if warmup:   # render_rnb_warmup: one light set shared by all rays (models/renderer.py:828-846)
        tilt = torch.deg2rad(torch.tensor([0.0, 120.0, 240.0]))[:n_lights]
        slant = math.radians(30.0)
        L = -torch.stack([math.sin(slant) * torch.cos(tilt), math.sin(slant) * torch.sin(tilt),
                          math.cos(slant) * torch.ones_like(tilt)], dim=-1)
        lights = L.reshape(n_lights, 1, 1, 3).contiguous()
    else:
        L = torch.randn(n_lights, n_rays, 1, 3, generator=g)
        lights = (L / L.norm(dim=-1, keepdim=True)).contiguous()
    true_rgb = torch.rand(n_lights, n_rays, 3, generator=g)
    closest = o + d * (-(o * d).sum(-1, keepdim=True))
    mask = (closest.norm(dim=-1, keepdim=True) < 0.5).float()
    return {"rays_o": o, "rays_d": d, "near": near, "far": far, "t_rand": t_rand,
            "lights_dir": lights, "true_rgb": true_rgb, "mask": mask}
