"""GPU dev script: where does |weight_sum - fp64| of the device path come from?  The composite of _core_common is
re-evaluated in fp64 on (a) the device's own sdf / normals and (b) the fp32 oracle's sdf / normals, which separates
the error carried by the network outputs from the error of the composite arithmetic itself."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.getcwd())
from oracle import rnb_oracle as O                    # noqa: E402
from tests.golden_util import Golden                  # noqa: E402
import rnb_neus_fork_amd as R                         # noqa: E402


def composite(sdf, g3, rays_d, z, inv_s, sample_dist, c=1.0):
    B, S = z.shape
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], sample_dist)], -1)
    dirs = rays_d[:, None, :].expand(B, S, 3)
    tc = (dirs * g3).sum(-1)
    ic = -(F.relu(-tc * 0.5 + 0.5) * (1.0 - c) + F.relu(-tc) * c)
    s = sdf.reshape(B, S)
    pc = torch.sigmoid((s - ic * dists * 0.5) * inv_s)
    nc = torch.sigmoid((s + ic * dists * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
    trans = torch.cumprod(torch.cat([torch.ones([B, 1], dtype=z.dtype), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    w = alpha * trans
    return w, w.sum(-1, keepdim=True)


dev_ = torch.device("cuda:0")
g = Golden("full_main_b512")
mc, p = g.mc, g.params()
sdf, dev, col, ren = R.build_from_named_params(mc, p, dev_)
ren.want_extras = True
torch.set_num_threads(16)
for seed in (22, 23):
    batch = O.synthetic_batch(512, seed=seed, step=7, warmup=False)
    b = {k: v.to(dev_) for k, v in batch.items()}
    with torch.no_grad():
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"])
        z = ren.last_z_vals.cpu()
        pr = {k: v.double() for k, v in p.items()}
        b64 = {k: v.double() for k, v in batch.items()}
        sd = 2.0 / mc.render.n_samples
        c64 = O._core_common(pr, mc, b64["rays_o"], b64["rays_d"], z.double(), sd, 1.0)
        c32 = O._core_common(p, mc, batch["rays_o"], batch["rays_d"], z, sd, 1.0)
    inv64 = float(O.inv_s_of(pr))
    ws64 = c64["weights"].sum(-1, keepdim=True)
    h_sdf = ren.last_extras["sdf"].cpu().double()
    h_g = out["gradients"].cpu().double()
    rms = lambda e: float((e ** 2).mean().sqrt())
    print(f"seed {seed}: inv_s {inv64:.3f}")
    print("  sdf     rms err: hip %.3e  fp32 oracle %.3e" % (rms(h_sdf - c64["sdf"]), rms(c32["sdf"].double() - c64["sdf"])))
    print("  normals rms err: hip %.3e  fp32 oracle %.3e" % (rms(h_g - c64["gradients"]), rms(c32["gradients"].double() - c64["gradients"])))
    for name, s_, g_ in (("hip", h_sdf, h_g), ("fp32 oracle", c32["sdf"].double(), c32["gradients"].double())):
        _, ws_in = composite(s_, g_, b64["rays_d"], z.double(), inv64, sd)
        _, ws_sdf = composite(s_, c64["gradients"], b64["rays_d"], z.double(), inv64, sd)
        _, ws_nrm = composite(c64["sdf"], g_, b64["rays_d"], z.double(), inv64, sd)
        got = out["weight_sum"].cpu().double() if name == "hip" else c32["weights"].sum(-1, keepdim=True).double()
        print(f"  {name:12s} weight_sum: total rms {rms(got - ws64):.3e} max {float((got - ws64).abs().max()):.3e} | "
              f"from its sdf+normals (fp64 composite) rms {rms(ws_in - ws64):.3e} (sdf alone {rms(ws_sdf - ws64):.3e}, "
              f"normals alone {rms(ws_nrm - ws64):.3e}) | composite arithmetic rms {rms(got - ws_in):.3e}", flush=True)
    # correlation of the sdf error along a ray: rms of the per-ray MEAN error over the 128 samples vs rms / sqrt(128)
    for name, s_ in (("hip", h_sdf), ("fp32 oracle", c32["sdf"].double())):
        e = (s_ - c64["sdf"]).reshape(512, -1)
        print(f"  {name:12s} sdf error: per-sample rms {rms(e):.3e}, per-ray mean rms {rms(e.mean(-1)):.3e} "
              f"(independent errors would give {rms(e) / e.shape[1] ** 0.5:.3e}), global mean {float(e.mean()):+.3e}")
