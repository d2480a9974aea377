"""Stability soak (development / evidence tool; GPU box): N train_rnb steps of the default (non-deterministic x3) variant on
the analytic sphere capture of the convergence test, 512 rays x (64+64) samples, schedule of exp_runner.py:320-332 scaled
to N; prints the loss every N/10 steps, the final PSNR on held-out batches, and whether every loss was finite.
`range`: every 100 steps one step is rendered with `track_range` and the largest operand magnitudes of that step are printed
(NeuSRenderer.range_report: weights, SDF-network activations, Jacobian rows, albedo activations, loss adjoints) — how far a
training run stays from the range the fixed fp16 scales of round 4 assumed (255 / 1023); at the end the same for the
reference-trained, sharpened state of tests/golden/full_main_sharp.npz.
usage: python tools/soak.py [steps] [bf16 | range]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import rnb_neus_fork_amd as R
from oracle import rnb_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
torch.manual_seed(0)
sdf = R.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                   geometric_init=True, weight_norm=True).to(dev)
devn = R.SingleVarianceNetwork(0.3).to(dev)
col = R.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, weight_norm=True,
                         multires_view=4, squeeze_out=True).to(dev)
ren = R.NeuSRenderer(None, sdf, devn, col, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    ren.set_variant(bf16=True)          # RNB_VARIANT_BF16 (BASELINE config 5's arithmetic)
    print("variant: bf16 sweeps")
log_range = len(sys.argv) > 2 and sys.argv[2] == "range"
opt = R.FlatAdam(list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters()), lr=5e-4)
B = 512
losses = []
t0 = time.time()
for it in range(steps):
    opt.param_groups[0]["lr"] = 5e-4 * O.lr_factor(it, steps // 10, steps, 0.05)
    warm = it < steps // 2
    b = {k: v.to(dev) for k, v in O.sphere_scene_batch(B, seed=31, step=it, warmup=warm).items()}
    fn = ren.render_rnb_warmup if warm else ren.render_rnb
    ren.track_range = log_range and it % 100 == 0
    out = fn(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0, t_rand=b["t_rand"])
    loss, _ = R.rnb_loss(out, b["true_rgb"], b["mask"])
    opt.zero_grad()
    loss.backward()
    opt.step()
    losses.append(loss.detach())
    if ren.track_range:
        r = ren.range_report()
        print(f"RANGE step {it}: " + "  ".join(f"{k[8:]} {v:.4g}" for k, v in r.items()), flush=True)
    if (it + 1) % max(steps // 10, 1) == 0:
        print(f"step {it + 1}: loss {float(loss):.5f}  inv_s {float(torch.exp(devn.variance * 10)):.1f}  ({time.time() - t0:.0f} s)", flush=True)
L = torch.stack(losses).cpu().numpy()
se, n = 0.0, 0
with torch.no_grad():
    for k in range(8):
        b = {kk: v.to(dev) for kk, v in O.sphere_scene_batch(B, seed=31, step=100000 + k, warmup=False).items()}
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], perturb_overwrite=0, cos_anneal_ratio=1.0)
        m = b["mask"][None]
        se += float((((out["color_fine"] - b["true_rgb"]) * m) ** 2).sum())
        n += int(m.sum()) * 9
print(f"all {steps} losses finite: {bool(np.isfinite(L).all())}; first {L[0]:.4f} last-decile mean {L[-steps // 10:].mean():.4f}; "
      f"held-out PSNR {-10.0 * np.log10(se / n):.2f} dB; weight_sum vs mask L1 "
      f"{float((out['weight_sum'] - b['mask']).abs().mean()):.4f}")
if log_range:   # the reference-trained, sharpened full-size state of the golden fixtures (inv_s ~ 403)
    from tests.golden_util import Golden
    g = Golden("full_main_sharp")
    s2, d2, c2, r2 = R.build_from_named_params(g.mc, g.params(), dev)
    r2.track_range = True
    bb = {kk: v.to(dev) for kk, v in O.synthetic_batch(512, seed=22, step=7, warmup=False).items()}
    o2 = r2.render_rnb(bb["rays_o"], bb["rays_d"], bb["near"], bb["far"], bb["lights_dir"], cos_anneal_ratio=1.0, t_rand=bb["t_rand"])
    R.rnb_loss(o2, bb["true_rgb"], bb["mask"])[0].backward()
    print("RANGE full_main_sharp (512 rays): " + "  ".join(f"{k[8:]} {v:.4g}" for k, v in r2.range_report().items()))
# a mesh of the trained surface: closed, genus 0, radius ~0.5
v, t = ren.extract_geometry(torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0]), 128, backend="native")
from oracle import mc_oracle as M
V, E, F, euler, closed = M.mesh_report(v, t)
r = np.linalg.norm(v, axis=1)
print(f"mesh at 128^3: {len(v)} vertices, {len(t)} triangles, closed {closed}, Euler {euler}, radius {r.min():.3f} .. {r.max():.3f}")
