"""GPU dev script: per-parameter gradient error (rel-L2 against the fp64 oracle) of the fp32 CPU oracle, the six-term bf16
arithmetic (x2h off) and the default (x2h), deterministic reductions, fixture full_main_b512 on 512 synthetic rays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rnb_neus_fork_amd as R  # noqa: E402
from oracle import rnb_oracle as O  # noqa: E402
from tests.golden_util import Golden  # noqa: E402

dev = torch.device("cuda:0")
g = Golden("full_main_b512")
mc, p = g.mc, g.params()
sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
named = {("sdf." + k): v for k, v in sdf.named_parameters()}
named["dev.variance"] = devn.variance
named.update({("color." + k): v for k, v in col.named_parameters()})
batch = O.synthetic_batch(512, seed=22, step=7, warmup=False)
b = {k: v.to(dev) for k, v in batch.items()}
b64 = {k: v.double() for k, v in batch.items()}
torch.set_num_threads(16)
res = {}
z = None
for tag, kw in (("x3", dict(x2h=False, deterministic=True)), ("x2h", dict(deterministic=True))):
    ren.set_variant(**kw)
    for x in named.values():
        x.grad = None
    out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0, t_rand=b["t_rand"],
                         z_vals=z)
    z = ren.last_z_vals
    O.rnb_loss(out, b["true_rgb"], b["mask"])[0].backward()
    res[tag] = {k: v.grad.detach().cpu().double().clone() for k, v in named.items()}
zc = z.cpu()
pr = {k: v.double().requires_grad_(True) for k, v in p.items()}
r = O.render_rnb(pr, mc, b64["rays_o"], b64["rays_d"], b64["near"], b64["far"], b64["lights_dir"], cos_anneal_ratio=1.0, z_vals=zc.double())
O.rnb_loss(r, b64["true_rgb"], b64["mask"])[0].backward()
p32 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
r32 = O.render_rnb(p32, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"], batch["lights_dir"], cos_anneal_ratio=1.0, z_vals=zc)
O.rnb_loss(r32, batch["true_rgb"], batch["mask"])[0].backward()
print(f"{'parameter':28s} {'fp32 CPU':>10s} {'x3':>10s} {'x2h':>10s}   (rel-L2 of the gradient against fp64)")
worst = {"fp32": 0.0, "x3": 0.0, "x2h": 0.0}
import math
lg = {"fp32": 0.0, "x3": 0.0, "x2h": 0.0}
for k in named:
    rg = pr[k].grad
    n = float(rg.norm())
    e32 = float((p32[k].grad.double() - rg).norm()) / n
    e3 = float((res["x3"][k].reshape(rg.shape) - rg).norm()) / n
    e2 = float((res["x2h"][k].reshape(rg.shape) - rg).norm()) / n
    for t, e in (("fp32", e32), ("x3", e3), ("x2h", e2)):
        worst[t] = max(worst[t], e)
        lg[t] += math.log(e)
    print(f"{k:28s} {e32:10.2e} {e3:10.2e} {e2:10.2e}")
print("worst:", {t: f"{v:.2e}" for t, v in worst.items()}, " geometric mean:", {t: f"{math.exp(v / len(named)):.2e}" for t, v in lg.items()})
