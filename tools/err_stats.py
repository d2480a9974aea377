"""GPU dev script: body of test_full_batch_512_matches_oracle with the error statistics printed (max and rms of
|hip - fp64| and |fp32 oracle - fp64| for every output).  Run from the repo root (or from a worktree of another
commit, to compare two builds): python tools/err_stats.py"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from oracle import rnb_oracle as O                    # noqa: E402
from tests.golden_util import Golden                  # noqa: E402
import rnb_neus_fork_amd as R                         # noqa: E402

dev_ = torch.device("cuda:0")
g = Golden("full_main_b512")
mc, p = g.mc, g.params()
sdf, dev, col, ren = R.build_from_named_params(mc, p, dev_)
for seed in (22, 23, 24):
    batch = O.synthetic_batch(512, seed=seed, step=7, warmup=False)
    b = {k: v.to(dev_) for k, v in batch.items()}
    with torch.no_grad():
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"])
    z = ren.last_z_vals.cpu()
    torch.set_num_threads(16)
    with torch.no_grad():
        pr = {k: v.double() for k, v in p.items()}
        b64 = {k: v.double() for k, v in batch.items()}
        ref = O.render_rnb(pr, mc, b64["rays_o"], b64["rays_d"], b64["near"], b64["far"], b64["lights_dir"],
                           cos_anneal_ratio=1.0, z_vals=z.double())
        ref32 = O.render_rnb(p, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"],
                             batch["lights_dir"], cos_anneal_ratio=1.0, z_vals=z)
    for k in ("sdf", "color_fine", "weights", "weight_sum", "gradients", "cdf_fine", "gradient_error"):
        if k not in out or k not in ref:
            continue
        r64 = ref[k].double()
        eh = (out[k].cpu().double() - r64).abs()
        er = (ref32[k].double() - r64).abs()
        print(f"seed {seed} {k:15s} hip max {float(eh.max()):.3e} rms {float((eh**2).mean().sqrt()):.3e} | "
              f"fp32 oracle max {float(er.max()):.3e} rms {float((er**2).mean().sqrt()):.3e}", flush=True)
