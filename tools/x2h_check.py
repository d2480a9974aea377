"""Development aid (GPU box): the fp16 three-term forward sweeps (RNB_VARIANT_X2H) against the bf16 six-term ones and the
fp64 CPU oracle on the full-size network in the sharpened state of the fixtures, plus timings.
usage: python tools/x2h_check.py [notime]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rnb_neus_fork_amd as R  # noqa: E402
from rnb_neus_fork_amd import runtime  # noqa: E402
from oracle import rnb_oracle as O  # noqa: E402
from tests.golden_util import Golden  # noqa: E402


def stats(name, got, ref):
    e = got.double() - ref
    print(f"  {name:24s} max {float(e.abs().max()):.3e} rms {float((e ** 2).mean().sqrt()):.3e} mean {float(e.mean()):+.3e}")


def main():
    dev = torch.device("cuda:0")
    g = Golden("full_main_b512")
    mc, p = g.mc, g.params()
    sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
    gen = torch.Generator().manual_seed(1)
    n = 40000
    pts = (torch.rand(n, 3, generator=gen) * 2 - 1) * 0.9
    p64 = {k: v.double() for k, v in p.items()}
    ref = O.sdf_forward(p64, mc.sdf, pts[:8192].double())
    ref32 = O.sdf_forward(p, mc.sdf, pts[:8192])
    print("fp32 CPU oracle vs fp64:")
    stats("sdf", ref32[:, 0], ref[:, 0])
    stats("features", ref32[:, 1:], ref[:, 1:])
    d_pts = pts.to(dev)
    outs = {}
    for tag, kw in (("x3 (6 bf16 terms)", dict(x2h=False)), ("x2h (3 fp16 terms)", dict(x2h=True))):
        ren.set_variant(**kw)
        packed = ren._pack(True)
        out = runtime.sdf_forward(ren.desc, packed, d_pts, True)
        torch.cuda.synchronize()
        outs[tag] = out.cpu()
        print(f"{tag} vs fp64:")
        stats("sdf", outs[tag][:8192, 0], ref[:, 0])
        stats("features", outs[tag][:8192, 1:], ref[:, 1:])
        o1 = runtime.sdf_forward(ren.desc, packed, d_pts[:8192], False).cpu()      # the small-batch kernels
        stats("sdf (32-point tiles)", o1[:, 0], ref[:, 0])
    # end to end: training-mode render (the SAVE kernel) against the fp64 oracle
    batch = O.synthetic_batch(512, seed=22, step=7, warmup=False)
    b = {k: v.to(dev) for k, v in batch.items()}
    b64 = {k: v.double() for k, v in batch.items()}
    for tag, kw in (("x3", dict(x2h=False)), ("x2h", dict(x2h=True))):
        ren.set_variant(**kw)
        for x in list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters()):
            x.grad = None
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0, t_rand=b["t_rand"])
        O.rnb_loss(out, b["true_rgb"], b["mask"])[0].backward()
        z = ren.last_z_vals.cpu()
        pr = {k: v.double().requires_grad_(True) for k, v in p.items()}
        r = O.render_rnb(pr, mc, b64["rays_o"], b64["rays_d"], b64["near"], b64["far"], b64["lights_dir"], cos_anneal_ratio=1.0,
                         z_vals=z.double())
        O.rnb_loss(r, b64["true_rgb"], b64["mask"])[0].backward()
        print(f"{tag}: render_rnb (training mode) vs fp64 on its own depths:")
        for k in ("color_fine", "weights", "weight_sum", "gradients", "cdf_fine"):
            stats(k, out[k].detach().cpu(), r[k].detach())
        worst = 0.0
        named = {("sdf." + k): v for k, v in sdf.named_parameters()}
        named["dev.variance"] = devn.variance
        named.update({("color." + k): v for k, v in col.named_parameters()})
        for k, v in named.items():
            rg = pr[k].grad
            worst = max(worst, float((v.grad.cpu().double() - rg).norm() / rg.norm()))
        print(f"  worst parameter-gradient rel-L2 vs fp64: {worst:.3e}")
    if len(sys.argv) > 1 and sys.argv[1] == "notime":
        return
    big = (torch.rand(1 << 20, 3, device=dev) * 2 - 1) * 0.9
    small = big[:8192].contiguous()
    for tag, kw in (("x3", dict(x2h=False)), ("x2h", dict(x2h=True)), ("x3", dict(x2h=False)), ("x2h", dict(x2h=True))):
        ren.set_variant(**kw)
        packed = ren._pack(True)
        for pts_, what, reps in ((big, "1M-point", 10), (small, "8192-point", 200)):
            for _ in range(3):
                runtime.sdf_forward(ren.desc, packed, pts_, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                runtime.sdf_forward(ren.desc, packed, pts_, False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{tag}: {dt * 1e3:.3f} ms per {what} forward-only sweep = {1.049e6 * pts_.shape[0] / dt / 1e12:.1f} TFLOP/s algorithmic")
    ren.set_variant()


if __name__ == "__main__":
    main()
