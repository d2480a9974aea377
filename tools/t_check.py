"""Development aid (GPU box): the register-tile sweeps (RNB_VARIANT_REG_TILE, csrc/fused_t.hip) against the LDS-tile
kernels and the CPU oracle on the full-size network, plus timings of the forward-only sweep.
usage: python tools/t_check.py [n_points]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rnb_neus_fork_amd as R  # noqa: E402
from rnb_neus_fork_amd import native, runtime  # noqa: E402
from oracle import rnb_oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    dev = torch.device("cuda:0")
    mc = O.ModelConf()
    torch.manual_seed(0)
    p = O.init_params(mc)
    # a trained-looking state: perturb every leaf a little so that no layer is special
    g = torch.Generator().manual_seed(1)
    for k in p:
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=g)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
    pts = (torch.rand(n, 3, generator=g) * 2 - 1) * 0.9
    ref = O.sdf_forward(p, mc.sdf, pts[:4096])
    d_pts = pts.to(dev)
    outs = {}
    for tag, kw in (("lds", dict(lds_tile=True)), ("reg", dict(reg_tile=True))):
        ren.set_variant(**kw)
        packed = ren._pack(True)
        out = runtime.sdf_forward(ren.desc, packed, d_pts, True)
        torch.cuda.synchronize()
        outs[tag] = out.cpu()
        err = (outs[tag][:4096] - ref).abs()
        print(f"{tag}: max |sdf - oracle| {float(err[:, 0].max()):.3e}   max |feat - oracle| {float(err[:, 1:].max()):.3e}")
        o1 = runtime.sdf_forward(ren.desc, packed, d_pts, False).cpu()
        print(f"{tag}: sdf-only vs with-feature max diff {float((o1 - outs[tag][:, :1]).abs().max()):.3e}")
    d = (outs["reg"] - outs["lds"]).abs()
    print(f"reg vs lds: sdf {float(d[:, 0].max()):.3e} feat {float(d[:, 1:].max()):.3e}")
    if len(sys.argv) > 2 and sys.argv[2] == "notime":
        return
    # timing: forward-only sweep over 1M points
    big = (torch.rand(1 << 20, 3, device=dev) * 2 - 1) * 0.9
    for tag, kw in (("lds", dict(lds_tile=True)), ("reg", dict(reg_tile=True)), ("lds", dict(lds_tile=True)), ("reg", dict(reg_tile=True))):
        ren.set_variant(**kw)
        packed = ren._pack(True)
        for _ in range(2):
            runtime.sdf_forward(ren.desc, packed, big, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            runtime.sdf_forward(ren.desc, packed, big, False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{tag}: {dt * 1e3:.3f} ms per 1M-point forward-only sweep = {1.049e6 * (1 << 20) / dt / 1e12:.1f} TFLOP/s algorithmic")
    ren.set_variant()


if __name__ == "__main__":
    main()
